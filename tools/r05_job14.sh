#!/bin/bash
# job 14: does the SIGSEGV of job 13 (host index build inside test_golden, after the traverse-mode cases) come again?  native backtrace on
O=gpurun_out/r05; mkdir -p $O
for i in 1 2 3; do
  PSIGPU_SEGV_TRACE=1 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden" > $O/j14_golden_$i.log 2>&1; echo "golden run $i rc $?"; tail -3 $O/j14_golden_$i.log | cut -c1-200
  grep -n "psigpu\] fatal\|#[0-9]\|Fatal Python" $O/j14_golden_$i.log | head -30
done
( time timeout 1200 python -m pytest tests -m gpu -x -q -k "wire or host_entry or two_sub or traverse or query_modes or random_graphs or one_kernel or device_entry or oversubscribed or packed" ) > $O/j14_tests.log 2>&1; echo "tests rc $?"; tail -8 $O/j14_tests.log | cut -c1-300
for t in 8 12 16; do E2E_QUICK=1 E2E_WIDEN_THREADS=$t timeout 300 python tools/e2e_packed.py 2>/dev/null | grep "wire formats\|default" | cut -c1-400; done
python bench.py --steps 10 --warmup 3 --lean --no-check --mode traverse > $O/j14_bench_traverse.json 2> $O/j14_bench_traverse.err; python -c "import json;d=json.load(open('$O/j14_bench_traverse.json'));print('traverse',d['ms_per_step'],d['roofline']['kernel_ms_per_step'])" || tail -5 $O/j14_bench_traverse.err
R=$PWD; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/j14_prof_t -- python3 $R/bench.py --lean --steps 5 --warmup 2 --no-check --mode traverse > $R/$O/j14_prof_t.log 2>&1
cd $R
python - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/r05/j14_prof_t/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:8]:
        print(r['Name'][:80].ljust(80), r['Calls'], r['AverageNs'], r['Percentage'])
PY
