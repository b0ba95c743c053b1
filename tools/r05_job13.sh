#!/bin/bash
# job 13: the packed wire records, the traverser's segmented roots and the one-kernel step in traverse mode
O=gpurun_out/r05; mkdir -p $O
( time timeout 1200 python -m pytest tests -m gpu -x -q -k "wire or host_entry or two_sub or traverse or query_modes or random_graphs or one_kernel or device_entry or oversubscribed or golden or packed" ) > $O/j13_tests.log 2>&1; echo "tests rc $?"; tail -8 $O/j13_tests.log
E2E_QUICK=1 E2E_TRACE=1 timeout 600 python tools/e2e_packed.py > $O/j13_e2e.jsonl 2> $O/j13_e2e.log; echo "e2e rc $?"; cat $O/j13_e2e.jsonl; grep "psigpu\]" $O/j13_e2e.log | head -40
for m in traverse kmer-table; do python bench.py --steps 10 --warmup 3 --lean --no-check --mode $m > $O/j13_bench_$m.json 2> $O/j13_bench_$m.err; python -c "import json;d=json.load(open('$O/j13_bench_$m.json'));print('$m',d['ms_per_step'],d['roofline']['kernel_ms_per_step'])" || tail -5 $O/j13_bench_$m.err; done
R=$PWD; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/j13_prof_t -- python3 $R/bench.py --lean --steps 5 --warmup 2 --no-check --mode traverse > $R/$O/j13_prof_t.log 2>&1
cd $R
python - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/r05/j13_prof_t/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:14]:
        print(r['Name'][:80].ljust(80), r['Calls'], r['AverageNs'], r['Percentage'])
PY
