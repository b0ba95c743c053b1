#!/bin/bash
# job 15: the filter without its 12-mer stage, packed records only for records in read order, twelve widening threads
O=gpurun_out/r05; mkdir -p $O
( time timeout 1200 python -m pytest tests -m gpu -x -q -k "wire or host_entry or two_sub or traverse or query_modes or random_graphs or one_kernel or device_entry or oversubscribed or packed" ) > $O/j15_tests.log 2>&1; echo "tests rc $?"; tail -5 $O/j15_tests.log | cut -c1-300
E2E_QUICK=1 timeout 300 python tools/e2e_packed.py 2>/dev/null | grep "wire formats\|default" | cut -c1-400
python bench.py --steps 10 --warmup 3 --lean --no-check --mode traverse > $O/j15_bench_traverse.json 2> $O/j15_bench_traverse.err; python -c "import json;d=json.load(open('$O/j15_bench_traverse.json'));print('traverse',d['ms_per_step'],d['roofline']['kernel_ms_per_step'])" || tail -5 $O/j15_bench_traverse.err
R=$PWD; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/j15_prof_t -- python3 $R/bench.py --lean --steps 5 --warmup 2 --no-check --mode traverse > $R/$O/j15_prof_t.log 2>&1
cd $R
python - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/r05/j15_prof_t/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:8]:
        print(r['Name'][:80].ljust(80), r['Calls'], r['AverageNs'], r['Percentage'])
PY
