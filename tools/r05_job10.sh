#!/bin/bash
O=gpurun_out/r05; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "header_api or (traverse and (golden or random_graphs or snv_graph or spill or hla or long_seeds))" > $O/j10_tests.log 2>&1; echo "tests rc $?"; tail -4 $O/j10_tests.log
python bench.py --steps 10 --warmup 3 --lean --no-check --mode traverse > $O/j10_bench_traverse.json 2> $O/j10_bench_traverse.err; python -c "import json;d=json.load(open('$O/j10_bench_traverse.json'));print('traverse',d['ms_per_step'],d['roofline']['kernel_ms_per_step'])" || tail -5 $O/j10_bench_traverse.err
R=$PWD; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/j10_prof_t -- python3 $R/bench.py --lean --steps 5 --warmup 2 --no-check --mode traverse > $R/$O/j10_prof_t.log 2>&1
cd $R
f=$(ls -t $O/j10_prof_t/*/*kernel_stats.csv | head -1); python - <<PY
import csv
rows=list(csv.reader(open("$f")))
for r in rows[:1]+[r for r in rows[1:] if int(r[1])>=5][:6]:
    print(r[0].replace('(anonymous namespace)::','')[:70], r[1:5])
PY
