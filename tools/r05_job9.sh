#!/bin/bash
O=gpurun_out/r05; mkdir -p $O
for m in traverse; do python bench.py --steps 10 --warmup 3 --lean --no-check --mode $m > $O/j9_bench_$m.json 2> $O/j9_bench_$m.err; python -c "import json;d=json.load(open('$O/j9_bench_$m.json'));print('$m',d['ms_per_step'],d['roofline']['kernel_ms_per_step'])" || tail -5 $O/j9_bench_$m.err; done
R=$PWD; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/j9_prof_t -- python3 $R/bench.py --lean --steps 5 --warmup 2 --no-check --mode traverse > $R/$O/j9_prof_t.log 2>&1
cd $R
f=$(ls -t $O/j9_prof_t/*/*kernel_stats.csv | head -1); python - <<PY
import csv
rows=list(csv.reader(open("$f")))
for r in rows[:1]+[r for r in rows[1:] if int(r[1])>=5][:8]:
    print(r[0].replace('(anonymous namespace)::','')[:70], r[1:5])
PY
for v in "" "E2E_NO_NUMA=1" ""; do
  env $v E2E_QUICK=1 timeout 300 python tools/e2e_packed.py > $O/j9_e2e.jsonl 2> $O/j9_e2e.log
  echo "== e2e $v"; grep "default\|alternated" $O/j9_e2e.jsonl | cut -c1-230
done
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/j9_tests.log 2>&1; echo "tests rc $?"; tail -6 $O/j9_tests.log
