#!/bin/bash
O=gpurun_out/r05; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/j6_tests.log 2>&1; echo "tests rc $?"; tail -6 $O/j6_tests.log
for m in traverse locus-table; do python bench.py --steps 10 --warmup 3 --lean --no-check --mode $m > $O/j6_bench_$m.json 2> $O/j6_bench_$m.err; python -c "import json;d=json.load(open('$O/j6_bench_$m.json'));print('$m',d['ms_per_step'],d['roofline']['kernel_ms_per_step'])" || tail -5 $O/j6_bench_$m.err; done
E2E_TRACE=1 timeout 600 python tools/e2e_packed.py > $O/j6_e2e_packed.jsonl 2> $O/j6_e2e_trace.log; echo "e2e rc $?"; grep "default\|alternated\|32 Mi\|16 Mi" $O/j6_e2e_packed.jsonl; grep "psigpu\]\|traced" $O/j6_e2e_trace.log | head -24
