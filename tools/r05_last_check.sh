#!/bin/bash
# The last GPU job of round 5: what the driver runs at round end -- the GPU suite, smoke(), the default bench run
O=gpurun_out/r05; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/last_tests.log 2>&1; echo "tests rc $?"; grep -n "passed\|failed" $O/last_tests.log | tail -2
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
python bench.py > $O/last_bench.json 2> $O/last_bench.err; echo "bench rc $?"; wc -c $O/last_bench.json; cp gpurun_out/bench_full.json $O/last_bench_full.json
python -c "
import json
d=json.load(open('$O/last_bench.json'))
print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['value_end_to_end'], d['end_to_end']['ms_per_step'], d['end_to_end']['first_calls_ms_per_step'], d['cpu_baseline']['value'], d['parity_vs_cpu_sample'])
"
