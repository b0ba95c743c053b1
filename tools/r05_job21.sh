#!/bin/bash
# job 21: the default bench run with the host entry timed in its steady state
O=gpurun_out/r05; mkdir -p $O
python bench.py > $O/j21_bench.json 2> $O/j21_bench.err; echo "bench rc $?"; wc -c $O/j21_bench.json; cp gpurun_out/bench_full.json $O/j21_bench_full.json
python -c "
import json
d=json.load(open('$O/j21_bench.json'))
print(d['ms_per_step'], d['value'], d['value_end_to_end'], json.dumps(d['end_to_end']))
"
