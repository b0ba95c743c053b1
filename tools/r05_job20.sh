#!/bin/bash
# job 20: the default bench run (the driver's command) after the fix of its routes section; the host entry with 16-Mi sub-batches
O=gpurun_out/r05; mkdir -p $O
python bench.py > $O/j20_bench.json 2> $O/j20_bench.err; echo "bench rc $?"; wc -c $O/j20_bench.json; cp gpurun_out/bench_full.json $O/j20_bench_full.json; tail -3 $O/j20_bench.err | cut -c1-300
cat $O/j20_bench.json | cut -c1-1500
E2E_QUICK=1 timeout 300 python tools/e2e_packed.py 2>/dev/null | cut -c1-400
for sb in 16 32; do echo "sub $sb"; PSIGPU_SUB_BYTES=$((sb<<20)) E2E_QUICK=1 timeout 300 python tools/e2e_packed.py 2>/dev/null | grep "default\|wire formats" | cut -c1-300; done
( time timeout 900 python -m pytest tests -m gpu -x -q -k "wire or host_entry or two_sub or oversubscribed or packed or bench" ) > $O/j20_tests.log 2>&1; echo "tests rc $?"; tail -3 $O/j20_tests.log | cut -c1-300
