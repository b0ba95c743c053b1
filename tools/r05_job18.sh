#!/bin/bash
# job 18: build tests + traverse-mode parity with the traverser's derived read ids; then load campaign c (arms B B A B B)
O=gpurun_out/r05; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_build.py -m gpu -x -q ) > $O/j18_build_tests.log 2>&1; echo "build tests rc $?"; tail -3 $O/j18_build_tests.log | cut -c1-300
( time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "traverse or cap1" ) > $O/j18_trav_tests.log 2>&1; echo "traverse tests rc $?"; tail -3 $O/j18_trav_tests.log | cut -c1-300
python bench.py --steps 10 --warmup 3 --lean --no-check --mode traverse > $O/j18_bench_traverse.json 2> $O/j18_bench_traverse.err; python -c "import json;d=json.load(open('$O/j18_bench_traverse.json'));print('traverse',d['ms_per_step'],d['roofline']['kernel_ms_per_step'])" || tail -5 $O/j18_bench_traverse.err
FUZZ_TIMEOUT=400 bash tools/r05_campaign.sh c "B B A B B" 16 11000000 2>&1 | tail -8
du -sh gpurun_out
