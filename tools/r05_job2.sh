#!/bin/bash
# round 5, second GPU job: the one-kernel step -- parity, bench, kernel stats
O=gpurun_out/r05; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "one_kernel_step or two_chunks_in_flight or golden or host_entry or packed_reads or random_graphs or gocc or uniform" > $O/j2_tests.log 2>&1; echo "tests rc $?"
tail -4 $O/j2_tests.log
python bench.py --steps 20 --warmup 5 > $O/j2_bench.json 2> $O/j2_bench.err; echo "bench rc $?"; cat $O/j2_bench.json
cp gpurun_out/bench_full.json $O/j2_bench_full.json 2>/dev/null
PSIGPU_NO_FUSED=1 python bench.py --steps 20 --warmup 5 --lean > $O/j2_bench_unfused.json 2> $O/j2_bench_unfused.err; echo "bench unfused rc $?"; cat $O/j2_bench_unfused.json
python bench.py --steps 20 --warmup 5 --lean --general-reads > $O/j2_bench_general.json 2> /dev/null; cat $O/j2_bench_general.json | cut -c1-400
bash tools/profile.sh r05k > $O/j2_profile.log 2>&1; echo "profile rc $?"; tail -30 $O/j2_profile.log
