#!/bin/bash
O=gpurun_out/r05; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "one_kernel_step or two_chunks_in_flight" > $O/j4_tests.log 2>&1; echo "tests rc $?"; tail -3 $O/j4_tests.log
for v in "" "--ordered"; do
  python bench.py --steps 20 --warmup 5 --lean --no-check $v > $O/j4_bench_lean$v.json 2>/dev/null
  python -c "import json;d=json.load(open('$O/j4_bench_lean$v.json'));print('lean $v',d['ms_per_step'],d['roofline']['frac'],d['roofline']['kernel_ms_per_step'])"
done
python bench.py --steps 20 --warmup 5 > $O/j4_bench.json 2> $O/j4_bench.err; echo "bench rc $?"
python -c "import json;d=json.load(open('$O/j4_bench.json'));print('full',d['ms_per_step'],d['roofline'],d['end_to_end'],d['parity_vs_cpu_sample'],d['two_chunks_in_flight_ms_per_step'])"
cp gpurun_out/bench_full.json $O/j4_bench_full.json
# the load campaign: default flags (arm A) and the loaders as they were (arm B), 16 processes on the one GPU
bash tools/r05_campaign.sh a "A B A B A A" 16 9000000
