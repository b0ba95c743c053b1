#!/bin/bash
O=gpurun_out/r05; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "one_kernel_step or two_chunks_in_flight or (full_size and kmer-table and not cap1) or packed_reads or uniform" > $O/j3_tests.log 2>&1; echo "tests rc $?"
tail -4 $O/j3_tests.log
python bench.py --steps 20 --warmup 5 --lean > $O/j3_bench_lean.json 2> $O/j3_bench_lean.err; echo "bench rc $?"; cut -c1-1500 $O/j3_bench_lean.json
python bench.py --steps 20 --warmup 5 --lean --general-reads > $O/j3_bench_general.json 2> /dev/null; python -c "import json;d=json.load(open('$O/j3_bench_general.json'));print('general',d['ms_per_step'],d['roofline']['kernel_ms_per_step'])"
python bench.py --steps 20 --warmup 5 > $O/j3_bench.json 2> $O/j3_bench.err; echo "bench rc $?"; python -c "import json;d=json.load(open('$O/j3_bench.json'));print('full',d['ms_per_step'],d['roofline'],d['end_to_end'],d['parity_vs_cpu_sample'],d['two_chunks_in_flight_ms_per_step'])"
cp gpurun_out/bench_full.json $O/j3_bench_full.json
