#!/bin/bash
# round 5, first GPU job: the changed entry points, the pad experiment, one bench line
O=gpurun_out/r05; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "two_chunks_in_flight or resident_arrays or golden or smoke" > $O/j1_tests.log 2>&1; echo "tests rc $?" 
tail -3 $O/j1_tests.log
# does any answer depend on the pads behind the arrays?  every fresh allocation poisoned, pads left as allocated
PSIGPU_POISON=0xA5 PSIGPU_AB_NO_PAD_ZERO=1 timeout 400 python tools/fuzz_modes.py 700000 700400 > $O/j1_pad_nozero.log 2>&1; echo "pad-nozero rc $?"
grep -v "^seed" $O/j1_pad_nozero.log | tail -5
PSIGPU_POISON=0xA5 timeout 400 python tools/fuzz_modes.py 700000 700400 > $O/j1_pad_control.log 2>&1; echo "pad-control rc $?"
grep -v "^seed" $O/j1_pad_control.log | tail -3
python bench.py --steps 20 --warmup 5 > $O/j1_bench.json 2> $O/j1_bench.err; echo "bench rc $?"; wc -c $O/j1_bench.json; cat $O/j1_bench.json
cp gpurun_out/bench_full.json $O/j1_bench_full.json 2>/dev/null
