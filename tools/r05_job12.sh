#!/bin/bash
O=gpurun_out/r05; mkdir -p $O
# the two arm-B events of campaigns a and b, alone on the GPU, in both arms
for sd in 10007003 9211004; do
  for arm in A B; do
    if [ "$arm" = "B" ]; then export PSIGPU_AB_LOAD_HOLE=1; else unset PSIGPU_AB_LOAD_HOLE; fi
    timeout 300 python tools/fuzz_modes.py $sd $((sd+1)) > $O/j12_replay_${sd}_$arm.log 2>&1; echo "replay $sd arm $arm rc $?"; grep -v "^seed" $O/j12_replay_${sd}_$arm.log | head -20
  done
done
unset PSIGPU_AB_LOAD_HOLE
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/j12_tests.log 2>&1; echo "tests rc $?"; tail -6 $O/j12_tests.log
for m in kmer-table traverse locus-table; do python bench.py --steps 10 --warmup 3 --lean --no-check --mode $m > $O/j12_bench_$m.json 2> $O/j12_bench_$m.err; python -c "import json;d=json.load(open('$O/j12_bench_$m.json'));print('$m',d['ms_per_step'],d['roofline']['kernel_ms_per_step'])" || tail -5 $O/j12_bench_$m.err; done
python bench.py --steps 10 --warmup 3 --lean --no-check --mode traverse --tune 8 > $O/j12_bench_traverse_fm.json 2>/dev/null; python -c "import json;d=json.load(open('$O/j12_bench_traverse_fm.json'));print('traverse fm route',d['ms_per_step'],d['roofline']['kernel_ms_per_step'])"
