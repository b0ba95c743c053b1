#!/bin/bash
# round 5, job 5: the whole GPU suite on the split build; the host entry's timeline; psikt's wall clock on configs[1]
O=gpurun_out/r05; mkdir -p $O
# the seed of campaign a's one disagreement, alone: with and without the loaders' fences
for hole in 0 1; do if [ $hole = 1 ]; then export PSIGPU_AB_LOAD_HOLE=1; else unset PSIGPU_AB_LOAD_HOLE; fi; timeout 300 python tools/fuzz_modes.py 9211004 9211005 > $O/j5_seed9211004_hole$hole.log 2>&1; echo "seed 9211004 alone, hole $hole: rc $?"; tail -2 $O/j5_seed9211004_hole$hole.log; done; unset PSIGPU_AB_LOAD_HOLE
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/j5_tests.log 2>&1; echo "tests rc $?"; tail -5 $O/j5_tests.log
for m in traverse locus-table; do python bench.py --steps 10 --warmup 3 --lean --no-check --mode $m > $O/j5_bench_$m.json 2>/dev/null; python -c "import json;d=json.load(open('$O/j5_bench_$m.json'));print('$m',d['ms_per_step'],d['roofline']['kernel_ms_per_step'])"; done
E2E_TRACE=1 timeout 600 python tools/e2e_packed.py > $O/j5_e2e_packed.jsonl 2> $O/j5_e2e_trace.log; echo "e2e rc $?"; cat $O/j5_e2e_packed.jsonl; grep "psigpu\]\|traced" $O/j5_e2e_trace.log | tail -40
timeout 600 python tools/psikt_config1.py > $O/j5_psikt_config1.json 2> $O/j5_psikt_config1.err; echo "psikt rc $?"; cat $O/j5_psikt_config1.json
timeout 600 python tools/psikt_config1.py --chunk 250000 > $O/j5_psikt_config1_chunks.json 2>> $O/j5_psikt_config1.err; cat $O/j5_psikt_config1_chunks.json
