#!/bin/bash
# The round's last measurements (job 19 of 21 GPU jobs of round 5) -- full GPU suite, the default bench run, the host entry end to end, psikt, profiles of the two
# modes whose kernels changed after job 17
O=gpurun_out/r05; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/j19_tests.log 2>&1; echo "tests rc $?"; tail -6 $O/j19_tests.log | cut -c1-300
python bench.py > $O/j19_bench.json 2> $O/j19_bench.err; echo "bench rc $?"; wc -c $O/j19_bench.json; cp gpurun_out/bench_full.json $O/j19_bench_full.json
python bench.py --ordered --steps 20 --warmup 5 --lean --no-check > $O/j19_bench_ordered.json 2>/dev/null; python -c "import json;d=json.load(open('$O/j19_bench_ordered.json'));print('ordered',d['ms_per_step'])"
for m in traverse locus-table; do python bench.py --steps 10 --warmup 3 --lean --no-check --mode $m > $O/j19_bench_$m.json 2>/dev/null; python -c "import json;d=json.load(open('$O/j19_bench_$m.json'));print('$m',d['ms_per_step'],d['roofline']['kernel_ms_per_step'])"; done
E2E_TRACE=1 timeout 600 python tools/e2e_packed.py > $O/j19_e2e.jsonl 2> $O/j19_e2e.log; echo "e2e rc $?"; cat $O/j19_e2e.jsonl | cut -c1-330; grep "psigpu\]" $O/j19_e2e.log | head -10
timeout 600 python tools/psikt_config1.py > $O/j19_psikt_config1.json 2> $O/j19_psikt_config1.err; echo "psikt rc $?"; cat $O/j19_psikt_config1.json | cut -c1-600
timeout 600 python tools/psikt_config1.py --chunk 250000 > $O/j19_psikt_config1_chunks.json 2>> $O/j19_psikt_config1.err; cat $O/j19_psikt_config1_chunks.json | cut -c1-400
export PSI_PROFILE_ROUND=r05
bash tools/profile.sh t --mode traverse > gpurun_out/prof_t.log 2>&1; tail -2 gpurun_out/prof_t.log
bash tools/profile.sh k > gpurun_out/prof_k.log 2>&1; tail -2 gpurun_out/prof_k.log
du -sh gpurun_out
