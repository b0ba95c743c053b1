#!/bin/bash
# The round's last measurements: the default bench run (the driver's command), the ordered step, the host entry end to end with
# engine copies against HIP stream copies, traverse mode on the two chr22-like stand-ins.  Output under gpurun_out/r06/.
O=gpurun_out/r06; mkdir -p $O
python bench.py > $O/bench_final.json 2> $O/bench_final.err; echo "bench rc $?"; wc -c $O/bench_final.json; cp gpurun_out/bench_full.json $O/bench_final_full.json
python bench.py --ordered --steps 20 --warmup 5 --lean --no-check > $O/bench_ordered.json 2>/dev/null; python -c "import json;d=json.load(open('$O/bench_ordered.json'));print('ordered',d['ms_per_step'], d['roofline']['avg_launch_ms'])"
E2E_QUICK=1 E2E_ENGINE_AB=1 E2E_TRACE=1 timeout 900 python tools/e2e_packed.py > $O/e2e.jsonl 2> $O/e2e.log; echo "e2e rc $?"; cat $O/e2e.jsonl | cut -c1-400; grep "psigpu\]" $O/e2e.log | head -12
for w in uniform clustered; do for m in traverse kmer-table; do timeout 600 python tools/standin_traverse.py $w $m 2> $O/standin_${w}_$m.err | tail -1 | tee -a $O/standin_traverse.jsonl; grep "prefix walks" $O/standin_${w}_$m.err | tail -1; done; done
du -sh gpurun_out
