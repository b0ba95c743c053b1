#!/bin/bash
# rocprofv3 kernel stats + PMC passes of bench.py in every mode / series (tools/profile.sh), summaries for profiles/r06_*
export PSI_PROFILE_ROUND=r06
bash tools/profile.sh k > gpurun_out/prof_k.log 2>&1
bash tools/profile.sh l --mode locus-table > gpurun_out/prof_l.log 2>&1
bash tools/profile.sh t --mode traverse > gpurun_out/prof_t.log 2>&1
bash tools/profile.sh t2 --mode traverse --tune 8 --series fm > gpurun_out/prof_t2.log 2>&1
bash tools/profile.sh f1 --mode locus-table --tune 3 --series after_ftab > gpurun_out/prof_f1.log 2>&1
bash tools/profile.sh f2 --mode locus-table --tune 3 --ftab -1 --series no_ftab > gpurun_out/prof_f2.log 2>&1
bash tools/profile.sh f3 --mode locus-table --sa-rate 32 --series sa32 > gpurun_out/prof_f3.log 2>&1
for t in k l t t2 f1 f2 f3; do echo "== $t"; tail -3 gpurun_out/prof_$t.log | cut -c1-300; done
