#!/bin/bash
# A/B of k_fm_sweep's constants (buckets per round, seeds per lane, staged blocks): K1 time of the two fm-lf series per library
# usage: bash tools/r06_sweep_ab.sh  (libraries psi_amd/libpsi_gpu_v*.so built with make DEFS=... LIBNAME=...)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for lib in libpsi_gpu.so libpsi_gpu_v1.so libpsi_gpu_v2.so libpsi_gpu_v3.so libpsi_gpu_v4.so; do
  [ -f $R/psi_amd/$lib ] || continue
  for t in "after_ftab:--mode locus-table --tune 3" "no_ftab:--mode locus-table --tune 3 --ftab -1" "after_ftab_tail0:--mode locus-table --tune 3 --sweep-tail 0"; do
    tag=${t%%:*}; a=${t#*:}
    PSI_AMD_LIB=$R/psi_amd/$lib python3 $R/bench.py --lean --steps 20 --warmup 5 $a 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$lib $tag step %.3f ms  K1 %.3f ms' % (j['ms_per_step'], j.get('roofline',{}).get('avg_launch_ms') or 0))"
  done
done
