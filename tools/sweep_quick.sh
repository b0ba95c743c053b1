#!/bin/bash
# quick look at the level-synchronous FM search on the GPU box: parity test, then per-kernel times of the two fm-lf series
# usage: bash tools/sweep_quick.sh [notest]
R=${GRAFT_REPO_ROOT:-$(pwd)}
if [ "$1" != "notest" ]; then python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "level_synchronous" 2>&1 | tail -3; fi
cd /tmp && export TMPDIR=/tmp
for t in "f1:--mode locus-table --tune 3" "f2:--mode locus-table --tune 3 --ftab -1"; do
  tag=${t%%:*}; a=${t#*:}
  rm -rf $R/gpurun_out/sw_$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/sw_$tag -- python3 $R/bench.py --lean --steps 10 --warmup 2 $a > $R/gpurun_out/sw_$tag.log 2>&1
  tail -1 $R/gpurun_out/sw_$tag.log | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$tag ms_per_step', j['ms_per_step'], 'K1 ms', j.get('roofline',{}).get('avg_launch_ms'))"
  python3 - <<PY
import csv,glob
f=glob.glob('$R/gpurun_out/sw_$tag/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:40]:
    n=r['Name'].replace('(anonymous namespace)::','').split('(')[0]
    if any(x in n for x in ('sweep','scan','k_fm_','k_wave','k_seed_pack','k_lkt_probe')):
        print('  %-44s calls %5s avg %9.1f us'%(n[:44], r['Calls'], float(r['AverageNs'])/1e3))
PY
  find $R/gpurun_out/sw_$tag -name "*kernel_trace.csv" -delete; find $R/gpurun_out/sw_$tag -name "*.db" -delete
done
