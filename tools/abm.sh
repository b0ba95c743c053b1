# bench in the three query modes (and serial kernels for per-kernel times)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for M in kmer-table locus-table traverse; do for S in 0 1; do
if [ $S = 1 ]; then unset PSIGPU_OVERLAP; else export PSIGPU_OVERLAP=1; fi
python $R/bench.py --steps 10 --warmup 2 --cpu-reads 0 --mode $M "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_step']
print('%-12s serial=%s step %.3f  %.2f Gseeds/s  ' % ('$M','$S',d['ms_per_step'],d['value']/1e9) + ' '.join('%s %.3f' % (a.replace('k_',''),b) for a,b in k.items()))"
done; done
