cd $GRAFT_REPO_ROOT
D=/dev/shm/pk; mkdir -p $D
python3 - <<PY
import sys; sys.path.insert(0,'tools'); sys.path.insert(0,'.')
import psikt_config1 as m
from psi_amd import synth
sg = synth.snv_graph(51_000_000, 1_100_000, n_block=11_000_000, seed=11)
m.write_gfa(sg, '$D/graph.gfa')
for a in range(0, 10_000_000, 2_000_000):
    b, o = synth.sim_reads_snv(sg, 2_000_000, 150, seed=13 + a // 2_000_000)
    m.write_fastq(b, o.astype('int64'), '$D/reads.fq', first=a, mode='wb' if a == 0 else 'ab')
PY
psi_amd/bin/psikt $D/graph.gfa -f $D/reads.fq -l 21 -o $D/out.gam -L $D/psi.log -c 1000000 -n 1 -I $D/ix > /dev/null 2>&1
for t in 4 8 16 32; do
  rm -f $D/psi.log
  PSIKT_WRITE_THREADS=$t psi_amd/bin/psikt $D/graph.gfa -f $D/reads.fq -l 21 -o $D/out.gam -L $D/psi.log -c 1000000 -n 1 -I $D/ix > /dev/null 2>&1
  echo "write threads $t: $(grep -o 'Found seed in [0-9.]* s' $D/psi.log) | $(grep -o 'Seed loop breakdown.*' $D/psi.log)"
done
for t in 8 16; do
  rm -f $D/psi.log
  PSIKT_WRITE_THREADS=$t psi_amd/bin/psikt $D/graph.gfa -f $D/reads.fq -l 21 -o /tmp/pk_out.gam -L $D/psi.log -c 1000000 -n 1 -I $D/ix > /dev/null 2>&1
  echo "out on /tmp, write threads $t: $(grep -o 'Found seed in [0-9.]* s' $D/psi.log) | $(grep -o 'Seed loop breakdown.*' $D/psi.log)"; df -T /tmp | tail -1
done
rm -rf $D /tmp/pk_out.gam
