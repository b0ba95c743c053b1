set -e
cd $GRAFT_REPO_ROOT
g++ -O3 -std=c++17 -Ipsi_amd/include -Iinclude tools/reader_bench.cpp -o /tmp/reader_bench -Lpsi_amd -lpsi_gpu -lz -lpthread -Wl,-rpath,$PWD/psi_amd
python3 - <<PY
import sys; sys.path.insert(0,'tools'); sys.path.insert(0,'.')
import numpy as np, psikt_config1 as m
rng=np.random.default_rng(1)
n=3_000_000
bases=np.frombuffer(b'ACGT',np.uint8)[rng.integers(0,4,size=n*150)]
off=np.arange(n+1,dtype=np.int64)*150
m.write_fastq(bases, off, '/dev/shm/rb.fq')
PY
for t in 4 8 16 32; do echo "threads $t"; PSI_READER_THREADS=$t /tmp/reader_bench /dev/shm/rb.fq 1000000 0 | tail -1; done
rm -f /dev/shm/rb.fq
