#!/bin/bash
# several build_stress processes beside each other on one GPU
# usage: TAG=x bash tools/build_par.sh FIRST N_PROCS SECONDS
F=${1:-8560000}; N=${2:-8}; S=${3:-120}
mkdir -p gpurun_out
pids=()
for i in $(seq 0 $((N-1))); do
  a=$((F + i*1000)); b=$((a + 1000))
  timeout $((S + 120)) python tools/build_stress.py $a $b $S > gpurun_out/build_${TAG:-r}_p$i.log 2>&1 &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait $p || rc=1; done
for i in $(seq 0 $((N-1))); do echo "== p$i"; grep -v "amdgpu.ids" gpurun_out/build_${TAG:-r}_p$i.log | cut -c1-700 | tail -n 6; done
exit $rc
