#!/bin/bash
# the targeted loop (tools/repro_loop.py) in HALF of N processes, the general fuzz campaign in the others (the same kind of
# company the failing run had).  usage: TAG=x bash tools/repro_par.sh SEED K STEP NPATHS N SECONDS FUZZ_FIRST
SEED=$1; K=$2; STEP=$3; NP=$4; N=${5:-8}; S=${6:-150}; F=${7:-8570000}
mkdir -p gpurun_out
pids=()
for i in $(seq 0 $((N-1))); do
  if [ $((i % 2)) = 0 ]; then
    timeout $((S + 200)) python tools/repro_loop.py $SEED $K $STEP $NP $S > gpurun_out/repro_${TAG:-r}_p$i.log 2>&1 &
  else
    a=$((F + i*1000)); b=$((a + 400))
    FUZZ_TRACE=1 timeout $((S + 60)) python tools/fuzz_modes.py $a $b > gpurun_out/repro_${TAG:-r}_p$i.log 2>&1 &
  fi
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
for i in $(seq 0 $((N-1))); do echo "== p$i"; grep -v "^seed\|amdgpu.ids" gpurun_out/repro_${TAG:-r}_p$i.log | cut -c1-900 | tail -n 14; grep "^seed" gpurun_out/repro_${TAG:-r}_p$i.log | tail -n 1; done
