#!/bin/bash
# several fuzz processes beside each other on one GPU (host-side races show under contention)
# usage: bash tools/fuzz_par.sh FIRST N_PROCS SEEDS_EACH
F=${1:-93000}; N=${2:-4}; E=${3:-40}
mkdir -p gpurun_out
pids=()
for i in $(seq 0 $((N-1))); do
  a=$((F + i*1000)); b=$((a + E))
  PSIGPU_BUILD_VERIFY=${PSIGPU_BUILD_VERIFY-1} PSIGPU_SEGV_TRACE=1 FUZZ_TRACE=1 timeout ${FUZZ_TIMEOUT:-400} python tools/fuzz_modes.py $a $b > gpurun_out/fuzz_${TAG:-r}_p$i.log 2>&1 &
  pids+=($!)
done
rc=0
# (exit codes: 0 finished its seeds, 124 ran until the timeout -- both fine; 1 a mismatch or an exception; 134 / 139 a fatal signal)
i=0
for p in "${pids[@]}"; do wait $p; e=$?; echo "EXIT p$i $e"; if [ $e -ne 0 ] && [ $e -ne 124 ]; then rc=1; fi; i=$((i+1)); done
for i in $(seq 0 $((N-1))); do echo "== p$i"; grep -v "^seed" gpurun_out/fuzz_${TAG:-r}_p$i.log | tail -n 70; grep "^seed" gpurun_out/fuzz_${TAG:-r}_p$i.log | tail -n 1; done
exit $rc
