#!/bin/bash
# one test, several processes at once, several times: bash tools/flaky.sh '<pytest -k expr>' N_PROCS REPS [ENV=1 ...]
K="$1"; N=${2:-4}; R=${3:-3}; shift 3
for e in "$@"; do export "$e"; done
fails=0
for r in $(seq 1 $R); do
  pids=()
  for i in $(seq 1 $N); do
    ( timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "$K" > gpurun_out/flaky_${r}_$i.log 2>&1 ) &
    pids+=($!)
  done
  for p in "${pids[@]}"; do wait $p || fails=$((fails+1)); done
done
echo "env: $* -> $fails failing processes of $((N*R))"
grep -l "failed" gpurun_out/flaky_*.log 2>/dev/null | head -3
