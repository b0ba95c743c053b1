#!/bin/bash
# N fuzz processes side by side on one GPU, every one against the AddressSanitizer build (tools/asan_full.sh)
# usage: bash tools/asan_par.sh FIRST N_PROCS SEEDS_EACH
F=${1:-400000}; N=${2:-8}; E=${3:-25}
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
mkdir -p "$ROOT/gpurun_out"
pids=()
for i in $(seq 0 $((N-1))); do
  a=$((F + i*1000)); b=$((a + E))
  ( FUZZ_TRACE=1 timeout ${FUZZ_TIMEOUT:-600} bash "$ROOT/tools/asan_run.sh" python "$ROOT/tools/fuzz_modes.py" $a $b > "$ROOT/gpurun_out/asan_${TAG:-r}_p$i.log" 2>&1; echo "exit $?" >> "$ROOT/gpurun_out/asan_${TAG:-r}_p$i.log" ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
for i in $(seq 0 $((N-1))); do
  f="$ROOT/gpurun_out/asan_${TAG:-r}_p$i.log"
  echo "== p$i: $(grep -c '^seed' $f) seeds, $(tail -n 1 $f)"
  grep -n "AddressSanitizer\|MISMATCH\|SUMMARY\|again:" $f | head -5
done
