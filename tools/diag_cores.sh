#!/bin/bash
# Where do the fuzz processes that share a GPU die?  N processes side by side, each in a directory of its own with
# core dumps on; half of them (every second one) directly under rocgdb, so that a fatal signal in ANY thread -- one
# that blocks signals, one without a usable stack -- ends in a backtrace of every thread.  Cores are opened with
# rocgdb afterwards and deleted.
# usage: bash tools/diag_cores.sh TAG FIRST N_PROCS SEEDS_EACH [ENV=1 ...]
TAG=$1; F=$2; N=$3; E=$4; shift 4
for e in "$@"; do export "$e"; done
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/diag_$TAG"
mkdir -p "$OUT"
echo "core_pattern: $(cat /proc/sys/kernel/core_pattern) uses_pid: $(cat /proc/sys/kernel/core_uses_pid)" > "$OUT/summary.txt"
ulimit -c unlimited
PY=$(readlink -f "$(which python3)")
pids=()
for i in $(seq 0 $((N-1))); do
  a=$((F + i*1000)); b=$((a + E))
  d="$OUT/p$i"; mkdir -p "$d"
  if [ $((i % 2)) = 1 ] && [ -z "$NO_GDB" ]; then
    ( cd "$d" && PSIGPU_SEGV_TRACE= FUZZ_TRACE=1 timeout ${FUZZ_TIMEOUT:-500} /opt/rocm/bin/rocgdb -q -batch \
        -ex "handle SIGUSR1 SIGUSR2 SIGPIPE SIGALRM nostop noprint pass" -ex run -ex "echo \n==== ALL THREADS ====\n" \
        -ex "thread apply all bt 24" -ex "info sharedlibrary psi" --args "$PY" "$ROOT/tools/fuzz_modes.py" $a $b > log.txt 2>&1 ) &
  else
    ( cd "$d" && FUZZ_TRACE=1 timeout ${FUZZ_TIMEOUT:-500} "$PY" -X faulthandler "$ROOT/tools/fuzz_modes.py" $a $b > log.txt 2>&1; echo "exit $?" >> log.txt ) &
  fi
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
for i in $(seq 0 $((N-1))); do
  d="$OUT/p$i"
  for c in "$d"/core*; do
    [ -f "$c" ] || continue
    echo "== p$i: core $(basename $c) $(stat -c %s $c) bytes" >> "$OUT/summary.txt"
    timeout 300 /opt/rocm/bin/rocgdb -q -batch -ex "thread apply all bt 24" "$PY" "$c" > "$d/core_bt.txt" 2>&1
    rm -f "$c"
  done
  {
    echo "== p$i: $(grep -c '^seed' $d/log.txt) seeds; last: $(grep '^seed\|^ok\|^exit\|MISMATCH' $d/log.txt | tail -n 2 | tr '\n' ' ')"
    grep -n "MISMATCH\|SIGSEGV\|SIGABRT\|SIGBUS\|received signal\|Fatal Python\|terminate\|double free\|corrupt" $d/log.txt | head -5
  } >> "$OUT/summary.txt"
done
cat "$OUT/summary.txt"
