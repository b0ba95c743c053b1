#!/bin/bash
# tools/fuzz_par.sh with EVERY process under rocgdb (same environment: PSIGPU_SEGV_TRACE=1): a fatal signal in any thread
# -- also one that blocks signals or has no usable stack, where no in-process handler gets to report -- stops the
# process in the debugger, which prints every thread's backtrace.
# usage: bash tools/gdb_par.sh FIRST N_PROCS SEEDS_EACH
F=${1:-500000}; N=${2:-8}; E=${3:-30}
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
PY=$(readlink -f "$(which python3)")
mkdir -p "$ROOT/gpurun_out"
pids=()
for i in $(seq 0 $((N-1))); do
  a=$((F + i*1000)); b=$((a + E))
  ( PSIGPU_SEGV_TRACE=1 FUZZ_TRACE=1 timeout -s INT -k 60 ${FUZZ_TIMEOUT:-400} /opt/rocm/bin/rocgdb -q -batch \
      -ex "handle SIGUSR1 SIGUSR2 SIGPIPE SIGALRM nostop noprint pass" -ex run -ex "echo \n==== ALL THREADS ====\n" \
      -ex "thread apply all bt 30" --args "$PY" "$ROOT/tools/fuzz_modes.py" $a $b > "$ROOT/gpurun_out/gdb_${TAG:-r}_p$i.log" 2>&1 ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
for i in $(seq 0 $((N-1))); do
  f="$ROOT/gpurun_out/gdb_${TAG:-r}_p$i.log"
  echo "== p$i: $(grep -c '^seed' $f) seeds; $(grep -c 'received signal' $f) fatal signals; $(grep -c MISMATCH $f) mismatches"
  grep -n "received signal\|MISMATCH\|again:" $f | head -4
  # keep the log small: thread churn lines out
  grep -v "^\[New Thread\|^\[Thread .* exited\]" $f > $f.tmp && mv $f.tmp $f
done
