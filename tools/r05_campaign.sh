#!/bin/bash
# Load campaign of round 5: ROUNDS rounds of NPROC fuzz processes sharing the one GPU, each until its
# FUZZ_TIMEOUT; arms alternate as ARMS says ("A" = the library as built; "B" = PSIGPU_AB_LOAD_HOLE=1, the loaders as they were
# before round 5: pads filled on the null stream with nobody waiting, no fence at the end of a loader, no checksum read-back).
# usage: bash tools/r05_campaign.sh TAG "A A B A ..." NPROC FIRST_SEED
TAG=${1:-c}; ARMS=${2:-"A B"}; N=${3:-16}; F=${4:-9000000}
O=gpurun_out/r05/camp_$TAG; mkdir -p $O
r=0
for arm in $ARMS; do
  r=$((r+1)); first=$((F + r*100000))
  # (arm C, for the next campaign: arm B with the buffers of the engine copies freed at once, as until the end of round 5)
  unset PSIGPU_AB_LOAD_HOLE PSIGPU_AB_EARLY_FREE
  if [ "$arm" = "B" ]; then export PSIGPU_AB_LOAD_HOLE=1; fi
  if [ "$arm" = "C" ]; then export PSIGPU_AB_LOAD_HOLE=1 PSIGPU_AB_EARLY_FREE=1; fi
  TAG=${TAG}_r${r}${arm} FUZZ_TIMEOUT=${FUZZ_TIMEOUT:-400} bash tools/fuzz_par.sh $first $N 100000 > $O/round_${r}${arm}.out 2>&1
  mkdir -p $O/logs; mv gpurun_out/fuzz_${TAG}_r${r}${arm}_p*.log $O/logs/ 2>/dev/null
  ok=0; bad=0; sig=0; seeds=0
  for f in $O/logs/fuzz_${TAG}_r${r}${arm}_p*.log; do
    if grep -q "MISMATCH\|HOST CANARY CHANGED\|LABELS CHANGED" $f; then bad=$((bad+1)); fi
    if grep -q "fatal signal\|Segmentation\|core dumped\|double free\|Aborted" $f; then sig=$((sig+1)); fi
    if grep -q "Traceback" $f; then bad=$((bad+1)); fi
    s=$(grep -c "^seed" $f); seeds=$((seeds+s))
  done
  # processes that ended by themselves with a signal leave no line: count logs whose last line is a "seed" line and whose process was not the timeout's
  codes=$(grep "^EXIT" $O/round_${r}${arm}.out | awk '{print $3}' | sort | uniq -c | awk '{printf "%s x%s ", $2, $1}')
  echo "round $r arm $arm: procs $N seeds $seeds mismatch_or_error $bad signal_lines $sig exit_codes: $codes" | tee -a $O/summary.txt
  # keep only the logs of processes with something to say (the rest: their last lines)
  for f in $O/logs/fuzz_${TAG}_r${r}${arm}_p*.log; do
    if grep -q "MISMATCH\|HOST CANARY CHANGED\|LABELS CHANGED\|fatal signal\|Segmentation\|Traceback\|double free\|Aborted" $f; then grep -v "^seed" $f > $f.keep; fi
    tail -n 2 $f > $f.tail; rm -f $f
  done
done
cat $O/summary.txt
