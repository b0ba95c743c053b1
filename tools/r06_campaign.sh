#!/bin/bash
# Load campaign of round 6: does handing the ends of the raw engine copies straight back to HIP (as until round 5)
# produce the wrong records / overwritten host memory of rounds 4-5?  ROUNDS of NPROC fuzz processes sharing the one GPU, each
# until FUZZ_TIMEOUT; arms as ARMS says:
#   A = the SHIPPED library, default flags (copy ends pooled, loaders fenced)
#   B = campaign build, PSIGPU_AB_LOAD_HOLE=1 (the loaders as before round 5), copy ends pooled
#   C = campaign build, PSIGPU_AB_LOAD_HOLE=1 PSIGPU_AB_EARLY_FREE=1 (copy ends freed at once: rounds 3-5 as they were)
# The A/B switches exist only in libpsi_gpu_campaign.so (make DEFS=-DPSIGPU_CAMPAIGN=1 LIBNAME=libpsi_gpu_campaign.so OBJDIR=/tmp/camp lib).
# usage: bash tools/r06_campaign.sh TAG "C B A ..." NPROC FIRST_SEED
TAG=${1:-c}; ARMS=${2:-"A B C"}; N=${3:-32}; F=${4:-12000000}
O=gpurun_out/r06/camp_$TAG; mkdir -p $O/logs
r=0
for arm in $ARMS; do
  r=$((r+1)); first=$((F + r*100000))
  unset PSIGPU_AB_LOAD_HOLE PSIGPU_AB_EARLY_FREE PSI_AMD_LIB
  if [ "$arm" = "B" ]; then export PSI_AMD_LIB=$PWD/psi_amd/libpsi_gpu_campaign.so PSIGPU_AB_LOAD_HOLE=1; fi
  if [ "$arm" = "C" ]; then export PSI_AMD_LIB=$PWD/psi_amd/libpsi_gpu_campaign.so PSIGPU_AB_LOAD_HOLE=1 PSIGPU_AB_EARLY_FREE=1; fi
  # (the hole in halves: P = only the pads left unwaited (fence and read-back in place), F = only the fence and the read-back missing)
  if [ "$arm" = "P" ]; then export PSI_AMD_LIB=$PWD/psi_amd/libpsi_gpu_campaign.so PSIGPU_AB_LOAD_HOLE=2; fi
  if [ "$arm" = "F" ]; then export PSI_AMD_LIB=$PWD/psi_amd/libpsi_gpu_campaign.so PSIGPU_AB_LOAD_HOLE=3; fi
  t0=$(date +%s)
  TAG=${TAG}_r${r}${arm} FUZZ_TIMEOUT=${FUZZ_TIMEOUT:-240} bash tools/fuzz_par.sh $first $N 100000 > $O/round_${r}${arm}.out 2>&1
  mv gpurun_out/fuzz_${TAG}_r${r}${arm}_p*.log $O/logs/ 2>/dev/null
  bad=0; sig=0; seeds=0
  for f in $O/logs/fuzz_${TAG}_r${r}${arm}_p*.log; do
    if grep -q "MISMATCH\|HOST CANARY CHANGED\|LABELS CHANGED\|Traceback" $f; then bad=$((bad+1)); fi
    if grep -q "fatal signal\|Segmentation\|core dumped\|double free\|Aborted" $f; then sig=$((sig+1)); fi
    s=$(grep -c "^seed" $f); seeds=$((seeds+s))
  done
  codes=$(grep "^EXIT" $O/round_${r}${arm}.out | awk '{print $3}' | sort | uniq -c | awk '{printf "%s x%s ", $2, $1}')
  echo "round $r arm $arm: procs $N seeds $seeds mismatch_or_error $bad signal_lines $sig wall $(( $(date +%s) - t0 )) s exit_codes: $codes" | tee -a $O/summary.txt
  for f in $O/logs/fuzz_${TAG}_r${r}${arm}_p*.log; do
    if grep -q "MISMATCH\|HOST CANARY CHANGED\|LABELS CHANGED\|fatal signal\|Segmentation\|Traceback\|double free\|Aborted" $f; then grep -v "^seed" $f | tail -n 200 > $f.keep; fi
    rm -f $f
  done
  rm -f $O/round_${r}${arm}.out
done
cat $O/summary.txt
