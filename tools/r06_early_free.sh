#!/bin/bash
# tools/early_free_repro.hip, both arms: NPROC processes beside each other for SECS seconds each arm (the events needed a loaded box).
# usage: bash tools/r06_early_free.sh NPROC SECS
N=${1:-16}; S=${2:-360}
O=gpurun_out/r06/early_free; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/early_free_repro.hip -o /tmp/efr -lhsa-runtime64 || exit 1
for arm in early late; do
  late=0; [ $arm = late ] && late=1000000
  pids=()
  for i in $(seq $N); do timeout $((S + 60)) /tmp/efr 100000000 $late $S > $O/${arm}_p$i.log 2>&1 & pids+=($!); done
  bad=0; i=0
  for p in "${pids[@]}"; do wait $p; e=$?; i=$((i+1)); echo "EXIT $arm p$i $e" >> $O/exits.txt; done
  changed=$(cat $O/${arm}_p*.log | grep -c "changed at offset")
  iters=$(cat $O/${arm}_p*.log | grep "iterations" | awk '{s+=$1} END {print s}')
  echo "arm $arm: $N processes x $S s, iterations $iters, bait blocks changed: $changed; exit codes: $(grep "^EXIT $arm" $O/exits.txt | awk '{print $4}' | sort | uniq -c | awk '{printf "%s x%s ", $2, $1}')" | tee -a $O/summary.txt
  cat $O/${arm}_p*.log | grep "changed at offset" | head -n 40 > $O/${arm}_changed.txt
  rm -f $O/${arm}_p*.log
done
