#!/bin/bash
# locus-table mode (and traverse mode's FM route) with and without the interval table that carries the first row's record
for cfg in "locus-table 0 0" "locus-table 0 1" "traverse 8 0" "traverse 8 1"; do
  set -- $cfg
  if [ "$3" = "1" ]; then export PSIGPU_NO_FTABX=1; else unset PSIGPU_NO_FTABX; fi
  for rep in 1 2; do
    timeout 300 python bench.py --lean --mode $1 --tune $2 --steps 60 --warmup 5 2>/dev/null | tail -1 > gpurun_out/fx.json
    python - <<PY
import json
d=json.load(open("gpurun_out/fx.json"))
print("$1 tune=$2 no_ftabx=$3", round(d["ms_per_step"],4), {k:round(v,3) for k,v in d["roofline"]["kernel_ms_per_step"].items()})
PY
  done
done
