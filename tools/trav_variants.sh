#!/bin/bash
# traverse-mode step with experiment builds of the library (make LIBNAME=... DEFS=...): usage: bash tools/trav_variants.sh lib1.so lib2.so ...
for so in "$@"; do
  for rep in 1 2; do
    PSI_AMD_LIB=$PWD/psi_amd/$so timeout 300 python bench.py --lean --mode traverse --steps 60 --warmup 5 2>/dev/null | tail -1 > gpurun_out/tv.json
    python - <<PY
import json
d=json.load(open("gpurun_out/tv.json"))
print("$so", round(d["ms_per_step"],4), {k:round(v,3) for k,v in d["roofline"]["kernel_ms_per_step"].items()})
PY
  done
done
