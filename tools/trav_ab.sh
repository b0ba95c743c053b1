for cfg in "0 0" "0 8" "1 0" "1 8"; do
  set -- $cfg
  if [ "$1" = "1" ]; then unset PSIGPU_OVERLAP; else export PSIGPU_OVERLAP=1; fi
  timeout 300 python bench.py --lean --mode traverse --tune $2 --steps 60 --warmup 5 2>/dev/null | tail -1 > gpurun_out/trav_ab_$1_$2.json
  python - <<PY
import json
d=json.load(open("gpurun_out/trav_ab_$1_$2.json"))
print("serial=$1 tune=$2", round(d["ms_per_step"],4), {k:round(v,3) for k,v in d["roofline"]["kernel_ms_per_step"].items()})
PY
done
