#!/bin/bash
# A/B of the one-kernel step (k_kmer_step): rounds per tile, with and without the look-back (timing only), against the
# three-kernel step; each line = bench.py --lean on the same box
O=gpurun_out/r05; mkdir -p $O
run() { tag=$1; shift; env "$@" python bench.py --steps 20 --warmup 5 --lean --no-check > $O/ab_$tag.json 2>/dev/null; python - <<PY
import json
d=json.load(open("$O/ab_$tag.json"))
print("$tag", "ms_per_step", d["ms_per_step"], d["roofline"]["kernel_ms_per_step"])
PY
}
run r4 X=1
run r4_nolookback PSIGPU_AB_NO_LOOKBACK=1
run r2 PSI_AMD_LIB=$PWD/psi_amd/libpsi_gpu_r2.so
run r8 PSI_AMD_LIB=$PWD/psi_amd/libpsi_gpu_r8.so
run r8_nolookback PSI_AMD_LIB=$PWD/psi_amd/libpsi_gpu_r8.so PSIGPU_AB_NO_LOOKBACK=1
run unfused PSIGPU_NO_FUSED=1
run r4_again X=1
