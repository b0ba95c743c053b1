#!/bin/bash
# The WHOLE host side of libpsi_gpu.so -- the C++ objects AND the host half of the .hip files (the host entry's pipeline,
# the contexts, the table builders' drivers) -- and the oracle under AddressSanitizer.  The sanitizer RUNTIME is gcc's
# libasan: ROCm's own compiler-rt intercepts hsa_amd_memory_pool_allocate for GPU AddressSanitizer, which this pool does
# not offer (every device allocation then fails); the instrumentation clang emits for the host half of the .hip files
# speaks the same runtime interface (v8).  Device code is NOT instrumented (-fno-gpu-sanitize).
#   bash tools/asan_full.sh          builds asan_build/libpsi_gpu_asan.so + asan_build/libpsi_oracle.so   (no GPU needed)
#   on the GPU box:  bash tools/asan_full.sh && bash tools/asan_run.sh python tools/fuzz_modes.py A B   (asan_build/ is not shipped)
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/asan_build"
mkdir -p "$OUT"
rm -f "$OUT"/*.o
cd "$ROOT/psi_amd/csrc"
GSAN="-fsanitize=address -fno-omit-frame-pointer -g -O1"
for f in graph index pathsel capi_host hits refio; do
  g++ $GSAN -std=c++17 -fPIC -fopenmp -I../../include -c $f.cpp -o "$OUT/$f.o" &
done
CSAN="-fsanitize=address -fno-gpu-sanitize -fno-omit-frame-pointer -g -O1 -mllvm -asan-globals=0"
for f in device build_gpu hits_gpu; do
  /opt/rocm/bin/hipcc $CSAN -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -Wno-unused-function -c $f.hip -o "$OUT/$f.o" &
done
/opt/rocm/bin/hipcc $CSAN -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -c gather.cpp -o "$OUT/gather.o" &
wait
ASANLIB="$(g++ -print-file-name=libasan.so)"
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o "$OUT/libpsi_gpu_asan.so" "$OUT"/*.o \
  -lz -lgomp -lpthread -lhsa-runtime64 -ldl -L"$(dirname "$ASANLIB")" -lasan
gcc $GSAN -std=c11 -fPIC -fopenmp -shared -o "$OUT/libpsi_oracle.so" "$ROOT/oracle/psi_oracle.c"
ls -la "$OUT"/*.so
