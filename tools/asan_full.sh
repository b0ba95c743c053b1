#!/bin/bash
# The WHOLE host side of libpsi_gpu.so -- the C++ objects AND the host half of the .hip files (the host entry's pipeline,
# the contexts, the table builders' drivers) -- and the oracle under AddressSanitizer, all with ROCm's clang so that one
# sanitizer runtime serves the process (tools/asan_host.sh builds the C++ objects alone, with gcc, for the CPU tests).
# Device code is NOT instrumented (-fno-gpu-sanitize: GPU AddressSanitizer is not available on the pool); what this
# catches is what the load campaign points at: a host-side use-after-free / overflow in the code around the transfers.
#   bash tools/asan_full.sh          builds asan_build/libpsi_gpu_asan.so + asan_build/libpsi_oracle.so   (no GPU needed)
#   on the GPU box:  bash tools/asan_run.sh python tools/fuzz_modes.py A B
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/asan_build"
LLVM=/opt/rocm/lib/llvm
mkdir -p "$OUT"
cd "$ROOT/psi_amd/csrc"
SAN="-fsanitize=address -shared-libasan -fno-omit-frame-pointer -g -O1"
for f in graph index pathsel capi_host hits refio; do
  $LLVM/bin/clang++ $SAN -std=c++17 -fPIC -fopenmp -I../../include -c $f.cpp -o "$OUT/$f.o" &
done
for f in device build_gpu hits_gpu; do
  /opt/rocm/bin/hipcc $SAN -fno-gpu-sanitize -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -Wno-unused-function -c $f.hip -o "$OUT/$f.o" &
done
/opt/rocm/bin/hipcc $SAN -fno-gpu-sanitize -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -c gather.cpp -o "$OUT/gather.o" &
wait
/opt/rocm/bin/hipcc -shared -fPIC $SAN --offload-arch=gfx950 -o "$OUT/libpsi_gpu_asan.so" "$OUT"/*.o \
  -lz -L$LLVM/lib -lomp -lpthread -lhsa-runtime64 -ldl -Wl,-rpath,$LLVM/lib
$LLVM/bin/clang $SAN -std=c11 -fPIC -fopenmp -shared -o "$OUT/libpsi_oracle.so" "$ROOT/oracle/psi_oracle.c" -L$LLVM/lib -lomp -Wl,-rpath,$LLVM/lib
ls -la "$OUT"/*.so
