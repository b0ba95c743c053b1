#!/bin/bash
# Host half of libpsi_gpu.so (graph loading, path picking, index builders, starting loci, file formats,
# host sort-unique) rebuilt with AddressSanitizer + UndefinedBehaviorSanitizer and the CPU tests of
# tests/test_host.py run against it.  (GPU sanitizers are not available on the pool; the device half
# is linked in unsanitized and not exercised here.)  Needs a prior normal build for the .hip objects.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="${1:-/tmp/psi_asan}"
mkdir -p "$OUT"
cd "$ROOT/psi_amd/csrc"
for f in graph index pathsel capi_host hits refio; do
  g++ -O1 -g -std=c++17 -fPIC -fopenmp -fsanitize=address,undefined -fno-omit-frame-pointer -I../../include -c $f.cpp -o "$OUT/$f.o" &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o "$OUT/libpsi_gpu_asan.so" "$OUT"/*.o build_gpu.o hits_gpu.o gather.o device.o \
  -lz -lgomp -lpthread -lhsa-runtime64 -ldl -L"$(dirname "$(g++ -print-file-name=libasan.so)")" -lasan -lubsan
cd "$ROOT"
LD_PRELOAD="$(g++ -print-file-name=libasan.so) $(g++ -print-file-name=libubsan.so)" \
  ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1 \
  PSI_AMD_LIB="$OUT/libpsi_gpu_asan.so" PSI_AMD_NO_TORCH=1 \
  python -m pytest tests/test_host.py -x -q -s -p no:cacheprovider 2>&1 | tee "$OUT/log.txt" | tail -3
if grep -q "runtime error\|AddressSanitizer" "$OUT/log.txt"; then echo "SANITIZER REPORTS in $OUT/log.txt"; exit 1; fi
if grep -q " failed\| error" "$OUT/log.txt"; then echo "TESTS FAILED under the sanitizers: $OUT/log.txt"; exit 1; fi
echo "sanitizers: clean"
