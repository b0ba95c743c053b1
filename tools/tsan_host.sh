#!/bin/bash
# Host half of libpsi_gpu.so under ThreadSanitizer (round 5: the load campaigns' wrong answers that a new finder over
# the same index objects reproduced point at something made ONCE under load -- the host builders run OpenMP / std::thread
# loops, and 8 processes on one box oversubscribe the cores).  libgomp is not instrumented: its barriers are invisible to
# the tool, so reports whose both stacks sit in one `omp parallel` region on either side of a barrier are expected noise;
# OMP_WAIT_POLICY / ignore_noninstrumented_modules keep that down.  Needs a prior normal build for the .hip objects.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="${1:-/tmp/psi_tsan}"
mkdir -p "$OUT"
cd "$ROOT/psi_amd/csrc"
for f in graph index pathsel capi_host hits refio; do
  g++ -O1 -g -std=c++17 -fPIC -fopenmp -fsanitize=thread -fno-omit-frame-pointer -I../../include -c $f.cpp -o "$OUT/$f.o" &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o "$OUT/libpsi_gpu_tsan.so" "$OUT"/*.o build_gpu.o hits_gpu.o gather.o device.o \
  -lz -lgomp -lpthread -lhsa-runtime64 -ldl -L"$(dirname "$(g++ -print-file-name=libtsan.so)")" -ltsan
cd "$ROOT"
LD_PRELOAD="$(g++ -print-file-name=libtsan.so)" \
  TSAN_OPTIONS="halt_on_error=0:report_signal_unsafe=0:history_size=4:second_deadlock_stack=1:log_path=$OUT/tsan" \
  PSI_AMD_LIB="$OUT/libpsi_gpu_tsan.so" PSI_AMD_NO_TORCH=1 OMP_NUM_THREADS=${OMP_NUM_THREADS:-8} \
  python -m pytest tests/test_host.py -x -q -p no:cacheprovider ${TSAN_K:+-k "$TSAN_K"} 2>&1 | tail -3
ls "$OUT"/tsan.* 2>/dev/null | head; grep -h "SUMMARY" "$OUT"/tsan.* 2>/dev/null | sort | uniq -c | sort -rn | head -40
