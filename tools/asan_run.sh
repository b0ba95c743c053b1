#!/bin/bash
# run a command against the sanitized build of tools/asan_full.sh (python: the runtime has to be loaded first)
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
RT="$(g++ -print-file-name=libasan.so)"
export LD_PRELOAD="$RT"
export ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0:print_stacktrace=1:handle_segv=1:allow_user_segv_handler=0:detect_stack_use_after_return=0:${ASAN_EXTRA}"
export PSI_AMD_LIB="$ROOT/asan_build/libpsi_gpu_asan.so" PSI_ORACLE_LIB="$ROOT/asan_build/libpsi_oracle.so" PSI_AMD_NO_TORCH=1
exec "$@"
