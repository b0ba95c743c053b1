#!/bin/bash
# job 17: device index construction incl. the starting loci of patched paths by path steps; then the round's profiles
O=gpurun_out/r05; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_build.py -m gpu -x -q ) > $O/j17_build_tests.log 2>&1; echo "build tests rc $?"; tail -5 $O/j17_build_tests.log | cut -c1-300
# fuzz with the device build verified against the host inside the library (patched paths now go through the device loci)
PSIGPU_BUILD_VERIFY=1 timeout 240 python tools/fuzz_modes.py 12000000 12000040 > $O/j17_fuzz.log 2>&1; echo "fuzz rc $?"; tail -2 $O/j17_fuzz.log | cut -c1-300
bash tools/r05_profiles.sh 2>&1 | tail -12 | cut -c1-200
du -sh gpurun_out
