#!/bin/bash
# job 16: wire tests again, then the round's profiles (tools/r05_profiles.sh)
O=gpurun_out/r05; mkdir -p $O
( time timeout 900 python -m pytest tests -m gpu -x -q -k "wire or host_entry or two_sub or oversubscribed or packed" ) > $O/j16_tests.log 2>&1; echo "tests rc $?"; tail -5 $O/j16_tests.log | cut -c1-300
bash tools/r05_profiles.sh 2>&1 | tail -30 | cut -c1-300
