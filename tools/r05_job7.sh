#!/bin/bash
O=gpurun_out/r05; mkdir -p $O
# host entry A/B: records out on two engines vs one; 8 vs 16 widening threads (each a process: the threads are made once)
for v in "" "E2E_ONE_OUT_ENGINE=1" "E2E_WIDEN_THREADS=16" "E2E_WIDEN_THREADS=16 E2E_ONE_OUT_ENGINE=1" "E2E_WIDEN_THREADS=4"; do
  env $v E2E_QUICK=1 E2E_TRACE=1 timeout 300 python tools/e2e_packed.py > $O/j7_e2e.jsonl 2> $O/j7_e2e.log
  echo "== $v"; grep "default\|alternated" $O/j7_e2e.jsonl | cut -c1-260; grep "copy engines" $O/j7_e2e.log | head -1; grep "psigpu\]   *[0-9]*: " $O/j7_e2e.log | tail -7
done
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/j7_tests.log 2>&1; echo "tests rc $?"; tail -6 $O/j7_tests.log
