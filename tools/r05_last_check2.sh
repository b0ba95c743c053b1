#!/bin/bash
# after the late frees of the engine copies' buffers: host-entry tests, smoke, the default bench run
O=gpurun_out/r05; mkdir -p $O
( time timeout 420 python -m pytest tests -m gpu -x -q -k "wire or host_entry or two_sub or oversubscribed or packed or device_entry or golden" ) > $O/last2_tests.log 2>&1; echo "tests rc $?"; grep -n "passed\|failed" $O/last2_tests.log | tail -2
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 200 python bench.py > $O/last2_bench.json 2> $O/last2_bench.err; echo "bench rc $?"; python -c "
import json
d=json.load(open('$O/last2_bench.json'))
print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['value_end_to_end'], d['end_to_end']['ms_per_step'], d['parity_vs_cpu_sample'])
"
