#!/bin/bash
O=gpurun_out/r05; mkdir -p $O
R=$PWD
echo "== cap1 check, this build"; timeout 300 python tools/r05_cap1_check.py 2>&1 | grep -v amdgpu.ids
echo "== cap1 check, round 4's library"; PSI_AMD_LIB=$PWD/psi_amd/libpsi_gpu_r04.so timeout 300 python tools/r05_cap1_check.py 2>&1 | grep -v amdgpu.ids
echo "== numa"; cat /sys/class/drm/card*/device/numa_node 2>/dev/null | head -3; ls /sys/devices/system/node/ | head; python -c "import os;print('cpus allowed', len(os.sched_getaffinity(0)), sorted(os.sched_getaffinity(0))[:4], '...')"; cat /sys/devices/system/node/node*/cpulist 2>/dev/null | head -4
python bench.py --steps 10 --warmup 3 --lean --no-check --mode traverse > $O/j8_bench_traverse.json 2> $O/j8_bench_traverse.err; python -c "import json;d=json.load(open('$O/j8_bench_traverse.json'));print('traverse',d['ms_per_step'],d['roofline']['kernel_ms_per_step'])" || tail -5 $O/j8_bench_traverse.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/j8_prof_t -- python3 $R/bench.py --lean --steps 5 --warmup 2 --no-check --mode traverse > $R/$O/j8_prof_t.log 2>&1
cd $R
f=$(ls -t $O/j8_prof_t/*/*kernel_stats.csv | head -1); python - <<PY
import csv
rows=list(csv.reader(open("$f")))
for r in rows[:1]+[r for r in rows[1:] if int(r[1])>=5][:14]:
    print(r[0].replace('(anonymous namespace)::','')[:70], r[1:5])
PY
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "traverse and (golden or random_graphs or snv_graph or spill or hla)" > $O/j8_tests.log 2>&1; echo "tests rc $?"; tail -3 $O/j8_tests.log
