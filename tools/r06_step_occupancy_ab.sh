R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do
for lib in libpsi_gpu.so libpsi_gpu_v1.so libpsi_gpu_v2.so; do
  for a in "" "--ordered"; do
    PSI_AMD_LIB=$R/psi_amd/$lib python3 $R/bench.py --lean --steps 50 --warmup 10 $a 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$lib [$a] step %.4f ms  kernel %.4f ms' % (j['ms_per_step'], j.get('roofline',{}).get('avg_launch_ms') or 0))"
  done
done
done
