for i in 1 2; do
  PSI_AMD_LIB=$PWD/psi_amd/bin/libpsi_gpu_prev.so python bench.py --lean --no-check --steps 100 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('prev', d['ms_per_step'], d['roofline']['kernel_ms_per_step'])"
  python bench.py --lean --no-check --steps 100 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new ', d['ms_per_step'], d['roofline']['kernel_ms_per_step'])"
done
