#!/bin/bash
# round 4, call E: the 1x1 kernel's 128 px x 256 ch tiles — parity, isolated A/B (DDPM_PW_NO_WIDE), step A/B
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-r4e}; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "conv1x1 or splitk or gemm" 2>&1 | grep -v amdgpu.ids | tail -8 | tee $O/tests.txt
for nw in 0 1; do
  echo "== DDPM_PW_NO_WIDE=$nw"
  if [ $nw = 1 ]; then export DDPM_PW_NO_WIDE=1; else unset DDPM_PW_NO_WIDE; fi
  timeout 300 python scripts/pw_ab.py 200 2>&1 | grep -v amdgpu.ids
done | tee $O/pw_ab.txt
export BENCH_NO_SWEEP=1
for nw in 0 1 0 1; do
  if [ $nw = 1 ]; then export DDPM_PW_NO_WIDE=1; else unset DDPM_PW_NO_WIDE; fi
  timeout 300 python bench.py --steps 40 --warmup 10 --sample-steps 0 --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('no_wide=$nw', d['ms_per_step'], 'ms/step', d['value'])"
done | tee $O/ab.txt
