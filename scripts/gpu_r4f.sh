#!/bin/bash
# round 4, call F: 1x1 kernel — (pixel, channel) tile pairs for the 8 x 8 level, 256-channel tiles for N = 384
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-r4f}; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q 2>&1 | grep -v amdgpu.ids | tail -8 | tee $O/tests.txt
for mode in "" "DDPM_PW_WIDE_MODE=0" "DDPM_PW_WIDE_MODE=2" "DDPM_POINTWISE_MIN_M=32768"; do
  echo "== $mode"
  env $mode timeout 300 python scripts/pw_ab.py 200 2>&1 | grep -v amdgpu.ids | cut -c1-120
done | tee $O/pw_ab.txt
export BENCH_NO_SWEEP=1
for mode in "A=1" "DDPM_POINTWISE_MIN_M=32768" "DDPM_PW_WIDE_MODE=2" "A=1" "DDPM_POINTWISE_MIN_M=32768" "DDPM_PW_WIDE_MODE=2"; do
  env $mode timeout 300 python bench.py --steps 40 --warmup 10 --sample-steps 0 --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$mode', d['ms_per_step'], 'ms/step', d['value'])"
done | tee $O/ab.txt
