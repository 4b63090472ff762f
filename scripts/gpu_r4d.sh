#!/bin/bash
# round 4, call D: the 1x1 kernel's epilogue modes and ablations, GroupNorm backward with "+=", the re-ordered skip backward in the step
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-r4d}; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_configs_gpu.py tests/test_unet_gpu.py -x -q 2>&1 | grep -v amdgpu.ids | tail -8 | tee $O/tests.txt
timeout 300 python scripts/pw_ab.py 200 2>&1 | grep -v amdgpu.ids | tee $O/pw_ab.txt
for acc in 0 1; do echo "== GN_ACC=$acc"; GN_ONLY=6 GN_ACC=$acc timeout 200 python scripts/gn_bench.py 2>&1 | grep -v amdgpu.ids; done | tee $O/gn_acc.txt
export BENCH_NO_SWEEP=1
for first in 1 0 1 0; do
  DDPM_SKIP_DGRAD_FIRST=$first timeout 300 python bench.py --steps 40 --warmup 10 --sample-steps 0 --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('skip_dgrad_first=$first', d['ms_per_step'], 'ms/step', d['value'])"
done | tee $O/ab.txt
