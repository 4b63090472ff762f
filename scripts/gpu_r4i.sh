#!/bin/bash
# round 4, call I: vectorised weight pack + wave-slab gradient unpack — parity, then the step A/B against the previous library
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-r4i}; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_configs_gpu.py -x -q -k "pack or unpack or steps_that_move or captured_training" 2>&1 | grep -v amdgpu.ids | tail -6 | tee $O/tests.txt
export BENCH_NO_SWEEP=1
L=$PWD/ddpm-torch_amd/csrc
for v in new prev new prev; do
  lib=$L/libddpm_hip.so; [ $v = prev ] && lib=$L/libddpm_hip_prev.so
  DDPM_HIP_LIB=$lib timeout 300 python bench.py --steps 40 --warmup 10 --sample-steps 0 --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], 'ms/step', d['value'])"
done | tee $O/ab.txt
cd /tmp && rm -rf /tmp/kt2 && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt2 -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --sample-steps 0 --no-cpu-baseline --no-extras > /dev/null 2>&1
grep -h "pack_weight_multi\|wgrad_unpack\|mt_adam" $(find /tmp/kt2 -name "*kernel_stats.csv" | head -1) | cut -c1-200 | tee $GRAFT_REPO_ROOT/$O/tail_kernels.txt
