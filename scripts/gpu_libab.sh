#!/bin/bash
# same-box A/B of the training step between csrc/libddpm_hip_prev.so and the working tree's library (alternating runs); optional pytest filter first.
# The variant is selected with DDPM_HIP_LIB (ddpm_torch/_hip.py): the product library is never overwritten.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-libab}; mkdir -p $O
[ -n "$2" ] && timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "$2" 2>&1 | tail -3 | tee $O/tests.txt
export BENCH_NO_SWEEP=1
L=$PWD/ddpm-torch_amd/csrc
for v in new prev new prev new prev; do
  lib=$L/libddpm_hip.so; [ $v = prev ] && lib=$L/libddpm_hip_prev.so
  DDPM_HIP_LIB=$lib timeout 300 python bench.py --steps 40 --warmup 10 --sample-steps 0 --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], 'ms/step', d['value'])"
done | tee $O/ab.txt
