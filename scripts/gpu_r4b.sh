#!/bin/bash
# conv3x3_pc_kernel: parity tests of the 3x3 paths, shape timings (pc / pc + priority / round-3 kernel), consumer timeline
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-r4b}; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "conv3x3 or conv2d" 2>&1 | tail -15 | tee $O/tests.txt
for cfg in "pc:" "pc_prio:DDPM_C3_PC_FLAGS=1" "old:DDPM_CONV_NO_PC=1"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  echo "== $name" | tee -a $O/shapes.txt
  env $envs timeout 300 python scripts/c3_bench.py 2>&1 | grep -v amdgpu.ids | tail -14 | tee -a $O/shapes.txt
done
L=ddpm-torch_amd/csrc
cp $L/libddpm_hip.so /tmp/prod.so
cp $L/libddpm_hip_timing.so $L/libddpm_hip.so
timeout 300 python scripts/pc_timeline.py 2>&1 | grep -v amdgpu.ids | tee $O/timeline.txt
cp /tmp/prod.so $L/libddpm_hip.so
