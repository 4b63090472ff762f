#!/bin/bash
# timeline of conv3x3_pc_kernel: swaps in the instrumented library (built in the container: see scripts/build_variant.sh) on the scratch copy
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-pctl}; mkdir -p $O
L=ddpm-torch_amd/csrc
cp $L/libddpm_hip.so /tmp/prod.so
cp $L/libddpm_hip_timing.so $L/libddpm_hip.so
timeout 300 python scripts/pc_timeline.py 2>&1 | grep -v amdgpu.ids | tee $O/timeline.txt
ZERO_DATA=1 timeout 300 python scripts/pc_timeline.py 2>&1 | grep -v amdgpu.ids | tee $O/timeline_zero.txt
cp /tmp/prod.so $L/libddpm_hip.so
