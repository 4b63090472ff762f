#!/bin/bash
# timeline of conv3x3_pc_kernel from the instrumented library (scripts/build_variant.sh timing "-DC3_TIMING" conv3x3.hip), selected with DDPM_HIP_LIB
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-pctl}; mkdir -p $O
export DDPM_HIP_LIB=$PWD/ddpm-torch_amd/csrc/libddpm_hip_timing.so
timeout 300 python scripts/pc_timeline.py 2>&1 | grep -v amdgpu.ids | tee $O/timeline.txt
ZERO_DATA=1 timeout 300 python scripts/pc_timeline.py 2>&1 | grep -v amdgpu.ids | tee $O/timeline_zero.txt
