#!/bin/bash
# round 5, call C: the round's profile set on the launch-plan step (kernel-trace stats, HBM traffic, MFMA-busy, default bench), the fixed one-rank
# RCCL test, and sweeps of the side-stream knobs now that the host no longer paces the step
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_unet_gpu.py -x -q -m gpu -k "rccl" 2>&1 | grep -v "amdgpu\|^HIP\|^ROCm\|^Hostname\|^Librccl\|^RCCL" | tail -3
bash scripts/gpu_profile.sh r05 > gpurun_out/r5c_profile.log 2>&1
tail -5 gpurun_out/r5c_profile.log | cut -c1-300
bash scripts/gpu_wcu.sh DDPM_WGRAD3_CUS 96 128 160 2>&1 | tee gpurun_out/r5c_sweep_wgrad3_cus.txt
bash scripts/gpu_wcu.sh DDPM_SLAB_FLUSH_ROWS 3 6 12 2>&1 | tee gpurun_out/r5c_sweep_flush.txt
