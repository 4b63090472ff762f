#!/bin/bash
# conv3x3_pc_kernel: parity + bit-stability tests, then the sustained same-process A/B against the round-3 kernel (libddpm_hip_prev.so)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-pcchk}; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "conv3x3 or conv2d or contention" 2>&1 | grep -v amdgpu.ids | tail -6 | tee $O/tests.txt
timeout 300 python scripts/c3_ab.py ${2:-1000} 2>&1 | grep -v amdgpu.ids | cut -c1-140 | tee $O/ab.txt
