#!/bin/bash
# same-process A/B (interleaved bursts of direct C-ABI launches: no Python wrapper in the loop) of libddpm_hip_prev.so vs libddpm_hip.so on the 3x3 shapes
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-c3ab}; mkdir -p $O
timeout 600 python scripts/c3_ab.py ${2:-2000} 2>&1 | grep -v amdgpu.ids | tee $O/ab.txt
