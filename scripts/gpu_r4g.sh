#!/bin/bash
# round 4, call G: full GPU suite, then the round's profile set (kernel-trace stats, HBM traffic, MFMA busy, default bench line)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-r4g}; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -8 | tee $O/tests.txt
bash scripts/gpu_profile.sh ${1:-r4g} 2>&1 | tail -45
