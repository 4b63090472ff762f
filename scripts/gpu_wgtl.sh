#!/bin/bash
# wgrad3x3: stage timeline (timing build: scripts/build_variant.sh wgt "-DWG_TIMING" wgrad.hip), parity of the product build, sustained rate, LDS read-rate probe
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-wgtl}; mkdir -p $O
{ timeout 200 python scripts/wg_timeline.py wgt 2>&1 | grep -v amdgpu.ids
  timeout 200 python scripts/wgrad_sustained.py 2>&1 | grep -v amdgpu.ids | tail -6
  timeout 60 scripts/probes/lds_read_rate; } | tee $O/timeline.txt
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "wgrad" 2>&1 | tail -3
