#!/bin/bash
# timing-only ablations of conv3x3_pc_kernel (results are WRONG by construction): which part of the step is the time?
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-pcabl}; mkdir -p $O
L=ddpm-torch_amd/csrc
cp $L/libddpm_hip.so /tmp/prod.so
for v in ${2:-timing abl_nodma abl_nomfma abl_noreads abl_noreads_nodma}; do
  cp $L/libddpm_hip_$v.so $L/libddpm_hip.so
  echo "=== $v" | tee -a $O/abl.txt
  timeout 300 python scripts/pc_timeline.py 2>&1 | grep -v amdgpu.ids | grep -v "block 0" | tee -a $O/abl.txt
done
cp /tmp/prod.so $L/libddpm_hip.so
