#!/bin/bash
# timing-only ablations of conv3x3_pc_kernel (results are WRONG by construction): which part of the step is the time?
# Variants (scripts/build_variant.sh) are selected with DDPM_HIP_LIB: the product library is never overwritten.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-pcabl}; mkdir -p $O
L=$PWD/ddpm-torch_amd/csrc
for v in ${2:-timing abl_nodma abl_nomfma abl_noreads abl_noreads_nodma}; do
  echo "=== $v" | tee -a $O/abl.txt
  DDPM_HIP_LIB=$L/libddpm_hip_$v.so timeout 300 python scripts/pc_timeline.py 2>&1 | grep -v amdgpu.ids | grep -v "block 0" | tee -a $O/abl.txt
done
