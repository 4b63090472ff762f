#!/bin/bash
# GroupNorm ablations (timing only; WRONG numerics by construction): transcendentals / dropout hash compiled out.
# Variants (scripts/build_variant.sh gn_notrans "-DGN_ABL_NO_TRANS" norm.hip ...) are selected with DDPM_HIP_LIB: the product library is never overwritten.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-gnabl}; mkdir -p $O
L=$PWD/ddpm-torch_amd/csrc
for v in prod gn_notrans gn_nohash gn_neither; do
  lib=$L/libddpm_hip_$v.so; [ $v = prod ] && lib=$L/libddpm_hip.so
  echo "=== $v" | tee -a $O/abl.txt
  DDPM_HIP_LIB=$lib timeout 300 python scripts/gn_bench.py 2>&1 | grep -v amdgpu.ids | tee -a $O/abl.txt
done
