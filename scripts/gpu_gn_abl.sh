#!/bin/bash
# GroupNorm ablations (timing only): transcendentals / dropout hash compiled out
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-gnabl}; mkdir -p $O
L=ddpm-torch_amd/csrc
cp $L/libddpm_hip.so /tmp/prod.so
for v in prod gn_notrans gn_nohash gn_neither; do
  [ $v = prod ] && cp /tmp/prod.so $L/libddpm_hip.so || cp $L/libddpm_hip_$v.so $L/libddpm_hip.so
  echo "=== $v" | tee -a $O/abl.txt
  timeout 300 python scripts/gn_bench.py 2>&1 | grep -v amdgpu.ids | tee -a $O/abl.txt
done
cp /tmp/prod.so $L/libddpm_hip.so
