#!/bin/bash
# Build variants of gemm.hip with different -D flags (timing experiments) and run the small-M conv sweep with each.
set -e
cd ddpm-torch_amd/csrc
cp libddpm_hip.so /tmp/lib_orig.so
for f in norm elementwise optim; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fPIC -c $f.hip -o /tmp/$f.o 2>/dev/null & done; wait
for v in ${VARIANTS:-"DEEP_ISSUE_KC=0" "DEEP_ISSUE_KC=1" "DEEP_ISSUE_KC=2" "DEEP_ISSUE_KC=3"}; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fPIC -D$v -c gemm.hip -o /tmp/gemm_abl.o 2>/dev/null
  hipcc --offload-arch=gfx950 -shared -fPIC -o libddpm_hip.so /tmp/gemm_abl.o /tmp/norm.o /tmp/elementwise.o /tmp/optim.o
  (cd ../../scripts && echo "== $v" && SWEEP_FAST=1 python smallm_sweep.py 2>&1 | grep -v amdgpu.ids)
done
cp /tmp/lib_orig.so libddpm_hip.so
