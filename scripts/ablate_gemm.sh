#!/bin/bash
# Build ablated variants of the generic GEMM kernel (wrong results, timing only) and time the small-M conv sweep with each.
set -e
cd ddpm-torch_amd/csrc
cp libddpm_hip.so /tmp/lib_orig.so
for f in norm elementwise optim; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fPIC -c $f.hip -o /tmp/$f.o 2>/dev/null & done; wait
for v in BASE GABL_NOMFMA GABL_NOREAD GABL_NOISSUE_A GABL_NOISSUE_B "GABL_NOISSUE_A -DGABL_NOISSUE_B" "GABL_NOMFMA -DGABL_NOREAD" "GABL_NOMFMA -DGABL_NOREAD -DGABL_NOISSUE_A -DGABL_NOISSUE_B"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fPIC -D$v -c gemm.hip -o /tmp/gemm_abl.o 2>/dev/null
  hipcc --offload-arch=gfx950 -shared -fPIC -o libddpm_hip.so /tmp/gemm_abl.o /tmp/norm.o /tmp/elementwise.o /tmp/optim.o
  (cd ../../scripts && echo "== $v" && SWEEP_FAST=1 python smallm_sweep.py 2>&1)
done
cp /tmp/lib_orig.so libddpm_hip.so
