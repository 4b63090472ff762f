#!/bin/bash
# Build variants of gemm.hip with different -D flags (timing experiments, results are wrong) and run a benchmark with each.
# VARIANTS="A B ..." (each a -D list without the leading -D), CMD="python scripts/..." (run from the repo root).
set -e
cd ddpm-torch_amd/csrc
cp libddpm_hip.so /tmp/lib_orig.so
for f in norm elementwise optim; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fPIC -c $f.hip -o /tmp/$f.o 2>/dev/null & done; wait
IFS=';' read -ra VS <<< "${VARIANTS:-BASE;GABL_NOMFMA;GABL_NOREAD;GABL_NOISSUE_A;GABL_NOISSUE_B;GABL_NOISSUE_A -DGABL_NOISSUE_B;GABL_NOMFMA -DGABL_NOREAD}"
for v in "${VS[@]}"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fPIC -D$v -c gemm.hip -o /tmp/gemm_abl.o 2>/dev/null
  hipcc --offload-arch=gfx950 -shared -fPIC -o libddpm_hip.so /tmp/gemm_abl.o /tmp/norm.o /tmp/elementwise.o /tmp/optim.o
  (cd ../.. && echo "== $v" && ${CMD:-python scripts/microbench.py} 2>&1 | grep -E "${FILTER:-wgrad_bfloat16_B128_(H32_C128|H16_C256_N256|H16_C512)}")
done
cp /tmp/lib_orig.so libddpm_hip.so
