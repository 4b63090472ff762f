#!/bin/bash
# Build ablated variants of the conv kernels (wrong results, timing only) and time the hot conv shape with each.
set -e
cd ddpm-torch_amd/csrc
cp libddpm_hip.so /tmp/lib_orig.so
for f in norm elementwise optim; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fPIC -c $f.hip -o /tmp/$f.o 2>/dev/null & done; wait
for v in BASE ABL_NOVMWAIT ABL_NOBARRIER ABL_NOMFMA ABL_NODMA "ABL_NOVMWAIT -DABL_NOBARRIER" "ABL_NODMA -DABL_NOVMWAIT -DABL_NOBARRIER"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fPIC -D$v -c gemm.hip -o /tmp/gemm_abl.o 2>/dev/null
  hipcc --offload-arch=gfx950 -shared -fPIC -o libddpm_hip.so /tmp/gemm_abl.o /tmp/norm.o /tmp/elementwise.o /tmp/optim.o
  (cd ../.. && echo "== $v" && python scripts/microbench.py 2>&1 | grep -E "conv3x3_fwd_bfloat16_B128_(H32_C128_N128|H16_C256_N256|H16_C512)")
done
cp /tmp/lib_orig.so libddpm_hip.so
