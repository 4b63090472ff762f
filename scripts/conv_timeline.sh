#!/bin/bash
# Conv-forward phase timeline.  Build the instrumented library first (where hipcc + the objects are):
#   cd ddpm-torch_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fPIC -DHALO_TIMING -c gemm.hip -o /tmp/gemm_t.o &&
#   hipcc --offload-arch=gfx950 -shared -fPIC -o libddpm_hip_timing.so /tmp/gemm_t.o wgrad.o attention.o norm.o elementwise.o optim.o
# then run this on the GPU box (its tree is a scratch copy: the product library is overwritten there only).
cd "$GRAFT_REPO_ROOT/ddpm-torch_amd/csrc" || exit 1
cp libddpm_hip_timing.so libddpm_hip.so
cd ../.. && timeout 300 python ${1:-scripts/conv_timeline.py}
