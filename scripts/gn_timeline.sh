#!/bin/bash
# GroupNorm phase timeline.  Build the instrumented library FIRST (where hipcc + the objects are):
#   cd ddpm-torch_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fPIC -DGN_TIMING -c norm.hip -o /tmp/norm_t.o &&
#   hipcc --offload-arch=gfx950 -shared -fPIC -o libddpm_hip_timing.so gemm.o wgrad.o attention.o /tmp/norm_t.o elementwise.o optim.o
# then run this on the GPU box (the box's tree is a scratch copy: the product library is overwritten there only).
cd "$GRAFT_REPO_ROOT/ddpm-torch_amd/csrc" || exit 1
cp libddpm_hip_timing.so libddpm_hip.so
cd ../.. && timeout 300 python scripts/gn_timeline.py
