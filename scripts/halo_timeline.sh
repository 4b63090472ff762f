#!/bin/bash
set -e
cd ddpm-torch_amd/csrc
cp libddpm_hip.so /tmp/lib_orig.so
for f in norm elementwise optim; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fPIC -c $f.hip -o /tmp/$f.o 2>/dev/null & done; wait
hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fPIC -DHALO_TIMING $EXTRA -c gemm.hip -o /tmp/gemm_t.o 2>/dev/null
hipcc --offload-arch=gfx950 -shared -fPIC -o libddpm_hip.so /tmp/gemm_t.o /tmp/norm.o /tmp/elementwise.o /tmp/optim.o
(cd ../.. && DDPM_CONV_HALO_MINC=64 python ${TIMELINE:-scripts/halo_timeline.py})
cp /tmp/lib_orig.so libddpm_hip.so
