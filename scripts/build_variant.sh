#!/bin/bash
# build a variant library: build_variant.sh NAME "-DFLAG ..." file.hip [file2.hip ...]  ->  csrc/libddpm_hip_NAME.so (the named sources rebuilt with the flags, the other objects reused)
set -e
cd "$(dirname "$0")/../ddpm-torch_amd/csrc"
name=$1; flags=$2; shift 2
objs=""
for s in gemm wgrad wgrad1x1 attention pointwise conv3x3 edgeconv norm elementwise optim probe plan; do
  if [[ " $* " == *" $s.hip "* ]]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fPIC $flags -c $s.hip -o /tmp/${s}_$name.o
    objs="$objs /tmp/${s}_$name.o"
  else objs="$objs $s.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libddpm_hip_$name.so $objs
echo built libddpm_hip_$name.so
