#!/bin/bash
# Conv-forward phase timeline.  Build the instrumented library first (where hipcc + the objects are): scripts/build_variant.sh timing "-DHALO_TIMING" gemm.hip
# then run this on the GPU box.  The variant is selected with DDPM_HIP_LIB: the product library is never overwritten.
cd "$GRAFT_REPO_ROOT" || exit 1
DDPM_HIP_LIB=$PWD/ddpm-torch_amd/csrc/libddpm_hip_timing.so timeout 300 python ${1:-scripts/conv_timeline.py}
