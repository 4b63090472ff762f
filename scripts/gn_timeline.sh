#!/bin/bash
# GroupNorm phase timeline.  Build the instrumented library FIRST: scripts/build_variant.sh timing "-DGN_TIMING" norm.hip
# then run this on the GPU box.  The variant is selected with DDPM_HIP_LIB: the product library is never overwritten.
cd "$GRAFT_REPO_ROOT" || exit 1
DDPM_HIP_LIB=$PWD/ddpm-torch_amd/csrc/libddpm_hip_timing.so timeout 300 python scripts/gn_timeline.py
