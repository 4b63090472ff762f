#!/bin/bash
# attention rework check: kernel tests, timings new vs previous library
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "attention or flash or attn" 2>&1 | tail -5
echo NEW; timeout 300 python scripts/attn_bench.py 2>&1 | tail -6
if [ -f ddpm-torch_amd/csrc/libddpm_hip_prev.so ]; then
  cp ddpm-torch_amd/csrc/libddpm_hip_prev.so ddpm-torch_amd/csrc/libddpm_hip.so
  echo PREV; timeout 300 python scripts/attn_bench.py 2>&1 | tail -6
fi
