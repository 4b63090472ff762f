#!/bin/bash
# GroupNorm rework check: kernel tests, isolated timings new vs previous library, training step
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-gn2}; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "groupnorm" 2>&1 | tail -5 > $O/pytest_gn.txt
timeout 300 python scripts/gn_bench.py > $O/gn_new.txt 2>&1
if [ -f ddpm-torch_amd/csrc/libddpm_hip_prev.so ]; then
  cp ddpm-torch_amd/csrc/libddpm_hip.so /tmp/new.so; cp ddpm-torch_amd/csrc/libddpm_hip_prev.so ddpm-torch_amd/csrc/libddpm_hip.so
  timeout 300 python scripts/gn_bench.py > $O/gn_prev.txt 2>&1
  cp /tmp/new.so ddpm-torch_amd/csrc/libddpm_hip.so
fi
cat $O/pytest_gn.txt; echo NEW; cat $O/gn_new.txt; echo PREV; cat $O/gn_prev.txt
