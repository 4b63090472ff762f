#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -p no:cacheprovider -k "groupnorm" 2>&1 | tail -4
echo "== new"; timeout 300 python scripts/gn_bench.py 2>&1 | tail -12
echo "== old"; DDPM_GN_NO_LDS_BWD=1 DDPM_GN_NO_LDS_FWD=1 timeout 300 python scripts/gn_bench.py 2>&1 | tail -12
