#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/wg; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -p no:cacheprovider -k "wgrad" 2>&1 | tail -8
timeout 300 python scripts/wgrad_bench.py $WG_SPLITS 2>&1 | tail -16
