#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
python -m pytest tests/test_kernels_gpu.py -q -x -k "gather_rows" 2>&1 | tail -2
python -m pytest tests/test_unet_gpu.py tests/test_configs_gpu.py tests/test_cli_gpu.py -q -x 2>&1 | grep -E "passed|failed|rror" | tail -3
for v in 1 0 1 0; do echo -n "DDPM_TIME_TABLE=$v: "; DDPM_TIME_TABLE=$v python scripts/sample_only.py 300 2>&1 | tail -1; done
