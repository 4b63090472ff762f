#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
python -m pytest tests/test_kernels_gpu.py -q -x -k "split_k or conv_fwd" 2>&1 | tail -3
python scripts/hq_step.py 12 train | tail -1
DDPM_SPLITK64_RUNS=2 python scripts/hq_step.py 12 train | tail -1
python scripts/hq_step.py 12 train | tail -1
DDPM_SPLITK64_RUNS=2 python scripts/hq_step.py 12 train | tail -1
python scripts/hq_step.py 20 sample | tail -1
DDPM_SPLITK64_RUNS=2 python scripts/hq_step.py 20 sample | tail -1
