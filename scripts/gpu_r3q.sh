#!/bin/bash
# round-3 batch: HBM rate probe, GroupNorm tests, CelebA-HQ B=2 step with the LDS-free finishing kernels
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-r3q}; mkdir -p $O
timeout 120 scripts/probes/hbm_rate > $O/hbm_rate.txt 2>&1
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "gn or norm or group" 2>&1 | tail -3 > $O/pytest_gn.txt
timeout 300 python scripts/hq_step.py 12 train > $O/hq_train.txt 2>&1
cat $O/hbm_rate.txt; cat $O/pytest_gn.txt; tail -2 $O/hq_train.txt
