#!/bin/bash
# GroupNorm kernels: parity tests (incl. the dropout mask), per-shape timings, then the phase timeline (instrumented library)
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -p no:cacheprovider -k "groupnorm or dropout or gn" 2>&1 | tail -4
timeout 300 python scripts/gn_bench.py 2>&1 | tail -14
bash scripts/gn_timeline.sh
