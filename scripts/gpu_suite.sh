#!/bin/bash
# the full GPU suite, N times (flakiness check)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-suite}; mkdir -p $O
for i in $(seq 1 ${2:-1}); do
  timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | grep "passed\|failed\|error" | tail -3 | tee -a $O/tests.txt
done
