#!/bin/bash
# same-box sweep of ONE knob on the CelebA-HQ 256x256 B = 2 training step: gpu_hq_sweep.sh VAR v1 v2 ...
cd "$GRAFT_REPO_ROOT" || exit 1
VAR=$1; shift
for rep in 1 2; do for v in "$@"; do env $VAR=$v python scripts/hq_step.py 40 train 2>/dev/null | tail -1 | sed "s/^/$VAR=$v : /"; done; done
