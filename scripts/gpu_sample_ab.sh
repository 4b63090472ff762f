#!/bin/bash
# sampling-step A/B of one env switch within ONE box: ms per p_sample step at B=128 (hipGraph replay), CIFAR UNet bf16
cd "$GRAFT_REPO_ROOT" || exit 1
for rep in 1 2; do for cfg in "$@"; do
  env $cfg timeout 300 python scripts/sample_only.py 2>&1 | tail -1 | sed "s/^/[$cfg] /"
done; done
