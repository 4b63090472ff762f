#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for abl in 0 1 2 8 9 10 25; do
  echo "== ablate $abl"; DDPM_WG_ABLATE=$abl timeout 120 python scripts/wgrad_bench.py 2>&1 | grep -E "32\^2 128->128|16\^2 256->256|32\^2 256->256" | sed 's/generic.*TF | s/ s/'
done
