#!/bin/bash
# side-stream weight-gradient kernels: how many CUs should a launch take?  (block budget -> K slices; fewer slices = fewer slab copies too)
#   usage: gpu_wcu.sh [ENVVAR] [values...]     default: DDPM_WGRAD3_CUS 256 128
cd "$GRAFT_REPO_ROOT" || exit 1
VAR=${1:-DDPM_WGRAD3_CUS}; shift; VALS=${@:-256 128}
CMD="python bench.py --steps 60 --warmup 15 --sample-steps 0 --no-cpu-baseline --no-extras"
run() { $CMD 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$1', d['ms_per_step'])"; }
for rep in 1 2 3; do
  for v in $VALS; do env $VAR=$v bash -c "$(declare -f run); CMD='$CMD'; run '$VAR=$v'"; done
done
