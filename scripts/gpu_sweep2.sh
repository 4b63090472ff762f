#!/bin/bash
# like gpu_sweep.sh but WITHOUT forcing the eager step: bench (train only) under several env settings within ONE box
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/$1; shift; mkdir -p $O
export TMPDIR=/tmp BENCH_NO_SWEEP=1
i=0
for rep in 1 2; do
for cfg in "$@"; do
  i=$((i+1))
  env $cfg timeout 600 python bench.py --steps 80 --warmup 12 --sample-steps 0 --no-cpu-baseline --no-extras > $O/b_$i.json 2> $O/b_$i.err
  python - <<PY
import json
try:
    d=json.load(open("$O/b_$i.json")); print("[$cfg]", d["value"], "img/s", d["ms_per_step"], "ms/step", d["config"]["step_execution"])
except Exception as e: print("[$cfg] failed", e)
PY
done; done
