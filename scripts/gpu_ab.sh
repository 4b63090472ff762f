#!/bin/bash
# A/B of one env switch within ONE box: bench (eager direct step, no sweep) with VAR=0 and VAR=1.   usage: gpu_ab.sh TAG VAR [steps]
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-ab}; VAR=$2; mkdir -p $O
export TMPDIR=/tmp BENCH_NO_SWEEP=1
for v in 1 0 1 0; do
  env $VAR=$v DDPM_TORCH_AMD_TRAIN_GRAPH=0 timeout 600 python bench.py --steps ${3:-60} --warmup 10 --sample-steps 100 --no-cpu-baseline --no-extras > $O/bench_$v.json 2> $O/bench_$v.err
  python - <<PY
import json
d=json.load(open("$O/bench_$v.json"))
print("$VAR=$v train", d["value"], "img/s", d["ms_per_step"], "ms/step; sampling", d["sampling"]["ms_per_step"], "ms/step")
PY
done
