#!/bin/bash
# full GPU check: test suite + bench (eager direct step) ; output under gpurun_out/$1
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-full}; mkdir -p $O
export TMPDIR=/tmp BENCH_NO_SWEEP=1
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -15 > $O/tests.log
tail -4 $O/tests.log
DDPM_TORCH_AMD_TRAIN_GRAPH=0 BENCH_SHAPES=$O/shapes_iso.txt timeout 600 python bench.py --steps 40 --warmup 6 --sample-steps 200 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d=json.load(open("$O/bench.json"))
print("train", d["value"], "img/s", d["ms_per_step"], "ms/step; sampling", d["sampling"]["ms_per_step"], "ms/step")
for tag in ("per_kernel",):
    for k,v in d["roofline"][tag].items(): print("   prod", k, v["launches"], v["ms"], v["tflops"])
for k,v in d["roofline"]["isolated"]["per_kernel"].items(): print("   iso ", k, v["launches"], v["ms"], v["tflops"])
PY
