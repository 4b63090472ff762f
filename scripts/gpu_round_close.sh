#!/bin/bash
# Round-closing profile set on ONE box: gpu_profile.sh (kernel stats, HBM counters, MFMA-busy, default bench line), the driver's bench command,
# the sampling kernel stats, the step timeline (launch-plan form) and the per-shape table.   usage: gpu_round_close.sh TAG
cd "$GRAFT_REPO_ROOT" || exit 1
T=${1:-close}; O=gpurun_out/$T; mkdir -p $O
bash scripts/gpu_profile.sh $T > $O/profile.log 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
bash scripts/gpu_sample_profile.sh $T > $O/sample.log 2>&1
GRAPH=plan TL_FIRST=30 TL_LAST=40 TL_TAIL=30 bash scripts/gpu_timeline.sh ${T}_tl > $O/tl.log 2>&1
cp gpurun_out/${T}_tl/timeline.txt $O/step_timeline.txt
BENCH_NO_SWEEP=1 BENCH_SHAPES=$O/step_shapes.txt timeout 600 python bench.py --steps 60 --warmup 12 --sample-steps 0 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $O/bench_shapes.json
python - <<PY
import json
for f in ("bench_default.json", "bench_driver_cmd.json"):
    d = json.loads(open("$O/" + f).read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d["ms_per_step_blocks"], "sampling", d.get("sampling", {}).get("ms_per_step"), "roofline", d["roofline"]["kernel"][:30], d["roofline"]["frac"])
PY
