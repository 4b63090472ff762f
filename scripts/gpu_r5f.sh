#!/bin/bash
# round 5, call F: ramped all-reduce chunks + per-chunk time-projection gradients — full GPU suite, the one-rank exchange study again, bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r5f_tests.txt 2>&1
grep -n "passed\|failed" gpurun_out/r5f_tests.txt | tail -3; grep -n "FAILED\|Error" gpurun_out/r5f_tests.txt | head
DP_RESERVED=0 DP_STEPS=30 timeout 600 python scripts/dp_one_rank.py > gpurun_out/r5f_dp.txt 2>&1; grep -c workload gpurun_out/r5f_dp.txt
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r5f_bench.json 2> gpurun_out/r5f_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5f_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["config"].get("step_execution")[:40], d["config"].get("step_probe"), d["roofline"]["frac"], d["sampling"]["ms_per_step"])
PY
