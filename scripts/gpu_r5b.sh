#!/bin/bash
# round 5, call B: full GPU suite on the plan-recordable engine; pc-kernel A/B (time-bounded waits); host time per form after the trims;
# one-rank data-parallel study (reserved CUs, stand-in copy kernels, CelebA-HQ trace); the driver's bench command
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -x -q -m gpu > gpurun_out/r5b_tests.txt 2>&1
tail -4 gpurun_out/r5b_tests.txt
timeout 300 python scripts/c3_ab.py 400 > gpurun_out/r5b_c3ab.txt 2>&1; grep -v amdgpu gpurun_out/r5b_c3ab.txt | cut -c1-120
timeout 300 python scripts/host_overhead.py > gpurun_out/r5b_host.txt 2>&1; grep "B=" gpurun_out/r5b_host.txt
timeout 600 python scripts/dp_one_rank.py > gpurun_out/r5b_dp.txt 2>&1; grep -c workload gpurun_out/r5b_dp.txt; tail -2 gpurun_out/r5b_dp.txt | cut -c1-600
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r5b_bench.json 2> gpurun_out/r5b_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5b_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["config"].get("step_execution")[:40], d["config"].get("step_probe"), d["roofline"]["frac"], d["sampling"]["ms_per_step"])
PY
