#!/bin/bash
# round 5: deterministic clip norm (per-block slots + fixed-order finish) — the two-ranks-on-one-GPU tests first, then the full GPU suite, smoke
# and the driver's bench command on the same tree
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_ddp_one_gpu.py -q > gpurun_out/r5h_one_gpu.txt 2>&1; tail -4 gpurun_out/r5h_one_gpu.txt
timeout 1500 python -m pytest tests -q -m gpu --deselect tests/test_ddp_one_gpu.py > gpurun_out/r5h_tests.txt 2>&1
grep -n "passed\|failed" gpurun_out/r5h_tests.txt | tail -3; grep -n "^FAILED\|^ERROR" gpurun_out/r5h_tests.txt | head
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -2
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r5h_bench_driver.json 2> gpurun_out/r5h_bench_driver.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5h_bench_driver.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["config"].get("step_probe"), d["roofline"]["frac"], d["sampling"]["ms_per_step"])
PY
