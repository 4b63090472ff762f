#!/bin/bash
# round 5, closing call (2): the driver's GPU test command + smoke + the driver's bench command on the final tree
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r5n_tests.txt 2>&1
grep -n "passed\|failed" gpurun_out/r5n_tests.txt | tail -3; grep -n "^FAILED\|^ERROR" gpurun_out/r5n_tests.txt | head
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -2
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r5n_bench_driver.json 2> gpurun_out/r5n_bench_driver.err
echo "stdout lines: $(wc -l < gpurun_out/r5n_bench_driver.json)"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5n_bench_driver.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["config"].get("step_probe"), d["roofline"]["frac"], d["sampling"]["ms_per_step"])
PY
