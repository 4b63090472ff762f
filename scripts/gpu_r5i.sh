#!/bin/bash
# round 5, closing call on the final tree: the driver's GPU test command, smoke, the round's profile set (kernel-trace stats, HBM traffic,
# MFMA-busy, default bench) and the driver's bench command
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r5i_tests.txt 2>&1
grep -n "passed\|failed" gpurun_out/r5i_tests.txt | tail -3; grep -n "^FAILED\|^ERROR" gpurun_out/r5i_tests.txt | head
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -2
bash scripts/gpu_profile.sh r05final2 > gpurun_out/r5i_profile.log 2>&1
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r5i_bench_driver.json 2> gpurun_out/r5i_bench_driver.err
python - <<'PY'
import json
for f in ("gpurun_out/r05final2/bench_default.json", "gpurun_out/r5i_bench_driver.json"):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d["config"].get("step_probe"), d["roofline"]["frac"], d["sampling"]["ms_per_step"])
PY
