#!/bin/bash
# round 5, first GPU call: the launch-plan form of the step — equality tests, host time per form, the driver's bench command
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_configs_gpu.py tests/test_abi_surface.py -x -q -m gpu -k "captured or auto_mode or recaptured or survives or plan" > gpurun_out/r5a_tests.txt 2>&1
tail -5 gpurun_out/r5a_tests.txt
timeout 300 python scripts/host_overhead.py > gpurun_out/r5a_host.txt 2>&1
cat gpurun_out/r5a_host.txt | tail -8
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r5a_bench.json 2> gpurun_out/r5a_bench.err
tail -3 gpurun_out/r5a_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5a_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["config"].get("step_execution"), d["config"].get("step_probe"))
PY
