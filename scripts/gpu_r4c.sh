#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-r4c}; mkdir -p $O
timeout 1200 python -m pytest tests/test_training_convergence_gpu.py -x -q -s 2>&1 | grep -v amdgpu.ids | tail -40 | tee $O/tests.txt
export BENCH_NO_SWEEP=1
BENCH_DDP=native timeout 600 python bench.py --steps 40 --warmup 10 --sample-steps 0 --no-cpu-baseline --no-extras 2>$O/bench.err | grep '^{' | tail -1 > $O/bench_ddp1.json; tail -3 $O/bench.err
python - $O/bench_ddp1.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read()); r = d["roofline"]
print(d["ms_per_step"], "ms/step; dp:", json.dumps(d["config"]["dp"])[:900])
print("practical:", r["practical_peak"]["tflops"], "frac_of_practical", r["frac_of_practical_peak"], "frac", r["frac"])
print("by_duration:", r["dominant_by_duration"], "| by_cu:", r["dominant_by_cu_time"])
for k, v in list(r["hbm_kernels"]["in_step"].items())[:6]: print("  hbm in-step", k[:70], v)
for k, v in list(r["hbm_kernels"]["isolated"].items())[:8]: print("  hbm isolated", k[:70], v)
PY
tail -5 $O/bench.err
