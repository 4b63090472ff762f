#!/bin/bash
# per-shape MFMA timings of the training step (bench.py's profiled eager step, gated so that the launches run back to back): in-step and isolated
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-shapes}; mkdir -p $O
export BENCH_NO_SWEEP=1
for cfg in "pc:" "old:DDPM_CONV_NO_PC=1" "pc_nogate:BENCH_NO_GATE=1"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  env $envs BENCH_SHAPES=$O/shapes_$name.txt timeout 600 python bench.py --steps 60 --warmup 12 --sample-steps 0 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $O/bench_$name.json
  python - $O/bench_$name.json $name <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read()); r = d["roofline"]
print(sys.argv[2], d["ms_per_step"], "ms/step |", r["kernel"][:40], "in-step", r["achieved"], "TF", r["avg_launch_us"], "us | isolated", r["isolated"]["per_kernel"][r["kernel"]]["tflops"], "TF | all-mfma in-step", r["all_mfma_kernels"]["ms"], "ms, isolated", r["isolated"]["all_mfma_kernels"]["ms"], "ms")
PY
done
grep "conv 3x3\|conv3x3" $O/shapes_pc.txt | head -50
