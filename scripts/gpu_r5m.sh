#!/bin/bash
# data parallel: the per-step loss reduce off the step's stream — DP tests on one GPU, then the one-rank RCCL bench line three times
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ddp_one_gpu.py tests/test_unet_gpu.py -q -k "two_ranks or rccl or data_parallel" > gpurun_out/r5m_tests.txt 2>&1; grep -n "passed\|failed" gpurun_out/r5m_tests.txt | tail -2
for i in 1 2 3; do
  BENCH_DDP=native timeout 300 python bench.py --steps 20 --warmup 5 --sample-steps 0 --no-cpu-baseline --no-extras > gpurun_out/r5m_ddp1.json 2> gpurun_out/r5m_ddp1.err
  python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5m_ddp1.json").read().strip().splitlines()[0])
print("loss reduce off-chain", d["ms_per_step"], d["config"]["step_probe"], d["config"]["dp"]["exposed_wait_ms"], d["config"]["final_loss"])
PY
done | tee gpurun_out/r5m_ab.txt
timeout 300 python bench.py --steps 20 --warmup 5 --sample-steps 0 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[0]); print('no process group', d['ms_per_step'], d['config']['step_probe'])"
