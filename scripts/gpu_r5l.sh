#!/bin/bash
# data parallel: chunk exchanges issued behind the SIDE stream (no main-stream joins) — the DP tests on one GPU, then the one-rank RCCL bench
# line with the round-4 form (DDPM_DP_ISSUE_ON_SIDE=0) and the new one, three pairs
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 200 python scripts/dp_capture_debug.py 2>&1 | grep "steps done"
timeout 900 python -m pytest tests/test_ddp_one_gpu.py tests/test_unet_gpu.py -q -k "two_ranks or rccl or data_parallel" > gpurun_out/r5l_tests.txt 2>&1; grep -n "passed\|failed" gpurun_out/r5l_tests.txt | tail -2
for i in 1 2 3; do for v in 0 1; do FIRST_FORM=x
  BENCH_DDP=native DDPM_DP_ISSUE_ON_SIDE=$v timeout 300 python bench.py --steps 20 --warmup 5 --sample-steps 0 --no-cpu-baseline --no-extras > gpurun_out/r5l_ddp1_$v.json 2> gpurun_out/r5l_ddp1_$v.err
  python - $v <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/r5l_ddp1_{sys.argv[1]}.json").read().strip().splitlines()[0])
print("DDPM_DP_ISSUE_ON_SIDE=" + sys.argv[1], d["ms_per_step"], d["config"]["step_probe"], d["config"]["dp"]["exposed_wait_ms"])
PY
done; done | tee gpurun_out/r5l_ab.txt
