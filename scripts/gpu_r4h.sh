#!/bin/bash
# round 4, call H: sampling-chain kernel trace; side stream on / off re-check; a second default-size step for box variance
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-r4h}; mkdir -p $O
bash scripts/gpu_sample_profile.sh ${1:-r4h} | tail -30
cd "$GRAFT_REPO_ROOT"
export BENCH_NO_SWEEP=1
for mode in "A=1" "DDPM_SIDE_STREAM=0" "A=1" "DDPM_SIDE_STREAM=0"; do
  env $mode timeout 300 python bench.py --steps 40 --warmup 10 --sample-steps 0 --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$mode', d['ms_per_step'], 'ms/step', d['value'])"
done | tee $O/ab.txt
