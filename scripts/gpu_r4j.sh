#!/bin/bash
# round 4, call J: does the driver's short run (--steps 20 --warmup 5) time the same step as the default (--steps 100 --warmup 20)?
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-r4j}; mkdir -p $O
export BENCH_NO_SWEEP=1
for a in "20 5" "100 20" "20 5" "100 20" "20 0"; do
  set -- $a
  timeout 300 python bench.py --gpus 1 --steps $1 --warmup $2 --sample-steps 0 --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('steps $1 warmup $2:', d['ms_per_step'], 'ms/step', d['value'], d['config'].get('step_execution'))"
done | tee $O/ab.txt
