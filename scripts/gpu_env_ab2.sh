#!/bin/bash
# Same-box A/B of environment switches on the training step: gpu_env_ab2.sh TAG "ENV_A=.. ENV_B=.." [label]; alternates baseline env / default 3 times
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-envab}; mkdir -p $O
CMD="python bench.py --steps 60 --warmup 15 --sample-steps 0 --no-cpu-baseline --no-extras"
run() { $CMD 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$1', d['ms_per_step'], d['ms_per_step_blocks'])"; }
for rep in 1 2 3; do
  env $2 bash -c "$(declare -f run); CMD='$CMD'; run 'with [$2] '" | tee -a $O/ab.txt
  run "default      " | tee -a $O/ab.txt
done
