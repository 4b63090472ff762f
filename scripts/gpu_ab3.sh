#!/bin/bash
# Same-box A/B of a library change: previous library (objects built from HEAD's sources, linked as libddpm_hip_prev.so) vs the working tree's.
# The variant is selected with DDPM_HIP_LIB (ddpm_torch/_hip.py): the product library is never overwritten.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-ab3}; mkdir -p $O
L=$PWD/ddpm-torch_amd/csrc
CMD="python bench.py --steps 80 --warmup 20 --sample-steps 300 --no-cpu-baseline --no-extras"
run() { $CMD 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$1', d['ms_per_step'], d['sampling']['ms_per_step'], d['config']['step_execution'])"; }
for rep in 1 2 3; do
  DDPM_HIP_LIB=${PREV_LIB:-$L/libddpm_hip_prev.so} run "prev        " | tee -a $O/ab.txt
  run "new         " | tee -a $O/ab.txt
done
