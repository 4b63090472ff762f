#!/bin/bash
# Same-box A/B of a library change: previous library (objects built from HEAD's sources, linked as libddpm_hip_prev.so) vs the working tree's.  Runs on a scratch copy: the product library is swapped there only.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-ab3}; mkdir -p $O
L=ddpm-torch_amd/csrc
cp $L/libddpm_hip.so /tmp/new.so
CMD="python bench.py --steps 80 --warmup 20 --sample-steps 300 --no-cpu-baseline --no-extras"
run() { $CMD 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$1', d['ms_per_step'], d['sampling']['ms_per_step'], d['config']['step_execution'])"; }
for rep in 1 2 3; do
  cp $L/libddpm_hip_prev.so $L/libddpm_hip.so; run "prev        " | tee -a $O/ab.txt
  cp /tmp/new.so $L/libddpm_hip.so; run "new         " | tee -a $O/ab.txt
done
