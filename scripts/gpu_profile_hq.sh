#!/bin/bash
# kernel-trace stats of the CelebA-HQ 256x256 training step at B = 2 (BASELINE config 5, per-GPU work) and of 256x256 sampling at B = 8
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-hq}; mkdir -p $O
export TMPDIR=/tmp DDPM_TORCH_AMD_TRAIN_GRAPH=0
cd /tmp
for mode in train sample; do
  rm -rf /tmp/kt
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/scripts/hq_step.py 12 $mode > $O/kt_$mode.log 2>&1
  cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $O/hq_${mode}_kernel_stats.csv
  tail -1 $O/kt_$mode.log
  python $R/scripts/kstats.py /tmp/kt 16 14
done
