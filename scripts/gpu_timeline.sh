#!/bin/bash
# kernel-trace of a short bench + the step timeline analysis.  usage: gpu_timeline.sh TAG
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-tl}; mkdir -p $O
export TMPDIR=/tmp DDPM_TORCH_AMD_TRAIN_GRAPH=${GRAPH:-0}
cd /tmp; rm -rf /tmp/kt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/bench.py --steps 12 --warmup 4 --sample-steps 0 --no-cpu-baseline --no-extras > $O/kt.log 2>&1
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
python $R/scripts/step_timeline.py /tmp/kt ${TL_FIRST:-} ${TL_LAST:-} | tee $O/timeline.txt
python $R/scripts/kstats.py /tmp/kt 18 30 | tee $O/kstats.txt
