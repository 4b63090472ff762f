#!/bin/bash
# Kernel-trace statistics of the sampling chain (graph replay), B=128:  gpu_sample_profile.sh TAG -> gpurun_out/TAG/sampling_kernel_stats.csv
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-sprof}; mkdir -p $O
export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/ks
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $R/bench.py --steps 2 --warmup 1 --sample-steps 300 --no-cpu-baseline --no-extras > $O/ks.log 2>&1
cp $(find /tmp/ks -name "*kernel_stats.csv" | head -1) $O/sampling_kernel_stats.csv
head -40 $O/sampling_kernel_stats.csv | cut -c1-150
