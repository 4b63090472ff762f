#!/bin/bash
# kernel-trace stats of the bench command -> per-step kernel totals
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-kt}; mkdir -p $O
export TMPDIR=/tmp DDPM_TORCH_AMD_TRAIN_GRAPH=0 BENCH_NO_SWEEP=1
CMD="python $R/bench.py --steps 10 --warmup 3 --sample-steps 0 --no-cpu-baseline --no-extras"
cd /tmp
rm -rf /tmp/kt; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- $CMD > $O/kt.log 2>&1
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
python $R/scripts/kstats.py /tmp/kt 18 45 | tee $O/kstats.txt
