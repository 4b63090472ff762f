#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2c; mkdir -p $O
export TMPDIR=/tmp BENCH_NO_SWEEP=1
for mode in eager graph; do
  if [ $mode = graph ]; then export DDPM_TORCH_AMD_TRAIN_GRAPH=1; else export DDPM_TORCH_AMD_TRAIN_GRAPH=0; fi
  rm -rf /tmp/prof_$mode
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$mode -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 4 --sample-steps 0 --no-cpu-baseline > /dev/null 2> $GRAFT_REPO_ROOT/$O/prof_$mode.err)
  python scripts/timeline.py /tmp/prof_$mode > $O/timeline_$mode.txt 2>&1
  head -60 $O/timeline_$mode.txt
done
