#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-tl}; mkdir -p $O
export TMPDIR=/tmp BENCH_NO_SWEEP=1 DDPM_TORCH_AMD_TRAIN_GRAPH=0
rm -rf /tmp/prof_tl
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tl -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 4 --sample-steps 0 --no-cpu-baseline > /dev/null 2> $GRAFT_REPO_ROOT/$O/prof.err)
python scripts/timeline.py /tmp/prof_tl > $O/timeline.txt 2>&1
head -75 $O/timeline.txt
