#!/bin/bash
# round 2, GPU call A: full GPU test suite + bench with the captured training step on / off
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2a; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -60 > $O/tests.log
echo "tests rc=$?" >> $O/tests.log
tail -5 $O/tests.log
BENCH_NO_SWEEP=1 timeout 600 python bench.py --steps 40 --warmup 6 --sample-steps 200 --no-cpu-baseline > $O/bench_graph.json 2> $O/bench_graph.err
echo "bench graph rc=$?"; head -c 600 $O/bench_graph.json; tail -3 $O/bench_graph.err
DDPM_TORCH_AMD_TRAIN_GRAPH=0 BENCH_NO_SWEEP=1 timeout 600 python bench.py --steps 40 --warmup 6 --sample-steps 0 --no-cpu-baseline > $O/bench_eager.json 2> $O/bench_eager.err
echo "bench eager rc=$?"; head -c 300 $O/bench_eager.json; tail -3 $O/bench_eager.err
DDPM_TORCH_AMD_DIRECT_STEP=0 BENCH_NO_SWEEP=1 timeout 600 python bench.py --steps 40 --warmup 6 --sample-steps 0 --no-cpu-baseline > $O/bench_autograd.json 2> $O/bench_autograd.err
echo "bench autograd rc=$?"; head -c 300 $O/bench_autograd.json; tail -3 $O/bench_autograd.err
