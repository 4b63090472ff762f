#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2b; mkdir -p $O
export TMPDIR=/tmp BENCH_NO_SWEEP=1
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_known_answers_gpu.py tests/test_unet_gpu.py tests/test_multi_gpu.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -30 > $O/tests.log
tail -3 $O/tests.log
run() { name=$1; shift; env "$@" timeout 300 python bench.py --steps 40 --warmup 6 --sample-steps 0 --no-cpu-baseline > /dev/null 2> $O/$name.err; grep -o '"train_only_ms_per_step": [0-9.]*' $O/$name.err | sed "s/^/$name /"; }
run graph_side DDPM_TORCH_AMD_TRAIN_GRAPH=1
run graph_noside DDPM_TORCH_AMD_TRAIN_GRAPH=1 DDPM_SIDE_STREAM=0
run eager_side DDPM_TORCH_AMD_TRAIN_GRAPH=0
run eager_noside DDPM_TORCH_AMD_TRAIN_GRAPH=0 DDPM_SIDE_STREAM=0
run autograd_side DDPM_TORCH_AMD_DIRECT_STEP=0
