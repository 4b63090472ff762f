#!/bin/bash
# round 5, call D: G14 (config 4's DDIM-50 chain at B = 128 vs the reference fixture), G12 with the tightened bars, slab-flush sweep
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_config5_config4_gpu.py tests/test_config2_bench_batch_gpu.py -x -q -m gpu -s -k "ddim50 or train_steps or reference_trainer or G12 or three" 2>&1 | grep -v "amdgpu" | grep "G14\|G12\|passed\|failed\|Error\|assert" | cut -c1-400 | tee gpurun_out/r5d_tests.txt
bash scripts/gpu_wcu.sh DDPM_SLAB_FLUSH_ROWS 2 3 4 6 2>&1 | tee gpurun_out/r5d_sweep_flush.txt
