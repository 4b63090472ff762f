#!/bin/bash
# Stride-2 data gradient in parity-phase form: parity tests, isolated timing and a same-box step A/B (DDPM_NO_PHASE_DGRAD=1 = the zero-dilated form).
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-phase}; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "conv_dgrad_stride2 or conv_fwd or small_grid" 2>&1 | tail -5 | tee $O/tests.txt
echo "--- zero-dilated" | tee $O/s2.txt; DDPM_NO_PHASE_DGRAD=1 timeout 300 python scripts/s2_bench.py 2>&1 | grep down | tee -a $O/s2.txt
echo "--- parity-phase" | tee -a $O/s2.txt; timeout 300 python scripts/s2_bench.py 2>&1 | grep down | tee -a $O/s2.txt
CMD="python bench.py --steps 60 --warmup 15 --sample-steps 0 --no-cpu-baseline --no-extras"
run() { $CMD 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$1', d['ms_per_step'], d['ms_per_step_blocks'])"; }
for rep in 1 2 3; do
  DDPM_NO_PHASE_DGRAD=1 run "dilated " | tee -a $O/ab.txt
  run "phase   " | tee -a $O/ab.txt
done
