#!/bin/bash
# round 4, first contact of conv3x3_pc_kernel: parity tests of the 3x3 paths, then same-process timings of the shapes with the
# wave-specialised kernel (default), consumers at priority 1, and the round-3 kernel.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-r4a}; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "conv3x3 or conv2d" 2>&1 | tail -15 | tee $O/tests.txt
for cfg in "pc:" "pc_prio:DDPM_C3_PC_FLAGS=1" "old:DDPM_CONV_NO_PC=1"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  echo "== $name" | tee -a $O/shapes.txt
  env $envs timeout 300 python scripts/c3_bench.py 2>&1 | tail -14 | tee -a $O/shapes.txt
done
