#!/bin/bash
# A/B of env settings within ONE box: usage: gpu_env_ab.sh TAG "ENV1=a ENV2=b" "ENV1=c" ...   (each argument = one configuration; "" = defaults)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-envab}; shift; mkdir -p $O
export TMPDIR=/tmp BENCH_NO_SWEEP=1
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else 'n/a')" 2>/dev/null
for rep in 1 2; do
  for cfg in "$@"; do
    env $cfg DDPM_TORCH_AMD_TRAIN_GRAPH=${GRAPH:-0} timeout 600 python bench.py --steps ${STEPS:-60} --warmup 12 --sample-steps ${SAMPLE:-0} --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']
print('[$cfg]', d['ms_per_step'], 'ms/step;', r['kernel'][:28], r['achieved'], 'TF in-step,', r['isolated']['per_kernel'][r['kernel']]['tflops'], 'isolated; all-mfma in-step', r['all_mfma_kernels']['ms'], 'ms;', ('sampling %.3f ms/step' % d['sampling']['ms_per_step']) if 'sampling' in d else '')" | tee -a $O/ab.txt
  done
done
