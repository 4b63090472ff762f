#!/bin/bash
# GroupNorm: second dispatch round started late (DDPM_GN_STAGGER = percent of the estimated load time of a round) — isolated totals, then the step
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-gnst}; mkdir -p $O
for pct in 0 50 100 150; do echo "== DDPM_GN_STAGGER=$pct"; DDPM_GN_STAGGER=$pct GN_ONLY=6 timeout 200 python scripts/gn_bench.py 2>&1 | grep -v amdgpu.ids; done | tee $O/gn.txt
export BENCH_NO_SWEEP=1
for pct in 0 100 0 100; do
  DDPM_GN_STAGGER=$pct timeout 300 python bench.py --steps 40 --warmup 10 --sample-steps 0 --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('stagger=$pct', d['ms_per_step'], 'ms/step', d['value'])"
done | tee $O/ab.txt
