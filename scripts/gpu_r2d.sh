#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2d; mkdir -p $O
export TMPDIR=/tmp BENCH_NO_SWEEP=1 DDPM_TORCH_AMD_TRAIN_GRAPH=0
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -p no:cacheprovider -k "wgrad" 2>&1 | tail -25 > $O/tests.log
tail -6 $O/tests.log
run() { name=$1; shift; env "$@" timeout 300 python bench.py --steps 40 --warmup 6 --sample-steps 0 --no-cpu-baseline > /dev/null 2> $O/$name.err; grep -o '"train_only_ms_per_step": [0-9.]*' $O/$name.err | sed "s/^/$name /"; python - <<PY
import json,re
s=open("$O/$name.err").read(); i=s.find('{"train_only'); d=json.loads(s[i:s.find('\n',i)] if '\n' in s[i:] else s[i:])
for k,v in d["roofline"]["isolated"]["per_kernel"].items(): print("   iso",k,v["launches"],v["ms"],v["tflops"])
for k,v in d["roofline"]["per_kernel"].items(): print("   prod",k,v["launches"],v["ms"],v["tflops"])
PY
}
run patch_slab DDPM_WGRAD3=1
run patch_atomic DDPM_WGRAD3=1 DDPM_WGRAD3_ATOMIC=1
run generic DDPM_WGRAD3=0
