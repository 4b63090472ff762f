#!/bin/bash
# power / clock of the chip while the training step runs: is the whole step at the power cap, or only its MFMA kernels?
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-pwr}; mkdir -p $O
export BENCH_NO_SWEEP=1
(timeout 300 python bench.py --steps 1500 --warmup 20 --sample-steps 0 --no-cpu-baseline --no-extras > $O/bench.json 2>/dev/null) &
for i in $(seq 1 45); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | sed 's/.*: //' | tr '\n' ' '; echo; sleep 0.7; done | grep -v "^$" > $O/power_train.txt
wait
sort -t' ' -k3 -n $O/power_train.txt | awk '{print $1, $NF}' | sort | uniq -c | sort -k3 -n | tail -12
python -c "import json; d=json.loads(open('$O/bench.json').readlines()[-1]); print('ms/step', d['ms_per_step'])"
