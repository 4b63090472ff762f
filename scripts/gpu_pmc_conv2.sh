#!/bin/bash
# 3x3 forward conv, isolated launches, three shapes: MFMA-busy counters (pass a) and kernel-trace durations without counters (pass b)
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-pmc2}; mkdir -p $O
export TMPDIR=/tmp
bash scripts/gpu_pmc_conv.sh ${1:-pmc2} > /dev/null 2>&1
echo "shape,kernel_trace_avg_us" > $O/conv3x3_durations.csv
for shape in "128 16 512 256" "128 32 128 128" "128 16 256 256"; do
  rm -rf /tmp/kt1
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt1 -- python $R/scripts/one_kernel.py fwd $shape 3 bf16 20 > /tmp/kt1.log 2>&1)
  f=$(find /tmp/kt1 -name "*kernel_stats.csv" | head -1)
  python - "$shape" "$f" >> $O/conv3x3_durations.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[2])):
    if "conv3x3" in r["Name"]:
        print(f'"{sys.argv[1]}",{float(r["AverageNs"]) / 1e3:.2f}')
PY
done
cat $O/pmc_mfma_conv3x3.csv $O/conv3x3_durations.csv
