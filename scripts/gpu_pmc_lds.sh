#!/bin/bash
# LDS / issue-stall counters of ONE conv shape (isolated launches of scripts/one_kernel.py): bank conflicts, LDS busy, wait buckets.
#   usage: gpu_pmc_lds.sh TAG "B H C N" [kernel-name substring]
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-pmclds}; mkdir -p $O
export TMPDIR=/tmp
shape=${2:-"128 32 128 128"}; kn=${3:-conv3x3}
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM"; do
  rm -rf /tmp/pm1
  (cd /tmp && timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pm1 -- python $R/scripts/one_kernel.py fwd $shape 3 bf16 6 > /tmp/pm1.log 2>&1)
  python - "$kn" <<'PY' | tee -a $O/pmc_lds.txt
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob("/tmp/pm1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[1] in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(f"{k:28s} {sum(v) / len(v):16.0f}   (n={len(v)})")
PY
done
