#!/bin/bash
# PMC counters of one kernel run: gpu_pmc.sh "<python cmd>" <kernel-substring>
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
CMD=$(echo "$1" | sed "s#scripts/#$GRAFT_REPO_ROOT/scripts/#g"); PAT="$2"
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS" "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM"; do
  rm -rf /tmp/pmc_out
  (cd /tmp && timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_out -- $CMD > /tmp/pmc_log.txt 2>&1)
  python - "$PAT" <<'PY'
import csv, glob, sys, collections
pat = sys.argv[1]
agg = collections.defaultdict(list)
for f in glob.glob("/tmp/pmc_out/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(f"  {k:36s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
done
