#!/bin/bash
# round 4, call K: LDS / wait / MFMA counters of the 3x3 weight-gradient kernel (isolated launches, scripts/wgrad_one.py) and of the 1x1 kernel's new tile
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r4k}; mkdir -p $O
export TMPDIR=/tmp
run() {   # name, command, kernel substring
  echo "== $1: $2   kernel ~ $3" | tee -a $O/counters.txt
  for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU" "SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM"; do
    rm -rf /tmp/pm1
    (cd /tmp && timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pm1 -- python $R/scripts/$2 > /tmp/pm1.log 2>&1)
    python - "$3" <<'PY' | tee -a $O/counters.txt
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob("/tmp/pm1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[1] in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(f"   {k:28s} {sum(v) / len(v):16.0f}   (n={len(v)})")
PY
  done
}
run "wgrad3x3_kernel 128x128 @ 32x32 B=128" "wgrad_one.py 32 128 128" wgrad3x3
run "wgrad3x3_kernel 256x256 @ 16x16 B=128" "wgrad_one.py 16 256 256" wgrad3x3
run "pw_conv_kernel<128,256> 256->256 @ 16x16 B=128" "one_kernel.py fwd 128 16 256 256 1 bf16 6" pw_conv
