#!/bin/bash
# round 4: LDS / wait counters of the 3x3 kernels (isolated launches), for the wave-specialised kernel, the round-3 kernel and the 3x3 weight gradient
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r4cnt}; mkdir -p $O
export TMPDIR=/tmp
run() {   # name, env, kind, shape, kernel substring
  echo "== $1: one_kernel.py $3 $4 3 bf16 6  [$2]  kernel ~ $5" | tee -a $O/counters.txt
  for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU" "SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM"; do
    rm -rf /tmp/pm1
    (cd /tmp && env $2 timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pm1 -- python $R/scripts/one_kernel.py $3 $4 3 bf16 6 > /tmp/pm1.log 2>&1)
    python - "$5" <<'PY' | tee -a $O/counters.txt
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
run "conv3x3_pc_kernel 128->128 @ 32x32 B=128" "X=1" fwd "128 32 128 128" conv3x3_pc
run "conv3x3_stream_kernel<16> (round 3) 128->128 @ 32x32 B=128" "DDPM_CONV_NO_PC=1" fwd "128 32 128 128" conv3x3_stream
run "conv3x3_pc_kernel 512->256 @ 16x16 B=128" "X=1" fwd "128 16 512 256" conv3x3_pc
run "wgrad3x3_kernel 128x128 @ 32x32 B=128" "X=1" wgrad "128 32 128 128" wgrad3x3
run "wgrad3x3_kernel 256x256 @ 16x16 B=128" "X=1" wgrad "128 16 256 256" wgrad3x3
