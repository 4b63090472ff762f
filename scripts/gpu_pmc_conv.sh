#!/bin/bash
# MFMA-busy counters of the 3x3 forward conv on the two shapes profiles/r01n used (isolated launches): -> gpurun_out/$1/pmc_mfma_conv3x3.csv
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-pmc}; mkdir -p $O
export TMPDIR=/tmp
echo "# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace -- python scripts/one_kernel.py fwd B H C N 3 bf16 6" > $O/pmc_mfma_conv3x3.csv
echo "# per-dispatch averages of the conv kernel; SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs: MFMA utilisation = MFMA_BUSY / (1024 x GRBM_GUI_ACTIVE / 8)" >> $O/pmc_mfma_conv3x3.csv
echo "shape,kernel,GRBM_GUI_ACTIVE,SQ_BUSY_CYCLES,SQ_VALU_MFMA_BUSY_CYCLES,SQ_WAVES,mfma_utilisation" >> $O/pmc_mfma_conv3x3.csv
for shape in "128 16 512 256" "128 32 128 128" "128 16 256 256"; do
  rm -rf /tmp/pm1
  (cd /tmp && timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pm1 -- python $R/scripts/one_kernel.py fwd $shape 3 bf16 6 > /tmp/pm1.log 2>&1)
  python - "$shape" >> $O/pmc_mfma_conv3x3.csv <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/pm1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv3x3" in r["Kernel_Name"]:
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
B, H, C, N = sys.argv[1].split()
for k, c in agg.items():
    g = lambda x: sum(c[x]) / max(len(c[x]), 1)
    name = k.split("(")[0].replace("void ", "").replace("(anonymous namespace)::", "")
    print(f'"3x3 conv B={B} {H}x{H} {C}->{N} (bf16)",{name},{g("GRBM_GUI_ACTIVE"):.1f},{g("SQ_BUSY_CYCLES"):.1f},{g("SQ_VALU_MFMA_BUSY_CYCLES"):.0f},{g("SQ_WAVES"):.0f},{g("SQ_VALU_MFMA_BUSY_CYCLES") / (128 * g("GRBM_GUI_ACTIVE")):.3f}')
PY
done
cat $O/pmc_mfma_conv3x3.csv
