#!/bin/bash
# Round profile set: kernel-trace stats of the bench command, then HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes), then MFMA-busy.
#   usage: gpu_profile.sh TAG      -> gpurun_out/TAG/{kernel_stats.csv, hbm_traffic.json, mfma_busy.csv, bench_default.json}
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-prof}; mkdir -p $O
export TMPDIR=/tmp DDPM_TORCH_AMD_TRAIN_GRAPH=${PROFILE_STEP_FORM:-plan}
CMD="python $R/bench.py --steps 10 --warmup 3 --sample-steps 0 --no-cpu-baseline --no-extras"
cd /tmp
rm -rf /tmp/kt; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- $CMD > $O/kt.log 2>&1
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
rm -rf /tmp/pf /tmp/pw /tmp/pm
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf -- $CMD > $O/pf.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pw -- $CMD > $O/pw.log 2>&1
python $R/scripts/pmc_traffic.py /tmp/pf /tmp/pw $O/hbm_traffic.json "bench.py --steps 10 --warmup 3 --sample-steps 0 --no-cpu-baseline --no-extras (DDPM_TORCH_AMD_TRAIN_GRAPH=plan: the launch-plan form of the step)"
timeout 600 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --kernel-trace --output-format csv -d /tmp/pm -- $CMD > $O/pm.log 2>&1
python - $O/mfma_busy.csv <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob("/tmp/pm/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        a = agg[r["Kernel_Name"]][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
with open(sys.argv[1], "w") as o:
    o.write("# rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --kernel-trace; per-dispatch averages\n")
    o.write("# mfma_utilisation = SQ_VALU_MFMA_BUSY_CYCLES (summed over 1024 SIMDs) / (1024 x GRBM_GUI_ACTIVE / 8)\n")
    o.write("kernel,dispatches,SQ_BUSY_CYCLES,SQ_VALU_MFMA_BUSY_CYCLES,GRBM_GUI_ACTIVE,SQ_WAVE_CYCLES,mfma_utilisation\n")
    rows = []
    for k, c in agg.items():
        n = c["SQ_BUSY_CYCLES"][0] or 1
        g = lambda x: c[x][1] / max(c[x][0], 1)
        rows.append((g("GRBM_GUI_ACTIVE") * n, k, n, g("SQ_BUSY_CYCLES"), g("SQ_VALU_MFMA_BUSY_CYCLES"), g("GRBM_GUI_ACTIVE"), g("SQ_WAVE_CYCLES")))
    for _, k, n, b, m, gui, w in sorted(rows, reverse=True)[:40]:
        o.write(f"\"{k[:120]}\",{n},{b:.0f},{m:.0f},{gui:.0f},{w:.0f},{(m / (128 * gui) if gui else 0):.4f}\n")
PY
cd $R
unset DDPM_TORCH_AMD_TRAIN_GRAPH
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
head -25 $O/kernel_stats.csv | cut -c1-200; head -12 $O/mfma_busy.csv | cut -c1-220; tail -c 600 $O/bench_default.json
