"""Register / LDS / occupancy table of every kernel in csrc/ as hipcc reports it (-Rpass-analysis=kernel-resource-usage; no GPU needed).
    python scripts/kernel_resources.py > profiles/rNN_kernel_resources.csv"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ddpm-torch_amd", "csrc")
sys.path.insert(0, CSRC)
import build  # noqa: E402
print("source,kernel,vgprs,agprs,sgprs,vgpr_spills,sgpr_spills,scratch_bytes,static_lds_bytes,occupancy_waves_per_simd")
for src in build.SOURCES:
    path = os.path.join(CSRC, src if src.endswith(".hip") else src + ".hip")
    out = subprocess.run(["/opt/rocm/bin/hipcc", *build.FLAGS, "-Rpass-analysis=kernel-resource-usage", "-c", path, "-o", "/dev/null"],
                         capture_output=True, text=True).stderr
    cur = {}
    for line in out.splitlines():
        m = re.search(r"remark: \s*(Function Name|VGPRs|AGPRs|TotalSGPRs|VGPRs Spill|SGPRs Spill|ScratchSize \[bytes/lane\]|LDS Size \[bytes/block\]|Occupancy \[waves/SIMD\]): (\S+)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == "Function Name":
            cur = {"name": v}
        cur[k] = v
        if k.startswith("LDS Size"):
            name = subprocess.run(["c++filt", cur["name"]], capture_output=True, text=True).stdout.strip()
            name = re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0]
            print(",".join([os.path.basename(path), '"' + name + '"', cur.get("VGPRs", ""), cur.get("AGPRs", ""), cur.get("TotalSGPRs", ""), cur.get("VGPRs Spill", ""),
                            cur.get("SGPRs Spill", ""), cur.get("ScratchSize [bytes/lane]", ""), cur.get("LDS Size [bytes/block]", ""), cur.get("Occupancy [waves/SIMD]", "")]))
