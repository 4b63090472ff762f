"""Per-step kernel totals from a rocprofv3 --kernel-trace --stats directory:  python scripts/kstats.py DIR STEPS [TOP]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
steps, top = float(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 24
rows = list(csv.DictReader(open(f)))
print("sum of kernel durations per step: %.3f ms" % (sum(float(r["TotalDurationNs"]) for r in rows) / steps / 1e6))
for r in rows[:top]:
    print("%7.3f ms  n/step=%6.1f avg=%7.1f us  %s" % (float(r["TotalDurationNs"]) / steps / 1e6, int(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3, r["Name"][:100]))
