#!/bin/bash
# true kernel durations (rocprofv3 kernel-trace stats) of a python micro-benchmark.  usage: gpu_kstat.sh TAG "ENV=.." script.py [grep pattern]
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-ks}; mkdir -p $O
export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/ks1
env $2 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks1 -- python $R/$3 > $O/run.log 2>&1
f=$(find /tmp/ks1 -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats.csv
python - "$f" "${4:-.}" <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], r["Name"]):
        print("%8.1f us avg  n=%5s  min %7.1f  max %7.1f  %s" % (float(r["AverageNs"]) / 1e3, r["Calls"], float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Name"][:90]))
PY
