#!/bin/bash
# per-kernel durations of the bench command: scripts/gpu_ktrace.sh <tag> [bench args]
tag=${1:-kt}; shift
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_$tag
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$tag -- python3 $R/bench.py --steps 10 --warmup 2 --cpu-seconds 0 "$@" > $R/gpurun_out/kt_$tag.log 2>&1
for f in $(find /tmp/kt_$tag -name "*kernel_stats.csv"); do cp $f $R/gpurun_out/kernel_stats_$tag.csv; done
python3 - <<PY
import csv
for r in csv.DictReader(open("$R/gpurun_out/kernel_stats_$tag.csv")):
    if float(r["Percentage"]) > 0.3: print(r["Name"][:70].ljust(70), r["Calls"], "avg_us %.1f" % (float(r["AverageNs"])/1e3), r["Percentage"]+"%")
PY
