#!/bin/bash
# derived utilisation counters for the trace kernel: scripts/gpu_pmc_derived.sh <tag> [bench args]
tag=${1:-dv}; shift
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for ctr in VALUBusy "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" "VALUUtilization" ; do
  name=$(echo $ctr | tr ' ' '_' | cut -c1-24)
  rm -rf /tmp/dv_$tag
  timeout 600 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/dv_$tag -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-seconds 0 "$@" > $R/gpurun_out/dv_$tag.log 2>&1
  for f in $(find /tmp/dv_$tag -name "*counter_collection.csv"); do head -1 $f > $R/gpurun_out/dv_${tag}_$name.csv; grep "trace_" $f >> $R/gpurun_out/dv_${tag}_$name.csv; done
  python3 - <<PY
import csv, collections
d=collections.defaultdict(list); dur=[]
try:
    for r in csv.DictReader(open("$R/gpurun_out/dv_${tag}_$name.csv")):
        d[r["Counter_Name"]].append(float(r["Counter_Value"])); dur.append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
    for k,v in sorted(d.items()): print("$tag", k, "mean", sum(v)/len(v), "kernel_us", sum(dur)/len(dur))
except Exception as e: print("$name failed", e)
PY
done
