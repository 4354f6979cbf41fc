#!/bin/bash
# SQ counter pass for the trace kernel: where do wave cycles go?  usage: gpu_pmc_sq.sh <tag> [bench args]
tag=${1:-sq}; shift
BARGS="$@"
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sq_$tag
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d /tmp/sq_$tag -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-seconds 0 $BARGS > $R/gpurun_out/sq_$tag.log 2>&1
for f in $(find /tmp/sq_$tag -name "*counter_collection.csv"); do head -1 $f > $R/gpurun_out/sq_$tag.csv; grep trace_ $f >> $R/gpurun_out/sq_$tag.csv; done
python3 - <<PY
import csv, collections
d=collections.defaultdict(list)
for r in csv.DictReader(open("$R/gpurun_out/sq_$tag.csv")):
    if "trace_" in r["Kernel_Name"]: d[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(d.items()): print(k, sum(v)/len(v))
PY
