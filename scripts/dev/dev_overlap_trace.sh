#!/bin/bash
# do consecutive frames on two streams overlap?  kernel start/end timestamps of a two-in-flight run of the 1/8 shard
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ov
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/ov -- python3 $R/scripts/dev/dev_shard_run.py ${1:-8} 40 two > /tmp/ov.log 2>&1
f=$(find /tmp/ov -name "*kernel_trace.csv" | head -1)
python3 - <<PY
import csv
rows=[r for r in csv.DictReader(open("$f"))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
rows=rows[-60:]
t0=int(rows[0]["Start_Timestamp"])
for r in rows[:36]:
    print("%-28s q%-3s start %9.1f end %9.1f dur %7.1f us" % (r["Kernel_Name"][:28], r.get("Queue_Id","?"), (int(r["Start_Timestamp"])-t0)/1e3, (int(r["End_Timestamp"])-t0)/1e3, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3))
PY
