#!/bin/bash
# rocprofv3 kernel stats of the emulated strong-scaling shards (1/1, 1/2, 1/4, 1/8 of the fixed 1024x1024 x5 frame on ONE GPU)
# -> gpurun_out/prof_shard<N>_kernel_stats.csv (+ the timing line); bash scripts/gpu_profile_shards.sh
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for N in 1 2 4 8; do
  rm -rf /tmp/sh_$N
  (cd $R && timeout 300 python3 scripts/dev/dev_shard_run.py $N 200 | tail -1) > $O/prof_shard${N}_timing.txt
  (cd $R && timeout 300 python3 scripts/dev/dev_shard_run.py $N 200 twoprio | tail -1) >> $O/prof_shard${N}_timing.txt
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sh_$N -- python3 $R/scripts/dev/dev_shard_run.py $N 200 > $O/prof_shard${N}_rocprof.log 2>&1
  for f in $(find /tmp/sh_$N -name "*kernel_stats.csv"); do cp $f $O/prof_shard${N}_kernel_stats.csv; done
  cat $O/prof_shard${N}_timing.txt
  python3 - <<PY
import csv
for r in csv.DictReader(open("$O/prof_shard${N}_kernel_stats.csv")):
    if float(r["Percentage"]) > 1.0:
        print("   %-60s calls %5s avg %10.1f us  %5.1f %%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
done
