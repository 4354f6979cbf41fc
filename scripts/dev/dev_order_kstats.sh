#!/bin/bash
# per-kernel durations of the Kerr + disk call under two tile orders (rocprofv3 kernel stats)
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for o in none measured; do
  rm -rf /tmp/ko_$o
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ko_$o -- python3 $R/bench.py --steps 40 --warmup 5 --lean --workload disk --rhs kerr --order $o > /tmp/ko_$o.log 2>&1
  tail -1 /tmp/ko_$o.log | cut -c1-300
  echo "== $o"
  for f in $(find /tmp/ko_$o -name "*kernel_stats.csv"); do python3 - <<PY
import csv
for r in csv.DictReader(open("$f")):
    if float(r["Percentage"]) > 0.5:
        print("   %-70s calls %5s avg %10.1f us  %5.1f %%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
  done
done
