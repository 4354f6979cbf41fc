#!/bin/bash
# HBM traffic per trace launch (two PMC passes) + time for a library variant: scripts/gpu_traffic.sh <variant|base> [bench args]
v=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
if [ "$v" != "base" ]; then export BHGEO_LIB=$R/build/variants/libbhgeo_$v.so; fi
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/tr_$c
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/tr_$c -- python3 $R/bench.py --steps 3 --warmup 1 --ramp-seconds 0 --cpu-seconds 0 "$@" > /dev/null 2>&1
  for f in $(find /tmp/tr_$c -name "*counter_collection.csv"); do head -1 $f > /tmp/tr_$c.csv; grep "trace_" $f >> /tmp/tr_$c.csv; done
done
python3 $R/scripts/summarize_pmc.py /tmp/tr_FETCH_SIZE.csv /tmp/tr_WRITE_SIZE.csv /tmp/tr_sum.json trace_ | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$v traffic MB per launch: %.1f (fetch x2 %.1f, write %.1f)' % (d['hbm_bytes_per_launch'] / 1e6, d['fetch_bytes_corrected'] / 1e6, d['write_bytes'] / 1e6))"
