#!/bin/bash
# round 6 lease i: the narrow-stores upper bound with a harness that does not read results
mkdir -p gpurun_out
for i in 1 2 3; do for v in base nonarrow; do
  BHGEO_LIB=$PWD/build/variants/libbhgeo_$v.so timeout 200 python scripts/dev/dev_r06_trace_time.py 5 200 2>/dev/null | tail -1
done; done | tee gpurun_out/r06_narrow_stores_ab.log
