#!/bin/bash
# round 5 experiment: work handed out in units of 64 / BHG_UNITS_PER_BATCH rays near the end of a launch (same box):
# parity tests against each variant, then the shards and the workloads, base against build/variants/libbhgeo_<name>.so
mkdir -p gpurun_out
{
for v in "$@"; do
  echo "== parity with $v"
  BHGEO_LIB=$PWD/build/variants/libbhgeo_$v.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_frame.py -q -m gpu -x --timeout 600 -p no:cacheprovider 2>&1 | tail -3 | cut -c1-300
done
for rep in 1 2; do
for v in base "$@"; do
  if [ "$v" = "base" ]; then lib=""; else lib="$PWD/build/variants/libbhgeo_$v.so"; fi
  for N in 8 4 2 1; do
    echo -n "$v  "; BHGEO_LIB=$lib timeout 300 python3 scripts/dev/dev_shard_run.py $N 400 bench 2>&1 | tail -1
  done
done; done
for w in "--workload frame" "--workload disk" "--workload orbit --steps 60 --warmup 5" "--rhs kerr --steps 40 --warmup 5"; do
  echo "== $w"
  bash scripts/ab.sh "$w" base "$@"
done
} 2>&1 | tee gpurun_out/r05_units_ab.log
