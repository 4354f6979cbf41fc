#!/bin/bash
# round 5 experiment: issue priority for waves that carry long rays (-DBHG_LONG_RAY_PRIO=<steps>), same box:
# the 1/8, 1/4 shards and the whole frame, base against build/variants/libbhgeo_<name>.so
mkdir -p gpurun_out
{
python3 scripts/dev/dev_steps_hist.py frame
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
} 2>&1 | tee gpurun_out/r05_prio_ab.log
