#!/bin/bash
# round 5 same-box A/B: the tree's library against build/variants/libbhgeo_<name>.so on the headline (full records), the
# direction-only sky frame, the disk frames and the orbit frame
mkdir -p gpurun_out
for w in "--workload frame" "--workload frame --dir-only" "--workload disk" "--workload orbit --steps 60 --warmup 5"; do
  echo "== $w"
  bash scripts/ab.sh "$w" base "$@"
done 2>&1 | tee gpurun_out/r05_ab_$1.log
