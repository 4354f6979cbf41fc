#!/bin/bash
# A/B of the working tree's library against build/variants/libbhgeo_head.so (the last commit) on the main workloads,
# then the full GPU suite on the working tree
mkdir -p gpurun_out
for w in "--workload frame" "--workload frame --full-records" "--workload disk" "--workload orbit --steps 40 --warmup 5" "--workload frame --rhs kerr --steps 60 --warmup 5"; do
  echo "== $w"
  bash scripts/ab.sh "$w" base head base head
done 2>&1 | tee gpurun_out/r04_ab.log
timeout 2400 python -m pytest tests -q -m gpu --timeout 900 -x > gpurun_out/r04_pytest_gpu.log 2>&1
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/r04_pytest_gpu.log | tail -10
