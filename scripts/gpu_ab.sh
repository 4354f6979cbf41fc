#!/bin/bash
# parity tests of the working tree's library, then an A/B of library variants over the main bench workloads
# usage (on the box): bash scripts/gpu_ab.sh <tag> "<pytest -k expr or empty>" variant1 variant2 ...   (base = the tree's libbhgeo.so)
tag=$1; shift; kexpr=$1; shift
if [ -n "$kexpr" ]; then
  timeout 1500 python -m pytest tests -q -m gpu --timeout 300 --maxfail=40 -k "$kexpr" > gpurun_out/pytest_gpu_$tag.log 2>&1; grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/pytest_gpu_$tag.log | tail -30
else
  timeout 1500 python -m pytest tests -q -m gpu --timeout 300 --maxfail=40 > gpurun_out/pytest_gpu_$tag.log 2>&1; grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/pytest_gpu_$tag.log | tail -30
fi
for w in "--workload frame" "--workload disk" "--workload orbit --steps 60 --warmup 5" "--workload frame --rhs kerr --steps 60 --warmup 5" "--workload disk --rhs kerr --steps 100 --warmup 10"; do
  echo "== $w"
  bash scripts/ab.sh "$w" "$@"
done 2>&1 | tee gpurun_out/ab_$tag.log
