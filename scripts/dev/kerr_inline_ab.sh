#!/bin/bash
# Kerr prepare inlined into the queue fill: parity tests of the Kerr paths, then A/B against the previous commit's library
timeout 1500 python -m pytest tests -q -m gpu --timeout 300 --maxfail=40 -k "kerr or Kerr" > gpurun_out/pytest_gpu_kerr_inline.log 2>&1
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/pytest_gpu_kerr_inline.log | tail -30
for w in "--workload frame --rhs kerr --steps 60 --warmup 5" "--workload disk --rhs kerr --steps 100 --warmup 10" "--workload frame"; do
  echo "== $w"
  bash scripts/ab.sh "$w" base prev
done 2>&1 | tee gpurun_out/ab_kerr_inline.log
