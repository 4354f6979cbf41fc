#!/bin/bash
# round 6: the GCN scheduler's max-ILP strategy (-mllvm -amdgpu-sched-strategy=max-ilp) against the default, same box
mkdir -p gpurun_out
out=gpurun_out/r06_sched_ab.log
: > $out
for v in base maxilp; do
  echo "== bits $v" >> $out
  BHGEO_LIB=$PWD/build/variants/libbhgeo_$v.so timeout 300 python scripts/dev/dev_r06_bits.py frame disk kerr 2>/dev/null >> $out
done
for w in "--workload frame" "--workload disk" "--rhs kerr --steps 60 --warmup 5"; do
  echo "== $w" >> $out
  bash scripts/ab.sh "$w" base maxilp >> $out 2>&1
done
cut -c1-200 $out
