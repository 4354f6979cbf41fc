#!/bin/bash
# round 6: the Boyer-Lindquist trace kernels at 3 waves per SIMD (168 VGPRs, pool of 96 records) against 2 (the tree): bits, then A/B
mkdir -p gpurun_out
out=gpurun_out/r06_kerr3_${1:-a}.log
: > $out
for v in base kerr3; do
  echo "== bits $v" >> $out
  BHGEO_LIB=$PWD/build/variants/libbhgeo_$v.so timeout 300 python scripts/dev/dev_r06_bits.py kerr kerroff diskkerr 2>/dev/null >> $out
done
for w in "--rhs kerr --steps 60 --warmup 5" "--workload disk --rhs kerr --steps 100 --warmup 10"; do
  echo "== $w" >> $out
  bash scripts/ab.sh "$w" base kerr3 >> $out 2>&1
done
cut -c1-230 $out
