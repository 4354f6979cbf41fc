#!/bin/bash
# round 6, VERDICT task 2b: in-place resolution of wave-coherent short events (-DBHG_INPLACE_MIN=K builds) against the tree's
# own build on ONE box: bit identity first, then the bench A/B on config 3, Kerr + disk, config 4 and the headline
mkdir -p gpurun_out
out=gpurun_out/r06_inplace_${1:-a}.log
: > $out
for v in base inplace20 inplace32 inplace48; do
  echo "== bits $v" >> $out
  BHGEO_LIB=$PWD/build/variants/libbhgeo_$v.so timeout 300 python scripts/dev/dev_r06_bits.py frame disk diskkerr orbit exit >> $out 2>&1
done
for w in "--workload disk" "--workload disk --rhs kerr" "--workload orbit --steps 60 --warmup 5" "--workload frame"; do
  echo "== $w" >> $out
  bash scripts/ab.sh "$w" base inplace20 inplace32 inplace48 >> $out 2>&1
done
cat $out
