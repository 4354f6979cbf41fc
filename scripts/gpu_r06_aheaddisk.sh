#!/bin/bash
# round 6: steps run ahead ALSO in the disk variants (-DBHG_AHEAD_DISK: 140 B of scratch per lane in <0,3>) against the tree
mkdir -p gpurun_out
out=gpurun_out/r06_aheaddisk.log
: > $out
for v in base aheaddisk; do
  echo "== bits $v" >> $out
  BHGEO_LIB=$PWD/build/variants/libbhgeo_$v.so timeout 600 python scripts/dev/dev_r06_bits.py disk frame orbit exit 2>&1 | grep -v amdgpu.ids >> $out
done
for w in "--workload disk" "--workload orbit --steps 60 --warmup 5"; do
  echo "== $w" >> $out
  bash scripts/ab.sh "$w" base aheaddisk >> $out 2>&1
done
cut -c1-160 $out
