#!/bin/bash
# round 6: steps run AHEAD by the short drain (-DBHG_AHEAD) against the tree: bits on every workload, then the A/B
mkdir -p gpurun_out
out=gpurun_out/r06_ahead_${1:-a}.log
: > $out
for v in base ahead; do
  echo "== bits $v" >> $out
  BHGEO_LIB=$PWD/build/variants/libbhgeo_$v.so timeout 600 python scripts/dev/dev_r06_bits.py frame disk diskkerr orbit exit exitkerr orbitkerr kerr 2>&1 | grep -v amdgpu.ids >> $out
done
echo "== exit frame (r_exit 40), trace call" >> $out
for i in 1 2; do for v in base ahead; do
  BHGEO_LIB=$PWD/build/variants/libbhgeo_$v.so timeout 200 python scripts/dev/dev_r06_trace_time.py 3 200 exit 2>/dev/null | tail -1 >> $out
done; done
echo "== Kerr exit frame, trace call" >> $out
for v in base ahead; do
  BHGEO_LIB=$PWD/build/variants/libbhgeo_$v.so timeout 300 python scripts/dev/dev_r06_trace_time.py 3 40 exitkerr 2>/dev/null | tail -1 >> $out
done
for w in "--workload orbit --steps 60 --warmup 5" "--workload disk" "--workload frame"; do
  echo "== $w" >> $out
  bash scripts/ab.sh "$w" base ahead >> $out 2>&1
done
cut -c1-200 $out
