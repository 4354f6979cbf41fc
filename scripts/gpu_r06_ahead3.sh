#!/bin/bash
# round 6: steps run ahead in the disk variants too (rejected steps requeued from P): bits against -DBHG_NO_AHEAD, the checks build,
# the A/B on configs 3, 4 and the exit frame
mkdir -p gpurun_out
out=gpurun_out/r06_ahead3.log
: > $out
for v in noahead base; do
  echo "== bits $v" >> $out
  BHGEO_LIB=$PWD/build/variants/libbhgeo_$v.so timeout 600 python scripts/dev/dev_r06_bits.py frame disk diskkerr orbit exit exitkerr kerr 2>&1 | grep -v amdgpu.ids >> $out
done
echo "== BHG_CHECK build (any SLOT_CHECK / INV_CHECK line is a bug)" >> $out
BHGEO_LIB=$PWD/build/variants/libbhgeo_check.so timeout 600 python scripts/dev/dev_r06_bits.py disk orbit exit 2>&1 | grep -v amdgpu.ids | grep -E "CHECK|rays" | head -12 >> $out
for w in "--workload disk" "--workload orbit --steps 60 --warmup 5"; do
  echo "== $w" >> $out
  bash scripts/ab.sh "$w" noahead base >> $out 2>&1
done
echo "== exit frame, trace call" >> $out
for i in 1 2; do for v in noahead base; do
  BHGEO_LIB=$PWD/build/variants/libbhgeo_$v.so timeout 200 python scripts/dev/dev_r06_trace_time.py 3 200 exit 2>/dev/null | tail -1 >> $out
done; done
cut -c1-170 $out
