#!/bin/bash
# round 6: the ahead steps in the Kerr + disk variants too: bits against -DBHG_NO_AHEAD, checks build, A/B, then the whole suite and the Kerr draws
mkdir -p gpurun_out
out=gpurun_out/r06_ahead_kerr_ab.log
: > $out
for v in noahead base; do
  echo "== bits $v" >> $out
  BHGEO_LIB=$PWD/build/variants/libbhgeo_$v.so timeout 600 python scripts/dev/dev_r06_bits.py diskkerr exitkerr orbitkerr kerr disk orbit 2>&1 | grep -v amdgpu.ids >> $out
done
echo "== BHG_CHECK build" >> $out
BHGEO_LIB=$PWD/build/variants/libbhgeo_check.so timeout 600 python scripts/dev/dev_r06_bits.py diskkerr 2>&1 | grep -v amdgpu.ids | grep -E "CHECK|rays" | head >> $out
echo "== --workload disk --rhs kerr" >> $out
bash scripts/ab.sh "--workload disk --rhs kerr --steps 100 --warmup 10" noahead base >> $out 2>&1
cut -c1-150 $out
timeout 1500 python -m pytest tests -q -m gpu --timeout 900 2>&1 | tail -3
BHG_FUZZ=2000 timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "kerr" --timeout 1400 2>&1 | tail -2
timeout 600 python scripts/dev/dev_kerr_every_ray_sweep.py r06c 2>&1 | tail -1 | cut -c1-160
