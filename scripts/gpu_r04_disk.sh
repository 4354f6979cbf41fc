#!/bin/bash
# round 4, task 1: the proven disk pre-filter -- parity (new library), the same grazing tests against round 3's library
# (they must FAIL there: that is the hole), and an A/B of the two libraries on the disk workloads
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu --timeout 600 --maxfail=20 -k "disk or grazing or randomised or golden" > gpurun_out/r04_disk_pytest.log 2>&1
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/r04_disk_pytest.log | tail -20
BHGEO_LIB=$PWD/build/variants/libbhgeo_r03.so timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu --timeout 600 -k "grazing" > gpurun_out/r04_disk_pytest_r03lib.log 2>&1
echo "-- round 3's library on the grazing tests:"; grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/r04_disk_pytest_r03lib.log | tail -12
for w in "--workload disk" "--workload disk --rhs kerr --steps 100 --warmup 10" "--workload frame"; do
  echo "== $w"
  bash scripts/ab.sh "$w" base r03 base r03
done 2>&1 | tee gpurun_out/r04_disk_ab.log
