#!/bin/bash
# round 4, task 2: the library-owned frame -- its tests, the add-on tests, the edge-concentrated disk test on this tree's
# library and (must FAIL) on a build of this tree with round 3's filter figure
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_frame_object.py tests/test_gpu_adaptors.py tests/test_host.py -q -m "gpu or not gpu" --timeout 600 > gpurun_out/r04_frame_pytest.log 2>&1
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/r04_frame_pytest.log | tail -20
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu --timeout 600 -k "keeps_every_grazing" > gpurun_out/r04_edge_pytest.log 2>&1
echo "-- edge-concentrated disk test, this tree:"; grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/r04_edge_pytest.log | tail -5
BHGEO_LIB=$PWD/build/variants/libbhgeo_r03filter.so timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu --timeout 600 -k "keeps_every_grazing" > gpurun_out/r04_edge_pytest_r03filter.log 2>&1
echo "-- the same with round 3's filter figure (must fail):"; grep -E "^FAILED|^ERROR|passed|failed|disk hits lost" gpurun_out/r04_edge_pytest_r03filter.log | tail -8
