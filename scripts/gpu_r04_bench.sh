#!/bin/bash
# round 4: the refactored bench.py -- its GPU tests, the tests that failed in the last call, then headline lines
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_rccl.py tests/test_gpu_adaptors.py tests/test_gpu_frame_object.py -q -m gpu --timeout 900 -x > gpurun_out/r04_bench_pytest.log 2>&1
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/r04_bench_pytest.log | tail -20
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu --timeout 600 -k "keeps_every_grazing" 2>&1 | tail -3
timeout 600 python bench.py --steps 100 --warmup 10 > gpurun_out/r04_bench_default.log 2>&1; tail -1 gpurun_out/r04_bench_default.log | cut -c1-1500
timeout 300 python bench.py --single-process --steps 100 --warmup 10 2>&1 | tail -1 | cut -c1-1200
BHGEO_DEVICES=0,0 timeout 300 python bench.py --single-process --gpus 2 --steps 100 --warmup 10 2>&1 | tail -1 | cut -c1-1600
