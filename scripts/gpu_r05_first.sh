#!/bin/bash
# round 5, first lease: the new tests first (their census numbers are wanted even if something else fails), the probes,
# then the whole GPU suite and the default bench line
mkdir -p gpurun_out
timeout 300 python scripts/dev/dev_probe.py > gpurun_out/r05_probe.json 2> gpurun_out/r05_probe.err; tail -3 gpurun_out/r05_probe.err
timeout 1500 python -m pytest tests/test_integration_stub.py tests/test_gpu_fullsize.py "tests/test_gpu_parity.py::test_full_frame_every_ray_matches_oracle" tests/test_gpu_adaptors.py -q -m gpu -s --timeout 600 2>&1 | grep -v "^$" | tail -60 > gpurun_out/r05_newtests.log
tail -25 gpurun_out/r05_newtests.log
timeout 1500 python -m pytest tests -q -m gpu --timeout 600 --durations=15 2>&1 | tail -40 > gpurun_out/r05_suite.log
tail -8 gpurun_out/r05_suite.log
timeout 400 python bench.py 2>&1 | tail -1 > gpurun_out/r05_bench_first.json
cut -c1-600 gpurun_out/r05_bench_first.json
