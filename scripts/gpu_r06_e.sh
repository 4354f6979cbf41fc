#!/bin/bash
# round 6 lease e: the in-place A/B (task 2b), then the GPU tests lease d did not reach
bash scripts/gpu_r06_inplace.sh e > /dev/null 2>&1
cat gpurun_out/r06_inplace_e.log | cut -c1-200
timeout 1500 python -m pytest tests/test_gpu_parity.py::test_trajectories_with_object_spheres "tests/test_gpu_fullsize.py::test_kerr_near_extremal_frame_full_size" \
  tests/test_gpu_adaptors.py tests/test_gpu_frame_object.py tests/test_gpu_threads.py tests/test_gpu_lifecycle.py tests/test_integration_stub.py tests/test_gpu_errors.py \
  -q -m gpu -s --timeout 900 > gpurun_out/r06_newtests_e.log 2>&1
grep -E "Kerr a/M|passed|failed|FAILED|Error|error|assert" gpurun_out/r06_newtests_e.log | cut -c1-1200 | tail -30
