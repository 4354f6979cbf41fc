#!/bin/bash
# round 6 lease d: the driver-command A/B on the fixed tree, then the new / changed GPU tests with their printed census lines
tag=${1:-d}
bash scripts/gpu_r06_driver_cmd.sh $tag > /dev/null 2>&1
cut -c1-260 gpurun_out/r06_driver_cmd_$tag.log
timeout 1500 python -m pytest tests/test_gpu_parity.py::test_trajectories_with_object_spheres tests/test_gpu_parity.py::test_kerr_object_spheres_golden_and_oracle \
  tests/test_gpu_fullsize.py tests/test_gpu_adaptors.py tests/test_gpu_frame_object.py tests/test_gpu_threads.py tests/test_gpu_lifecycle.py tests/test_integration_stub.py \
  -q -m gpu -s --timeout 900 -x > gpurun_out/r06_newtests_$tag.log 2>&1
grep -E "config 5|Kerr a/M|T2|passed|failed|FAILED|Error|error|assert" gpurun_out/r06_newtests_$tag.log | cut -c1-900 | tail -40
