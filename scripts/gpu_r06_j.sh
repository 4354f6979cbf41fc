#!/bin/bash
# round 6 lease j: the every-ray sweeps of round 5 again on the round-6 tree (20 full-size frames against the checker), the new
# destroyed-stream test, smoke()
mkdir -p gpurun_out
timeout 900 python scripts/dev/dev_every_ray_sweep.py r06 2>&1 | tail -3 | cut -c1-400
timeout 900 python scripts/dev/dev_kerr_every_ray_sweep.py r06 2>&1 | tail -2 | cut -c1-400
timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -k "stream" --timeout 300 2>&1 | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
