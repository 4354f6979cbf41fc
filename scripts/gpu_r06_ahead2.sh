#!/bin/bash
# round 6: the tree with steps run ahead (the default now): slot / invariant checks build on the event workloads, the whole GPU
# suite, smoke, then the A/B against -DBHG_NO_AHEAD on config 4 and the evidence set of config 4
mkdir -p gpurun_out
echo "== BHG_CHECK build: any SLOT_CHECK / INV_CHECK line is a bug"
BHGEO_LIB=$PWD/build/variants/libbhgeo_check.so timeout 600 python scripts/dev/dev_r06_bits.py orbit exit disk frame 2>&1 | grep -v amdgpu.ids | grep -E "CHECK|rays" | head -20
timeout 1500 python -m pytest tests -q -m gpu --timeout 900 -x 2>&1 | tail -5
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "== A/B orbit: noahead vs the tree"
bash scripts/ab.sh "--workload orbit --steps 60 --warmup 5" noahead base 2>&1 | cut -c1-120
bash scripts/gpu_profile_config.sh orbit --workload orbit --steps 60 --warmup 5 2>&1 | tail -8
