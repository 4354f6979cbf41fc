#!/bin/bash
# round 6, final tree (steps run ahead): the evidence set of every bench configuration from ONE lease + the default line + the driver's command
bash scripts/gpu_profiles_all.sh 2>&1 | tail -62
timeout 600 python3 bench.py > gpurun_out/prof_default_line.log 2>&1; tail -1 gpurun_out/prof_default_line.log > gpurun_out/prof_default_line.json
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/prof_driver_cmd_line.json
python3 scripts/r06_line_summary.py driver_cmd < gpurun_out/prof_driver_cmd_line.json | cut -c1-330
