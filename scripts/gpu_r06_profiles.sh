#!/bin/bash
# round 6: one lease = the driver-command A/B (third lease of it) + the evidence set of every bench configuration
bash scripts/gpu_r06_driver_cmd.sh ${1:-g} > /dev/null 2>&1
cut -c1-200 gpurun_out/r06_driver_cmd_${1:-g}.log
bash scripts/gpu_profiles_all.sh 2>&1 | tail -60
timeout 600 python3 bench.py > gpurun_out/prof_default_line.log 2>&1; tail -1 gpurun_out/prof_default_line.log > gpurun_out/prof_default_line.json
