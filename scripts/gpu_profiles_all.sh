#!/bin/bash
# evidence for every bench configuration in one GPU call: bash scripts/gpu_profiles_all.sh  (outputs gpurun_out/prof_<name>_*)
bash scripts/gpu_profile_config.sh frame
bash scripts/gpu_profile_config.sh disk --workload disk
bash scripts/gpu_profile_config.sh orbit --workload orbit --steps 60 --warmup 5
bash scripts/gpu_profile_config.sh kerr --rhs kerr --steps 60 --warmup 5
bash scripts/gpu_profile_config.sh kerr_disk --workload disk --rhs kerr --steps 100 --warmup 10
for r in fine rk4; do
  timeout 600 python3 bench.py --regime $r --steps 20 --warmup 3 --lean 2>/dev/null | tail -1 > gpurun_out/prof_${r}_bench.json
done
timeout 600 python3 bench.py --rhs reduced --lean 2>/dev/null | tail -1 > gpurun_out/prof_reduced_bench.json
