#!/bin/bash
# evidence for every bench configuration in ONE GPU call, i.e. from ONE box: bash scripts/gpu_profiles_all.sh
# (outputs gpurun_out/prof_<name>_*: the plain bench line, the --lean line printed under rocprofv3 with the kernel trace
# reduced to its timed region, kernel stats, FETCH / WRITE / VALU counter passes; copy what is cited to profiles/rNN_*)
bash scripts/gpu_profile_config.sh frame_full          # the headline: whole end states (x, k) written
bash scripts/gpu_profile_config.sh frame_dir --dir-only   # the sky frame's direction-only form
bash scripts/gpu_profile_config.sh disk --workload disk
bash scripts/gpu_profile_config.sh orbit --workload orbit --steps 60 --warmup 5
bash scripts/gpu_profile_config.sh kerr --rhs kerr --steps 60 --warmup 5
bash scripts/gpu_profile_config.sh kerr_disk --workload disk --rhs kerr --steps 100 --warmup 10
for r in fine rk4; do
  timeout 600 python3 bench.py --regime $r --steps 20 --warmup 3 --lean 2>/dev/null | tail -1 > gpurun_out/prof_${r}_bench.json
done
timeout 600 python3 bench.py --rhs reduced --lean 2>/dev/null | tail -1 > gpurun_out/prof_reduced_bench.json
timeout 600 python3 bench.py --single-process --steps 100 --warmup 10 2>/dev/null | tail -1 > gpurun_out/prof_single_process_bench.json
BHGEO_DEVICES=0,0 timeout 600 python3 bench.py --single-process --gpus 2 --steps 100 --warmup 10 2>/dev/null | tail -1 > gpurun_out/prof_single_process_2ctx_bench.json
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/prof_*_timed_region.json")):
    t = json.load(open(f))
    print(f.split("prof_")[1].split("_timed")[0], "box", t["box"]["hostname"], "frac trace %.4f bench %.4f (%+.2f %%)" % (t["frac_from_kernel_trace"], t["frac_from_bench_line"], 100 * t["relative_difference"]))
PY
