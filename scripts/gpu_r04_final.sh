#!/bin/bash
# the round's closing GPU call: the whole GPU suite, then every configuration's evidence from this one box
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -q -m gpu --timeout 900 > gpurun_out/r04_pytest_gpu.log 2>&1
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/r04_pytest_gpu.log | tail -12
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash scripts/gpu_profiles_all.sh 2>&1 | tee gpurun_out/r04_profiles_all.log | grep -vE "^\s+void|^\s+bhg::" 
