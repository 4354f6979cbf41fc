#!/bin/bash
# quick GPU round for kernel work: parity tests, three benches (no CPU baseline), per-wave diagnostics
# usage (on the box): bash scripts/gpu_quick.sh <tag> [pytest-args]
tag=$1; shift
timeout 1200 python -m pytest tests -x -q -m gpu --timeout 180 "$@" 2>&1 | tail -8 | tee gpurun_out/pytest_gpu_$tag.log
for w in "frame" "disk" "orbit" "frame --rhs kerr" "disk --rhs kerr"; do
  n=$(echo $w | tr -d ' -')
  timeout 300 python bench.py --workload $w --cpu-seconds 0 2>&1 | tail -1 > gpurun_out/bench_${tag}_$n.json
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/bench_${tag}_$n.json"))
    print("$w:", "ms/step %.3f" % d["ms_per_step"], "kernel_ms %.3f" % d["roofline"]["kernel_ms"], "Mrays/s %.0f" % d["value"], "frac %.3f" % d["roofline"]["frac"])
except Exception as e:
    print("$w: FAILED", e)
PY
done
if [ -f build/variants/libbhgeo_diag.so ]; then
for w in frame exit disk orbit; do
  BHGEO_LIB=$PWD/build/variants/libbhgeo_diag.so timeout 120 python scripts/dev/dev_diag_run.py $w gpurun_out/diag_$w.bin > /dev/null 2>&1
  echo "== $w"; python scripts/diag_analyze.py gpurun_out/diag_$w.bin | tail -3
done
fi
