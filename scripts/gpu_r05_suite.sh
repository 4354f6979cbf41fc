#!/bin/bash
# round 5: the whole GPU suite with its printed census lines kept (-s), then the default bench line
tag=${1:-r05}
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu -s --timeout 900 --durations=12 > gpurun_out/${tag}_suite_full.log 2>&1
grep -E "every ray|config [2345]|Kerr off-axis|T2:|passed|failed|FAILED|Error" gpurun_out/${tag}_suite_full.log | cut -c1-700 | tail -60
timeout 600 python bench.py > gpurun_out/${tag}_bench.log 2>&1; tail -1 gpurun_out/${tag}_bench.log > gpurun_out/${tag}_bench.json
python - <<PY
import json
d = json.load(open("gpurun_out/${tag}_bench.json"))
r = d["roofline"]
print("value", d["value"], "ms", d["ms_per_step"], "frac", r["frac"], "frac_of_measured_peak", r.get("frac_of_measured_peak"))
print(json.dumps(r.get("calibration"), indent=1)[:3000])
print({k: (v.get("value"), v.get("frac")) for k, v in d.items() if isinstance(v, dict) and "value" in v})
print(d["config"].get("north_star_output"), d["config"]["trace_output"])
PY
