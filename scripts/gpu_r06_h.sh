#!/bin/bash
# round 6 lease h: (1) upper bound of what packing the three narrow result stores could save (variant without them), same box;
# (2) a larger randomised run of the parity suite with round 6's stricter gates (BHG_FUZZ=2000)
mkdir -p gpurun_out
{ echo "== --workload frame: base vs nonarrow (flags / n_steps / n_accepted not stored at all: an UPPER BOUND, not a product build)"
  bash scripts/ab.sh "--workload frame" base nonarrow base nonarrow
  echo "== --workload frame --dir-only"
  bash scripts/ab.sh "--workload frame --dir-only" base nonarrow; } > gpurun_out/r06_narrow_stores_ab.log 2>&1
cat gpurun_out/r06_narrow_stores_ab.log | cut -c1-160
BHG_FUZZ=2000 timeout 2400 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "fuzz or random" -s --timeout 2000 > gpurun_out/r06_fuzz2000.log 2>&1
grep -E "rounding flips|passed|failed|FAILED|Error" gpurun_out/r06_fuzz2000.log | tail -12 | cut -c1-300
grep -c "kerr fuzz" gpurun_out/r06_fuzz2000.log
python3 - <<'PY'
import re, ast
rows = [ast.literal_eval(l.split(" ", 3)[3]) for l in open("gpurun_out/r06_fuzz2000.log") if l.startswith("kerr fuzz")]
if rows:
    worst_h = max((r["differ_horizon"] / max(r["horizon_rays"], 1), r["differ_horizon"], r["horizon_rays"]) for r in rows if r["differ_horizon"] > 3)  if any(r["differ_horizon"] > 3 for r in rows) else None
    print("kerr draws", len(rows), "rays", sum(r["rays"] for r in rows), "differ", sum(r["differ"] for r in rows), "horizon", sum(r["differ_horizon"] for r in rows),
          "other", sum(r["differ_other"] for r in rows), "neither", sum(r["differ_neither"] for r in rows), "worst horizon fraction (draws with > 3)", worst_h)
PY
