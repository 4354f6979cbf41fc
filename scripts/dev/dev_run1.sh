BHG_FUZZ=240 timeout 2400 python -m pytest tests/test_gpu_parity.py -q -m gpu --timeout 900 -k "randomised" -s 2>&1 | grep -E "fuzz draw|passed|failed" > gpurun_out/r04_fuzz240.log
tail -2 gpurun_out/r04_fuzz240.log
python3 - <<'PY'
import re, ast
rows=[]
for l in open('gpurun_out/r04_fuzz240.log'):
    m=re.match(r'fuzz draw (\d+): (\{.*\})', l.strip())
    if m: rows.append((int(m.group(1)), ast.literal_eval(m.group(2))))
import numpy as np
w=np.array([r[1]['worst_multiple_of_sensitivity'] for r in rows]); bf=np.array([r[1]['beyond_floor'] for r in rows]); bb=np.array([r[1]['beyond_bound'] for r in rows]); rays=np.array([r[1]['rays'] for r in rows]); worst=np.array([r[1]['worst'] for r in rows])
print(len(rows),'draws; worst multiple of S_i: max %.3g, 99th pct %.3g, median %.3g; draws with any ray beyond floor: %d; beyond_bound total %d; worst |d| %.3g' % (w.max(), np.quantile(w,0.99), np.median(w), (bf>0).sum(), bb.sum(), worst.max()))
for i in np.argsort(-w)[:8]: print(rows[i])
PY
