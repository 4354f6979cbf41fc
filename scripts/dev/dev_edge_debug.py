"""Worst end-state differences of the edge-concentrated disk test (tests/test_gpu_parity.py), per ray."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from blackhole_geodesic_calculator_amd import _ffi
from oracle import oracle as oc
ctx = _ffi.Context(0)
rtol = float(sys.argv[1]) if len(sys.argv) > 1 else 3e-2
n = 400000
rng = np.random.default_rng(int(rtol * 1e3))
inc = np.deg2rad(rng.uniform(89.7, 89.999, n))
cam = 30.0 * np.stack([np.sin(inc), np.zeros(n), np.cos(inc)], -1)
ph, R = rng.uniform(0.0, 2.0 * np.pi, n), 17.0 + rng.uniform(-0.3, 0.3, n)
k = np.stack([R * np.cos(ph), R * np.sin(ph), rng.normal(0.0, 0.01, n)], -1) - cam
k /= np.linalg.norm(k, axis=1)[:, None]
kw = dict(r_s=1.0, lambda_end=90.0, disk_r_in=4.5, disk_r_out=17.0, rtol=rtol, atol=rtol * 1e-3, rhs_form=1)
o = oc.trace(k, cam, **kw)
end, flags, steps, acc = ctx.trace(k, cam, _ffi.make_params(**kw))
hit = np.nonzero(flags == 128)[0]
d = np.abs(end[hit] - o["end"][hit]).max(1)
steep = np.abs(o["end"][hit, 5]) / np.linalg.norm(o["end"][hit, 3:6], axis=1)
worst = np.argsort(-d * steep)[:8]
for w in worst:
    i = hit[w]
    print(i, "d", d[w], "steep", steep[w], "d*steep", d[w] * steep[w], "steps", steps[i], "t_end", o["t_end"][i])
    print("   gpu ", end[i]); print("   orac", o["end"][i])
print("quantiles of d*steep:", np.quantile(d * steep, [0.5, 0.99, 0.9999, 1.0]))
