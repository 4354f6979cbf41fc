"""Reproduce one draw of tests/test_gpu_parity.py::test_randomised_kerr and print the per-class differences."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from blackhole_geodesic_calculator_amd import _ffi
from oracle import oracle as oc
seed = int(sys.argv[1])
rng = np.random.default_rng(5000 + seed)
r_s = float(rng.choice([0.6, 1.0, 2.0]))
spin = float(rng.uniform(-0.98, 0.98)) * 0.5 * r_s
dist_cam = float(rng.uniform(6.0, 50.0)) * r_s
cam = rng.normal(size=3)
cam[2] *= 0.7
cam = dist_cam * cam / np.linalg.norm(cam)
if abs(cam[0]) + abs(cam[1]) < 0.05 * dist_cam:
    cam[0] += 0.2 * dist_cam
n = int(rng.integers(1, 2500))
aim = rng.normal(size=(n, 3)) * r_s * float(rng.uniform(1.0, 6.0))
k = aim - cam
k /= np.linalg.norm(k, axis=1)[:, None]
kw = dict(r_s=r_s, spin=spin, rhs_form=2, lambda_end=float(rng.uniform(1.0, 3.0)) * dist_cam)
mode = int(rng.integers(0, 3))
if mode == 0:
    kw.update(rtol=float(10 ** rng.uniform(-6, -2)), atol=float(10 ** rng.uniform(-9, -4)))
elif mode == 1:
    kw.update(max_step=float(rng.uniform(0.1, 2.0)) * r_s)
if rng.random() < 0.4:
    kw["r_exit"] = float(rng.uniform(0.6, 1.4)) * dist_cam
if rng.random() < 0.2:
    kw["max_steps"] = int(rng.integers(1, 60))
ctx = _ffi.Context(0)
o = oc.trace(k, cam, **kw)
end, flags, steps, acc = ctx.trace(k, cam, _ffi.make_params(**kw))
same = (steps == o["n_attempted"]) & (acc == o["n_accepted"]) & (flags == o["flags"])
d = np.abs(end - o["end"]).max(1)
print("n", n, kw, "same", same.mean())
for f in np.unique(o["flags"]):
    m = (o["flags"] == f) & same
    if m.any():
        print("flag", f, "count", m.sum(), "median d", np.median(d[m]), "max d", d[m].max(), "median |end|", np.median(np.abs(o["end"][m]).max(1)))
