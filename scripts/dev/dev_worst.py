"""Development aid: rerun one fuzz draw (tests/test_gpu_parity.py::test_randomised_configurations[seed]) and print the
rays whose end state differs most from the oracle.  usage: dev_worst.py <seed>"""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from blackhole_geodesic_calculator_amd import _ffi
from oracle import oracle as oc
oc.build()
seed = int(sys.argv[1])
rng = np.random.default_rng(1000 + seed)
r_s = float(rng.choice([0.0, 0.3, 1.0, 2.5]))
dist_cam = float(rng.uniform(3.0, 60.0)) * max(r_s, 0.5)
cam = rng.normal(size=3)
cam = dist_cam * cam / np.linalg.norm(cam)
n = int(rng.integers(1, 3000))
aim = rng.normal(size=(n, 3)) * max(r_s, 0.5) * float(rng.uniform(1.0, 8.0))
k = aim - cam
k /= np.linalg.norm(k, axis=1)[:, None]
x0 = cam + rng.normal(size=(n, 3)) * 0.1 * dist_cam if rng.random() < 0.3 else cam
kw = dict(r_s=r_s, lambda_end=float(rng.uniform(0.5, 4.0)) * dist_cam, rhs_form=int(rng.integers(0, 2)))
mode = int(rng.integers(0, 4))
if mode == 0:
    kw.update(rtol=float(10 ** rng.uniform(-7, -2)), atol=float(10 ** rng.uniform(-10, -4)))
elif mode == 1:
    kw.update(max_step=float(rng.uniform(0.05, 2.0)) * max(r_s, 0.5))
elif mode == 2:
    kw.update(method=1, h_fixed=float(rng.uniform(0.05, 0.5)) * max(r_s, 0.5))
if rng.random() < 0.4:
    kw["r_exit"] = float(rng.uniform(0.5, 1.5)) * dist_cam
if rng.random() < 0.4:
    a = float(rng.uniform(1.5, 6.0)) * max(r_s, 0.5)
    kw.update(disk_r_in=a, disk_r_out=a * float(rng.uniform(1.1, 3.0)))
if rng.random() < 0.2:
    kw["max_steps"] = int(rng.integers(1, 40))
print("draw", seed, "n", n, kw, "cam", cam)
ctx = _ffi.Context(0)
o = oc.trace(k, x0, **kw)
end, flags, steps, acc = ctx.trace(k, x0, _ffi.make_params(**kw))
print("flags equal", np.array_equal(flags, o["flags"]), "steps equal", np.array_equal(steps, o["n_attempted"]))
d = np.abs(end - o["end"]).max(1)
d = np.where(np.isfinite(d), d, np.inf)
for i in np.argsort(-d)[:8]:
    print(i, "d %.3e" % d[i], "flags gpu/oracle", flags[i], o["flags"][i], "steps", steps[i], o["n_attempted"][i])
    print("   gpu   ", end[i])
    print("   oracle", o["end"][i])
