import sys, os, math
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blackhole_geodesic_calculator_amd import _ffi
from oracle import oracle as oc
ctx = _ffi.Context(0)
inc = math.radians(80.0)
cam = np.array([30 * math.sin(inc), 0.0, 30 * math.cos(inc)])
aim = np.random.default_rng(80).normal(size=(20000, 3)) * np.array([9.0, 9.0, 1.0])
d = aim - cam
k = d / np.linalg.norm(d, axis=1)[:, None]
kw = dict(r_s=1.0, lambda_end=80.0, disk_r_in=4.5, disk_r_out=10.5)
o = oc.trace(k, cam, **kw)
end, flags, steps, acc = ctx.trace(k, cam, _ffi.make_params(**kw))
dd = np.abs(end - o["end"]).max(1)
kp = np.nextafter(k, np.inf)
s = np.abs(oc.trace(kp, cam, **kw)["end"] - o["end"]).max(1)
for i in np.argsort(dd)[-6:]:
    print(i, "diff %.3e sens %.3e" % (dd[i], s[i]), "flags", flags[i], o["flags"][i], "steps", steps[i], o["n_attempted"][i], "acc", acc[i], o["n_accepted"][i], "t_end", o["t_end"][i])
    print("   gpu", end[i]); print("   ora", o["end"][i])
print(ctx.last_launch())
