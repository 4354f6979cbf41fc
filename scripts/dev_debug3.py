import sys, os, math
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blackhole_geodesic_calculator_amd import _ffi
from oracle import oracle as oc
ctx = _ffi.Context(0)
cam = np.array([4.0, -24.0, 13.0])
rng = np.random.default_rng(41)
k = (-cam / np.linalg.norm(cam))[None, :] + rng.normal(size=(20000, 3)) * 0.12
k /= np.linalg.norm(k, axis=1)[:, None]
kw = dict(r_s=1.0, lambda_end=60.0, rhs_form=2, spin=0.45)
o = oc.trace(k, cam, **kw)
end, flags, steps, acc = ctx.trace(k, cam, _ffi.make_params(**kw))
print("flags eq", (flags == o["flags"]).mean(), "steps eq", (steps == o["n_attempted"]).mean(), "acc eq", (acc == o["n_accepted"]).mean())
d = np.abs(end - o["end"]).max(1)
s = np.abs(oc.trace(np.nextafter(k, np.inf), cam, **kw)["end"] - o["end"]).max(1)
ratio = d / (1e-9 + s)
print("d percentiles", np.percentile(d, [50, 90, 99, 99.9, 100]))
print("s percentiles", np.percentile(s, [50, 90, 99, 99.9, 100]))
print("ratio percentiles", np.percentile(ratio, [50, 90, 99, 99.9, 100]))
hz = (flags & 1) != 0
print("horizon frac", hz.mean(), "d max horizon %.3e escaping %.3e" % (d[hz].max(), d[~hz].max()))
i = np.argmax(ratio); print(i, d[i], s[i], flags[i], steps[i], end[i], o["end"][i])
import time
t = time.time(); ctx.trace(k, cam, _ffi.make_params(**kw)); print("gpu host call %.1f ms" % ((time.time() - t) * 1e3))
