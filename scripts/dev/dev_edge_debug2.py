"""Which rays of the edge-concentrated disk test differ in step counts, and how."""
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
for r_in, r_out in ((4.5, 17.0), (17.0, 35.0)):
    kw = dict(r_s=1.0, lambda_end=90.0, disk_r_in=r_in, disk_r_out=r_out, rtol=rtol, atol=rtol * 1e-3, rhs_form=1)
    o = oc.trace(k, cam, **kw)
    end, flags, steps, acc = ctx.trace(k, cam, _ffi.make_params(**kw))
    bad = np.nonzero((steps != o["n_attempted"]) | (acc != o["n_accepted"]) | (flags != o["flags"]))[0]
    print(r_in, r_out, "differing rays:", len(bad))
    for i in bad[:6]:
        print("  ray", i, "gpu flags/steps/acc", flags[i], steps[i], acc[i], "oracle", o["flags"][i], o["n_attempted"][i], o["n_accepted"][i],
              "end diff", np.abs(end[i] - o["end"][i]).max(), "R_end", np.hypot(*o["end"][i, :2]), "t_end", o["t_end"][i])
        # the same ray without the disk
        kw2 = dict(kw); kw2.pop("disk_r_in"); kw2.pop("disk_r_out")
        o2 = oc.trace(k[i:i+1], cam[i:i+1], **kw2)
        e2, f2, s2, a2 = ctx.trace(k[i:i+1], cam[i:i+1], _ffi.make_params(**kw2))
        print("     without disk: gpu", f2[0], s2[0], a2[0], "oracle", o2["flags"][0], o2["n_attempted"][0], o2["n_accepted"][0])
