import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from blackhole_geodesic_calculator_amd import _ffi
from oracle import oracle as oc
from conftest import frame_rays, CAM
ctx = _ffi.Context(0)
for seed, n in ((25, 600), (125, 20000)):
    k = frame_rays(n, seed=seed)
    for kw in (dict(max_step=0.1), dict(max_step=1e4), dict(rtol=1e-8, atol=1e-10), dict(rtol=1e-8, atol=1e-10, rhs_form=1), dict(rtol=1e-6, atol=1e-9)):
        kw = dict(r_s=1.0, lambda_end=50.0, **kw)
        o = oc.trace(k, CAM, **kw)
        end, flags, steps, acc = ctx.trace(k, CAM, _ffi.make_params(**kw))
        bad = np.nonzero((steps != o["n_attempted"]) | (flags != o["flags"]) | (acc != o["n_accepted"]))[0]
        d = np.abs(end - o["end"]).max(1)
        print(kw, "n", n, "mismatch", len(bad), "maxdiff %.3e" % d.max())
        for i in bad[:5]:
            print("   ray", i, "gpu steps/acc/flags", steps[i], acc[i], flags[i], "oracle", o["n_attempted"][i], o["n_accepted"][i], o["flags"][i], "diff %.3e" % d[i], "k0", k[i])
