"""The three calls of tests/test_gpu_parity.py::test_kerr_seeded_rays_and_rk4, with the rays that differ printed."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from blackhole_geodesic_calculator_amd import _ffi
from oracle import oracle as oc
cam = np.array([4.0, -24.0, 13.0])
rng = np.random.default_rng(41)
k = (-cam / np.linalg.norm(cam))[None, :] + rng.normal(size=(20000, 3)) * 0.12
k /= np.linalg.norm(k, axis=1)[:, None]
ctx = _ffi.Context(0)
for n, kw in ((20000, dict(r_s=1.0, lambda_end=60.0, rhs_form=2, spin=0.45)),
              (3000, dict(r_s=1.0, lambda_end=60.0, rhs_form=2, spin=-0.3, r_exit=35.0)),
              (2000, dict(r_s=1.0, lambda_end=40.0, rhs_form=2, spin=0.45, method=1, h_fixed=0.1))):
    o = oc.trace(k[:n], cam, **kw)
    end, flags, steps, acc = ctx.trace(k[:n], cam, _ffi.make_params(**kw))
    bad = np.nonzero(flags != o["flags"])[0]
    cnt = np.nonzero((steps != o["n_attempted"]) | (acc != o["n_accepted"]))[0]
    print(kw, "flag mismatches", len(bad), "count mismatches", len(cnt))
    for i in bad[:8]:
        print("  ray", i, "gpu", flags[i], steps[i], acc[i], end[i], "oracle", o["flags"][i], o["n_attempted"][i], o["n_accepted"][i], o["end"][i])
