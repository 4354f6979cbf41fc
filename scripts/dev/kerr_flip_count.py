"""How many Kerr rays change flag / step count against the CPU checker (dev aid; BHGEO_LIB selects the library)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import oracle as oc
from blackhole_geodesic_calculator_amd import _ffi
oc.build()
ctx = _ffi.Context(0)
def run(cam, k, **kw):
    o = oc.trace(k, cam, **kw)
    end, flags, steps, acc = ctx.trace(k, cam, _ffi.make_params(**kw))
    same = (flags == o["flags"]) & (steps == o["n_attempted"])
    d = np.abs(end - o["end"]).max(1)
    print("  n %d  flag diffs %d  step diffs %d  median |d| %.2e  q99 %.2e  max(same) %.2e" % (
        len(k), (flags != o["flags"]).sum(), (steps != o["n_attempted"]).sum(), np.median(d[same]), np.quantile(d[same], 0.99), d[same].max()))
print(os.environ.get("BHGEO_LIB", "base"))
cam = np.array([4.0, -24.0, 13.0]); rng = np.random.default_rng(41)
k = (-cam / np.linalg.norm(cam))[None, :] + rng.normal(size=(20000, 3)) * 0.12; k /= np.linalg.norm(k, axis=1)[:, None]
run(cam, k, r_s=1.0, lambda_end=60.0, rhs_form=2, spin=0.45)
cam = np.array([1e-4, 0.0, 30.0]); rng = np.random.default_rng(5)
k = np.array([0, 0, -1.0])[None, :] + rng.normal(size=(20000, 3)) * 0.2; k /= np.linalg.norm(k, axis=1)[:, None]
run(cam, k, r_s=1.0, lambda_end=50.0, rhs_form=2, spin=0.45)
cam = np.array([0.0, -30.0, 0.5]); rng = np.random.default_rng(6)
k = (-cam / np.linalg.norm(cam))[None, :] + rng.normal(size=(20000, 3)) * 0.2; k /= np.linalg.norm(k, axis=1)[:, None]
run(cam, k, r_s=1.0, lambda_end=60.0, rhs_form=2, spin=0.45, disk_r_in=3.0, disk_r_out=12.0)
sys.path.insert(0, "tests")
from test_gpu_parity import _grazing_rays
k, cam = _grazing_rays(6000, 77, sigma=(9.0, 9.0, 0.1), inc_lo=75.0)
cam = cam + np.array([0.0, 3.0, 0.0])
for rtol in (1e-3, 1e-2):
    kw = dict(r_s=1.0, spin=0.45, rhs_form=2, lambda_end=90.0, disk_r_in=3.0, disk_r_out=35.0, rtol=rtol, atol=rtol * 1e-3)
    o = oc.trace(k, cam, **kw)
    end, flags, steps, acc = ctx.trace(k, cam, _ffi.make_params(**kw))
    same = (flags == o["flags"]) & (steps == o["n_attempted"]) & (o["flags"] == 128)
    d = np.abs(end[same] - o["end"][same]).max(1)
    steep = np.abs(o["end"][same, 5]) / np.linalg.norm(o["end"][same, 3:6], axis=1)
    print("  grazing rtol %g: same %d  d: median %.1e q99 %.1e q999 %.1e max %.1e | d*steep: q99 %.1e max %.1e" % (
        rtol, same.sum(), np.median(d), np.quantile(d, 0.99), np.quantile(d, 0.999), d.max(), np.quantile(d * steep, 0.99), (d * steep).max()))
