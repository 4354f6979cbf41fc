"""Round 5: flags / attempted / accepted step counts of the device against the checker on EVERY ray of a sweep of full-size
Schwarzschild frames (1024 x 1024 x 5 each) -- cameras near and far, on and off the axis, narrow and wide fields of view,
other masses, tolerances, step caps, exit sphere, both right-hand-side forms -> gpurun_out/<tag>_every_ray_sweep.json (tag = argv[1], default r05)"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from blackhole_geodesic_calculator_amd import _ffi
from blackhole_geodesic_calculator_amd.device_frame import DeviceFrame
from oracle import oracle as oc

oc.build()
ctx = _ffi.Context(0)
cases = [
    dict(name="config 2", cam=(1e-4, 0.0, 30.0), euler=(0, 0, 0), fov=0.6, kw=dict(r_s=1.0, lambda_end=50.0)),
    dict(name="config 2, reduced RHS", cam=(1e-4, 0.0, 30.0), euler=(0, 0, 0), fov=0.6, kw=dict(r_s=1.0, lambda_end=50.0, rhs_form=1)),
    dict(name="near camera r=8, wide", cam=(0.5, -0.3, 8.0), euler=(0.02, -0.03, 0.1), fov=1.6, kw=dict(r_s=1.0, lambda_end=40.0)),
    dict(name="far camera r=120, narrow", cam=(3.0, 2.0, 120.0), euler=(0, 0, 0), fov=0.12, kw=dict(r_s=1.0, lambda_end=260.0)),
    dict(name="inclined 70 deg, exit sphere 40", cam=(30 * np.sin(1.2217), 0.0, 30 * np.cos(1.2217)), euler=(0.0, 1.2217, 0.0), fov=0.9,
         kw=dict(r_s=1.0, lambda_end=80.0, r_exit=40.0)),
    dict(name="mass 1.25 (r_s 2.5)", cam=(1.0, 1.0, 45.0), euler=(0, 0, 0), fov=0.7, kw=dict(r_s=2.5, lambda_end=100.0)),
    dict(name="rtol 1e-5", cam=(1e-4, 0.0, 30.0), euler=(0, 0, 0), fov=0.6, kw=dict(r_s=1.0, lambda_end=50.0, rtol=1e-5, atol=1e-8, rhs_form=1)),
    dict(name="rtol 1e-2", cam=(1e-4, 0.0, 30.0), euler=(0, 0, 0), fov=0.6, kw=dict(r_s=1.0, lambda_end=50.0, rtol=1e-2, atol=1e-5)),
    dict(name="max_step 1.0", cam=(1e-4, 0.0, 30.0), euler=(0, 0, 0), fov=0.6, kw=dict(r_s=1.0, lambda_end=50.0, max_step=1.0)),
    dict(name="zoom on the shadow edge", cam=(1e-4, 0.0, 30.0), euler=(0.0, 0.0866, 0.0), fov=0.05, kw=dict(r_s=1.0, lambda_end=60.0)),
    dict(name="flat space r_s=0", cam=(1.0, 2.0, 30.0), euler=(0, 0, 0), fov=0.6, kw=dict(r_s=0.0, lambda_end=50.0)),
    dict(name="disk 3..12 from 80 deg", cam=(30 * np.sin(1.3963), 0.0, 30 * np.cos(1.3963)), euler=(0.0, 1.3963, 0.0), fov=0.9,
         kw=dict(r_s=1.0, lambda_end=80.0, r_exit=40.0, disk_r_in=3.0, disk_r_out=12.0)),
]
out = []
for c in cases:
    fr = DeviceFrame(ctx, 1024, 1024, 5, fov_x=c["fov"], fov_y=c["fov"], origin=c["cam"], rotation_euler=c["euler"])
    fr.generate_rays()
    k0 = fr.d_k0.cpu().numpy()
    cam = np.asarray(c["cam"], float)
    t = time.time()
    end, flags, steps, acc = ctx.trace(k0, cam, _ffi.make_params(**c["kw"]))
    tg = time.time() - t
    t = time.time()
    o = oc.trace(k0, cam, **c["kw"])
    to = time.time() - t
    fbad = flags != o["flags"]
    sbad = steps != o["n_attempted"]
    abad = acc != o["n_accepted"]
    d = np.abs(end - o["end"]).max(1)
    esc = ((flags == 4) | (flags == 8)) & ~fbad & ~sbad
    rec = dict(name=c["name"], rays=int(len(k0)), flag_diff=int(fbad.sum()), attempted_diff=int(sbad.sum()), accepted_diff=int(abad.sum()),
               steps_per_ray=float(steps.mean()), horizon_fraction=float(((flags & 1) != 0).mean()),
               census={int(f): int(n) for f, n in zip(*np.unique(flags, return_counts=True))},
               escaped_median=float(np.median(d[esc])) if esc.any() else None, escaped_worst=float(d[esc].max()) if esc.any() else None,
               escaped_beyond_1e_8=int((d[esc] > 1e-8).sum()) if esc.any() else 0, gpu_s=round(tg, 2), oracle_s=round(to, 2))
    if sbad.any():
        i = np.nonzero(sbad)[0][:5]
        rec["examples"] = [dict(i=int(j), gpu=int(steps[j]), oracle=int(o["n_attempted"][j]), flag=int(flags[j])) for j in i]
    print(json.dumps(rec), flush=True)
    out.append(rec)
    del fr
json.dump(out, open("gpurun_out/" + (sys.argv[1] if len(sys.argv) > 1 else "r05") + "_every_ray_sweep.json", "w"), indent=1)
print("TOTAL rays", sum(r["rays"] for r in out), "flag diffs", sum(r["flag_diff"] for r in out), "attempted diffs", sum(r["attempted_diff"] for r in out))
