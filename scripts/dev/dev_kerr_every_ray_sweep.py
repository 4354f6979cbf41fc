"""Round 5: flags / attempted / accepted step counts of the device against the checker on EVERY ray of a sweep of full-size
Kerr frames (1024 x 1024 x 5 each, Boyer-Lindquist) -- spins from -0.95 to 0.998 M, cameras from 5 to 85 degrees off the
axis, near and far, with the thin disk, the exit sphere, other tolerances -> gpurun_out/<tag>_kerr_every_ray_sweep.json (tag = argv[1], default r05)"""
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


def cam_at(r, inc_deg, y=0.3):
    i = np.radians(inc_deg)
    return (r * np.sin(i), y, r * np.cos(i)), (0.0, i, 0.0)


cases = []
for name, r, inc, fov, kw in [
    ("a/M 0.9, 60 deg (the test's frame)", 30.0, 60.0, 0.6, dict(spin=0.45, lambda_end=50.0)),
    ("a/M 0.9, 85 deg (near the equator)", 30.0, 85.0, 0.6, dict(spin=0.45, lambda_end=50.0)),
    ("a/M 0.9, 5 deg (near the axis)", 30.0, 5.0, 0.6, dict(spin=0.45, lambda_end=50.0)),
    ("a/M 0.998, 75 deg", 30.0, 75.0, 0.6, dict(spin=0.499, lambda_end=50.0)),
    ("a/M -0.95 (retrograde), 70 deg", 30.0, 70.0, 0.6, dict(spin=-0.475, lambda_end=50.0)),
    ("a/M 0.5, 40 deg, near camera r = 10, wide", 10.0, 40.0, 1.4, dict(spin=0.25, lambda_end=40.0)),
    ("a/M 0.9, 80 deg, disk 3..10 + exit sphere 40", 30.0, 80.0, 0.9, dict(spin=0.45, lambda_end=80.0, r_exit=40.0, disk_r_in=3.0, disk_r_out=10.0)),
    ("a/M 0.7, 60 deg, rtol 1e-5", 30.0, 60.0, 0.6, dict(spin=0.35, lambda_end=50.0, rtol=1e-5, atol=1e-8)),
]:
    c, e = cam_at(r, inc)
    cases.append(dict(name=name, cam=c, euler=e, fov=fov, kw=dict(dict(r_s=1.0, rhs_form=2), **kw)))
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
    sbad = (steps != o["n_attempted"]) | (acc != o["n_accepted"])
    hor = ((flags | o["flags"]) & (1 | 64)) != 0
    d = np.abs(end - o["end"]).max(1)
    esc = ((flags == 4) | (flags == 8) | (flags == 128)) & ~fbad & ~sbad      # (not the STEP_TOO_SMALL rays of the near-extremal frame: they stall at the horizon)
    ds = np.abs(steps.astype(int) - o["n_attempted"].astype(int))
    rec = dict(name=c["name"], rays=int(len(k0)), flag_diff=int(fbad.sum()), step_diff=int(sbad.sum()),
               step_diff_horizon_rays=int((sbad & hor).sum()), step_diff_other_rays=int((sbad & ~hor).sum()), horizon_rays=int(hor.sum()),
               largest_step_diff=int(ds.max()), steps_per_ray=float(steps.mean()),
               census={int(f): int(n) for f, n in zip(*np.unique(flags, return_counts=True))},
               escaped_median=float(np.median(d[esc])), escaped_p999=float(np.percentile(d[esc], 99.9)), escaped_worst=float(d[esc].max()),
               escaped_beyond_5e_8=int((d[esc] > 5e-8).sum()), gpu_s=round(tg, 2), oracle_s=round(to, 2))
    print(json.dumps(rec), flush=True)
    out.append(rec)
    del fr
json.dump(out, open("gpurun_out/" + (sys.argv[1] if len(sys.argv) > 1 else "r05") + "_kerr_every_ray_sweep.json", "w"), indent=1)
print("TOTAL rays", sum(r["rays"] for r in out), "flag diffs", sum(r["flag_diff"] for r in out), "step diffs", sum(r["step_diff"] for r in out),
      "of them on rays that do not end on the horizon", sum(r["step_diff_other_rays"] for r in out))
