"""Developer check on the GPU box: GPU vs oracle on the event-heavy golden sets, mismatches by class."""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from blackhole_geodesic_calculator_amd import _ffi
from oracle import oracle as oc
from conftest import load_golden

ctx = _ffi.Context(0)
def run(name, k0, x0, spheres=None, **kw):
    o = oc.trace(k0, x0, **(dict(kw, spheres=spheres) if spheres is not None else kw))
    p = _ffi.make_params(**{a: b for a, b in kw.items() if a != "spheres"})
    if spheres is not None:
        end, flags, steps, acc, obj = ctx.trace(k0, x0, p, spheres=spheres)
    else:
        end, flags, steps, acc = ctx.trace(k0, x0, p)
    d = np.abs(end - o["end"]).max(1)
    fbad = flags != o["flags"]; sbad = steps != o["n_attempted"]; abad = acc != o["n_accepted"]
    print(f"{name}: n {len(d)} flags-bad {fbad.sum()} steps-bad {sbad.sum()} acc-bad {abad.sum()}")
    for f in np.unique(o["flags"]):
        m = o["flags"] == f
        dm = d[m & ~fbad]
        print(f"   oracle flag {f:3d}: {m.sum():6d} rays, gpu flag differs {int((m & fbad).sum()):5d}, end diff max {np.nanmax(dm) if len(dm) else 0:.2e} median {np.nanmedian(dm) if len(dm) else 0:.2e}, >1e-9: {int((dm > 1e-9).sum())}")
    bad = np.nonzero(fbad | sbad)[0][:6]
    for i in bad:
        print("     ray", i, "gpu flag", flags[i], "oracle", o["flags"][i], "steps", steps[i], o["n_attempted"][i], "acc", acc[i], o["n_accepted"][i], "end gpu", np.round(end[i], 6), "oracle", np.round(o["end"][i], 6))
    big = np.argsort(-np.nan_to_num(np.where(fbad, 0, d)))[:3]
    for i in big:
        print("     worst", i, "flag", flags[i], "steps", steps[i], "diff %.3e" % d[i], "end gpu", end[i], "oracle", o["end"][i])

g = load_golden("sphere_exit"); run("sphere_exit", g["k0"], g["x0"], r_s=float(g["r_s"]), lambda_end=float(g["lambda_end"]), r_exit=float(g["r_exit"]))
g = load_golden("disk"); run("disk", g["k0"], g["x0"], r_s=1.0, lambda_end=80.0, disk_r_in=4.5, disk_r_out=10.5)
g = load_golden("disk"); run("disk+exit", g["k0"], g["x0"], r_s=1.0, lambda_end=80.0, r_exit=40.0, disk_r_in=4.5, disk_r_out=10.5)
g = load_golden("objects"); run("objects", g["k0"], g["x0"], spheres=g["spheres"], r_s=1.0, lambda_end=70.0, max_step=0.25, r_exit=35.0, disk_r_in=3.0, disk_r_out=7.0)
g = load_golden("kerr_disk"); run("kerr_disk", g["k0"], g["x0"], r_s=1.0, lambda_end=80.0, rhs_form=2, spin=float(g["spin"]), disk_r_in=float(g["disk_r_in"]), disk_r_out=float(g["disk_r_out"]))
g = load_golden("frame64_christoffel"); run("frame64", g["k0"], g["x0"], r_s=1.0, lambda_end=50.0)
from conftest import frame_rays, CAM
k = frame_rays(20000, seed=3, fov=0.9)
run("seeded disk+exit rk4", k[:4000], CAM, r_s=1.0, lambda_end=80.0, r_exit=40.0, disk_r_in=4.5, disk_r_out=10.5, method=1, h_fixed=0.1)
run("seeded disk+exit", k, CAM, r_s=1.0, lambda_end=80.0, r_exit=40.0, disk_r_in=4.5, disk_r_out=10.5)
