import os, sys, numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
os.chdir(R)
import test_gpu_parity as T
from blackhole_geodesic_calculator_amd import _ffi
from oracle import oracle as oc
oc.build()
ctx = _ffi.Context(0)
for seed in (180, 240, 241):
    rng = np.random.default_rng(1000 + seed)
    r_s = float(rng.choice([0.0, 0.3, 1.0, 2.5]))
    dist_cam = float(rng.uniform(3.0, 60.0)) * max(r_s, 0.5)
    cam = rng.normal(size=3); cam = dist_cam * cam / np.linalg.norm(cam)
    n = int(rng.integers(1, 3000))
    aim = rng.normal(size=(n, 3)) * max(r_s, 0.5) * float(rng.uniform(1.0, 8.0))
    k = aim - cam; k /= np.linalg.norm(k, axis=1)[:, None]
    x0 = cam + rng.normal(size=(n, 3)) * 0.1 * dist_cam if rng.random() < 0.3 else cam
    kw = dict(r_s=r_s, lambda_end=float(rng.uniform(0.5, 4.0)) * dist_cam, rhs_form=int(rng.integers(0, 2)))
    mode = int(rng.integers(0, 4))
    if mode == 0: kw.update(rtol=float(10 ** rng.uniform(-7, -2)), atol=float(10 ** rng.uniform(-10, -4)))
    elif mode == 1: kw.update(max_step=float(rng.uniform(0.05, 2.0)) * max(r_s, 0.5))
    elif mode == 2: kw.update(method=1, h_fixed=float(rng.uniform(0.05, 0.5)) * max(r_s, 0.5))
    if rng.random() < 0.4: kw["r_exit"] = float(rng.uniform(0.5, 1.5)) * dist_cam
    if rng.random() < 0.4:
        a = float(rng.uniform(1.5, 6.0)) * max(r_s, 0.5); kw.update(disk_r_in=a, disk_r_out=a * float(rng.uniform(1.1, 3.0)))
    if rng.random() < 0.2: kw["max_steps"] = int(rng.integers(1, 40))
    end, flags, steps, acc = ctx.trace(k, x0, _ffi.make_params(**kw))
    o = oc.trace(k, x0, **kw)
    d = np.abs(end - o["end"]).max(1)
    fin = np.isfinite(o["end"]).all(1)
    w = np.nanargmax(np.where(fin, d, 0))
    print("seed", seed, "dist_cam %.1f" % dist_cam, {a: (round(b, 6) if isinstance(b, float) else b) for a, b in kw.items()})
    print("   rays", n, "steps mean %.1f max %d" % (steps.mean(), steps.max()), "flags same", np.array_equal(flags, o["flags"]), "steps same", np.array_equal(steps, o["n_attempted"]))
    print("   worst ray", w, "d %.3g" % d[w], "flag", flags[w], "steps", steps[w], "|end| %.1f" % np.abs(end[w]).max(), "end", end[w].round(4))
    print("   d percentiles 50/90/99/max: ", np.percentile(d[fin], [50, 90, 99, 100]))
    print("   d by flag:", {int(f): float(d[fin & (flags == f)].max()) for f in np.unique(flags[fin])})
