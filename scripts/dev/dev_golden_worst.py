"""Development aid: measured worst-case |gpu - oracle| and |gpu - scipy golden| per golden set and ray class."""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from conftest import GOLDEN_TRACE_SETS, golden_kwargs, load_golden
from blackhole_geodesic_calculator_amd import _ffi
from oracle import oracle as oc
oc.build()
ctx = _ffi.Context(0)
CLASSES = {"escaped": lambda f: (f == 4) | (f == 8), "horizon": lambda f: (f & 1) != 0, "disk": lambda f: f == 128, "object": lambda f: f == 0x88}
def run(name, g, kw, sl=slice(None)):
    k0, x0 = g["k0"][sl], g["x0"][sl] if g["x0"].ndim == 2 else g["x0"]
    sp = kw.get("spheres")
    p = _ffi.make_params(**{a: b for a, b in kw.items() if a != "spheres"})
    r = ctx.trace(k0, x0, p, spheres=sp) if sp is not None else ctx.trace(k0, x0, p)
    end, fl = r[0], r[1]
    o = oc.trace(k0, x0, **kw)
    d_o = np.abs(end - o["end"]).max(1); d_g = np.abs(end - g["end"][sl]).max(1)
    for cn, sel in CLASSES.items():
        m = sel(fl)
        if m.any():
            print(f'    ("{name}", "{cn}"): ({d_o[m].max():.1e}, {d_g[m].max():.1e}),   # n = {int(m.sum())}, rays over 1e-9 vs oracle: {int((d_o[m] > 1e-9).sum())}')
for rhs in (0, 1):
    for name in GOLDEN_TRACE_SETS:
        g = load_golden(name)
        run(f"{name}/{rhs}", g, golden_kwargs(g, rhs))
g = load_golden("disk")
for rhs in (0, 1):
    run(f"disk/{rhs}", g, dict(r_s=1.0, lambda_end=80.0, disk_r_in=4.5, disk_r_out=10.5, rhs_form=rhs))
g = load_golden("objects")
run("objects/0", g, dict(r_s=1.0, lambda_end=70.0, max_step=0.25, r_exit=35.0, disk_r_in=3.0, disk_r_out=7.0, spheres=g["spheres"]))
g = load_golden("kerr_a09")
run("kerr_a09/2", g, dict(r_s=1.0, lambda_end=60.0, rhs_form=2, spin=float(g["spin"])), slice(48, None))
g = load_golden("kerr_disk")
run("kerr_disk/2", g, dict(r_s=1.0, lambda_end=80.0, rhs_form=2, spin=float(g["spin"]), disk_r_in=3.0, disk_r_out=10.0))
