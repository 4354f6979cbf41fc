"""Which property of the adaptive frame lowers the clock: kernel length or memory intensity?"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from blackhole_geodesic_calculator_amd import _ffi
from blackhole_geodesic_calculator_amd.device_frame import DeviceFrame
ctx = _ffi.Context(0)
fr = DeviceFrame(ctx, 1024, 1024, 5, fov_x=0.6, fov_y=0.6)
fr.generate_rays()
def run(name, reps=30, **kw):
    p = _ffi.make_params(r_s=1.0, **kw)
    fr.trace(p); torch.cuda.synchronize()
    ctx.set_profiling(True)
    tr = []
    for _ in range(reps):
        fr.trace(p); tr.append(ctx.last_pass_ms()["trace"])
    ctx.set_profiling(False)
    steps = int(fr.d_steps.to(torch.int64).sum().item())
    ms = float(np.median(tr))
    print("%-34s trace %.3f ms  steps/ray %6.1f  %.2f Gsteps/s  bytes/step %.1f" % (name, ms, steps / fr.n, steps / ms / 1e6, 129.0 * fr.n / steps))
run("adaptive lambda 50", lambda_end=50.0)
run("adaptive lambda 50 reduced", lambda_end=50.0, rhs_form=1)
run("fine 0.1 lambda 50", reps=4, lambda_end=50.0, max_step=0.1)
run("fine 0.1 lambda 5", lambda_end=5.0, max_step=0.1)
run("fine 0.1 lambda 1.5", lambda_end=1.5, max_step=0.1)
run("fine 0.5 lambda 50", reps=10, lambda_end=50.0, max_step=0.5)
run("fine 2 lambda 50", reps=20, lambda_end=50.0, max_step=2.0)
run("fine 5 lambda 50", lambda_end=50.0, max_step=5.0)
run("tight rtol 1e-8", reps=10, lambda_end=50.0, rtol=1e-8, atol=1e-10, rhs_form=1)
