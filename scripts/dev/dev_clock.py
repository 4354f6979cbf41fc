"""Does the trace kernel speed up when launched back-to-back (DVFS ramp)?"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from blackhole_geodesic_calculator_amd import _ffi
from blackhole_geodesic_calculator_amd.device_frame import DeviceFrame
ctx = _ffi.Context(0)
fr = DeviceFrame(ctx, 1024, 1024, 5, fov_x=0.6, fov_y=0.6)
fr.generate_rays()
p = _ffi.make_params(r_s=1.0, lambda_end=50.0)
torch.cuda.synchronize()
for reps in (1, 5, 50, 300):
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    evs[0].record()
    for i in range(reps):
        fr.trace(p)
        evs[i + 1].record()
    torch.cuda.synchronize()
    ts = [evs[i].elapsed_time(evs[i + 1]) for i in range(reps)]
    print("back-to-back", reps, "first %.3f" % ts[0], "last %.3f" % ts[-1], "min %.3f" % min(ts), "mean %.3f" % np.mean(ts))
    import time; time.sleep(0.5)
