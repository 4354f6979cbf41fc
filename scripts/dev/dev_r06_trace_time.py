"""Round 6: the trace call's time on the headline frame with the library BHGEO_LIB names, results not looked at (for
measurement-only variants whose results are incomplete): python scripts/dev/dev_r06_trace_time.py [reps=5] [steps=200]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from blackhole_geodesic_calculator_amd import _ffi
from blackhole_geodesic_calculator_amd.device_frame import DeviceFrame
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
what = sys.argv[3] if len(sys.argv) > 3 else "frame"       # frame | exit (the same frame with an exit sphere at 40) | exitkerr
ctx = _ffi.Context(0)
fr = DeviceFrame(ctx, 1024, 1024, 5, fov_x=0.6, fov_y=0.6)
fr.generate_rays()
p = _ffi.make_params(r_s=1.0, lambda_end=50.0)
if what == "exit":
    p = _ffi.make_params(r_s=1.0, lambda_end=80.0, r_exit=40.0)
elif what == "exitkerr":
    p = _ffi.make_params(r_s=1.0, lambda_end=80.0, r_exit=40.0, rhs_form=2, spin=0.45)
for _ in range(300):
    fr.trace(p)
torch.cuda.synchronize()
out = []
for r in range(reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        fr.trace(p)
    e1.record()
    torch.cuda.synchronize()
    out.append(e0.elapsed_time(e1) / steps)
print(os.path.basename(os.environ.get("BHGEO_LIB", "tree")), what, "trace call ms:", " ".join("%.4f" % v for v in out), "median %.4f" % float(np.median(out)))
