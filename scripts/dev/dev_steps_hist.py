"""Distribution of attempted steps per ray for a workload, and where in the launch order the long rays sit:
python3 scripts/dev/dev_steps_hist.py frame|disk|diskkerr|kerr"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from blackhole_geodesic_calculator_amd import _ffi
from blackhole_geodesic_calculator_amd.device_frame import DeviceFrame, FrameBatch
what = sys.argv[1] if len(sys.argv) > 1 else "frame"
ctx = _ffi.Context(0)
if what in ("disk", "diskkerr"):
    cams = [dict(origin=(30 * np.sin(i), 0.0, 30 * np.cos(i)), rotation_euler=(0.0, i, 0.0)) for i in np.radians([85.0, 80.0, 60.0, 30.0, 5.0])]
    fr = FrameBatch(ctx, cams, 1024, 1024, 1, fov_x=0.9, fov_y=0.9)
    p = _ffi.make_params(r_s=1.0, lambda_end=80.0, r_exit=40.0, disk_r_in=4.5, disk_r_out=10.5, **(dict(rhs_form=2, spin=0.45) if what == "diskkerr" else {}))
    frames = fr.frames
else:
    fr = DeviceFrame(ctx, 1024, 1024, 5, fov_x=0.6, fov_y=0.6)
    p = _ffi.make_params(r_s=1.0, lambda_end=50.0, **(dict(rhs_form=2, spin=0.45) if what == "kerr" else {}))
    frames = [fr]
fr.generate_rays(); fr.trace(p); torch.cuda.synchronize()
st = torch.cat([f.d_steps for f in frames]).cpu().numpy().astype(np.int64)
print(what, "rays", len(st), "mean %.2f" % st.mean(), "max", st.max(), "percentiles 50/90/99/99.9/99.99/99.999:", np.percentile(st, [50, 90, 99, 99.9, 99.99, 99.999]).round(1))
for thr in (50, 100, 200, 400):
    idx = np.nonzero(st > thr)[0]
    if len(idx):
        pos = (idx % (len(st) // len(frames))) / (len(st) // len(frames))     # position within its block = launch-order fraction (chunk-major hand-out)
        print("  > %d steps: %d rays, launch-order position min %.3f median %.3f max %.3f" % (thr, len(idx), pos.min(), np.median(pos), pos.max()))
