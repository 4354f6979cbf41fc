"""Developer stress check on the GPU box: mid-size event-heavy frames, census of flags, A/B against another library build
(BHGEO_LIB_REF) for identical flags / step counts."""
import os, sys, time, subprocess, json
import numpy as np
sys.path.insert(0, os.getcwd())
which = sys.argv[1] if len(sys.argv) > 1 else "disk"
n_side = int(sys.argv[2]) if len(sys.argv) > 2 else 512
out = sys.argv[3] if len(sys.argv) > 3 else None
import torch
from blackhole_geodesic_calculator_amd import _ffi
from blackhole_geodesic_calculator_amd.device_frame import DeviceFrame, FrameBatch
ctx = _ffi.Context(0)
if which.startswith("disk"):
    cams = [dict(origin=(30 * np.sin(i), 0.0, 30 * np.cos(i)), rotation_euler=(0.0, i, 0.0)) for i in np.radians([85.0, 80.0, 60.0, 30.0, 5.0])]
    fr = FrameBatch(ctx, cams, n_side, n_side, 1, fov_x=0.9, fov_y=0.9)
    kw = dict(r_s=1.0, lambda_end=80.0, r_exit=40.0, disk_r_in=4.5, disk_r_out=10.5)
    if which == "diskkerr":
        kw.update(rhs_form=2, spin=0.45)
else:
    fr = DeviceFrame(ctx, n_side, n_side, 4, fov_x=0.6, fov_y=0.6)
    fr.set_objects([[8.0, 0.0, 0.0, 1.5]])
    kw = dict(r_s=1.0, lambda_end=80.0, r_exit=40.0)
p = _ffi.make_params(**kw)
fr.generate_rays()
fr.d_flags.fill_(0); fr.d_steps.fill_(0)
torch.cuda.synchronize()
t = time.time(); fr.trace(p); torch.cuda.synchronize(); dt = time.time() - t
fl = fr.d_flags.cpu().numpy(); st = fr.d_steps.cpu().numpy()
u, c = np.unique(fl, return_counts=True)
print(which, n_side, "n", fr.n, "first call %.1f ms" % (dt * 1e3), "census", dict(zip(u.tolist(), c.tolist())), "steps sum", int(st.astype(np.int64).sum()))
t = time.time(); fr.trace(p); torch.cuda.synchronize(); print("second call %.2f ms" % ((time.time() - t) * 1e3))
if out:
    np.savez(out, flags=fl, steps=st, end=fr.d_end.cpu().numpy())
