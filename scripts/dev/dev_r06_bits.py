"""Round 6: sha256 of every result array of a set of full-size traces with the library BHGEO_LIB names -- run once per build and
compare the lines (an A/B variant must not change a bit): python scripts/dev/dev_r06_bits.py [frame disk diskkerr orbit exit]"""
import hashlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from blackhole_geodesic_calculator_amd import _ffi
from blackhole_geodesic_calculator_amd.device_frame import DeviceFrame, FrameBatch
ctx = _ffi.Context(0)
for what in (sys.argv[1:] or ["frame", "disk", "diskkerr", "orbit", "exit"]):
    if what in ("disk", "diskkerr"):
        cams = [dict(origin=(30 * np.sin(i), 0.0, 30 * np.cos(i)), rotation_euler=(0.0, i, 0.0)) for i in np.radians([85.0, 80.0, 60.0, 30.0, 5.0])]
        fr = FrameBatch(ctx, cams, 1024, 1024, 1, fov_x=0.9, fov_y=0.9)
        p = _ffi.make_params(r_s=1.0, lambda_end=80.0, r_exit=40.0, disk_r_in=4.5, disk_r_out=10.5, **(dict(rhs_form=2, spin=0.45) if what == "diskkerr" else {}))
    elif what == "orbit":
        fr = DeviceFrame(ctx, 2048, 2048, 4, fov_x=0.6, fov_y=0.6)
        fr.set_objects([[8.0, 0.0, 0.0, 1.5]])
        p = _ffi.make_params(r_s=1.0, lambda_end=80.0, r_exit=40.0)
    elif what == "exitkerr":
        fr = DeviceFrame(ctx, 1024, 1024, 5, fov_x=0.6, fov_y=0.6)
        p = _ffi.make_params(r_s=1.0, lambda_end=80.0, r_exit=40.0, rhs_form=2, spin=0.45)
    elif what == "orbitkerr":
        fr = DeviceFrame(ctx, 1024, 1024, 5, fov_x=0.6, fov_y=0.6)
        fr.set_objects([[8.0, 0.0, 0.0, 1.5]])
        p = _ffi.make_params(r_s=1.0, lambda_end=80.0, r_exit=40.0, rhs_form=2, spin=0.45)
    elif what == "exit":
        fr = DeviceFrame(ctx, 1024, 1024, 5, fov_x=0.6, fov_y=0.6)
        p = _ffi.make_params(r_s=1.0, lambda_end=80.0, r_exit=40.0)
    elif what in ("kerr", "kerroff"):
        # config 5 (on-axis camera) / the off-axis Kerr frame of the suite
        inc = np.radians(60.0)
        kwf = dict(origin=(30 * np.sin(inc), 0.0, 30 * np.cos(inc)), rotation_euler=(0.0, inc, 0.0)) if what == "kerroff" else {}
        fr = DeviceFrame(ctx, 1024, 1024, 5, fov_x=0.6, fov_y=0.6, **kwf)
        p = _ffi.make_params(r_s=1.0, lambda_end=50.0, rhs_form=2, spin=0.45)
    else:
        fr = DeviceFrame(ctx, 1024, 1024, 5, fov_x=0.6, fov_y=0.6)
        p = _ffi.make_params(r_s=1.0, lambda_end=50.0)
    fr.generate_rays()
    fr.trace(p)
    torch.cuda.synchronize()
    frames = [fr]        # (a FrameBatch owns the arrays its frames are slices of)
    h = hashlib.sha256()
    for f in frames:
        for name in ("d_end", "d_flags", "d_steps", "d_acc", "d_obj"):
            t = getattr(f, name, None)
            if t is not None:
                h.update(t.cpu().numpy().tobytes())
    print(what, h.hexdigest()[:24], "rays", sum(f.n for f in frames), flush=True)
    del fr
