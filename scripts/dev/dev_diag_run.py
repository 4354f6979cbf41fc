"""Run one workload with a -DBHG_DIAG build and dump the per-wave stamps: dev_diag_run.py frame|disk|orbit <out.bin>"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from blackhole_geodesic_calculator_amd import _ffi
from blackhole_geodesic_calculator_amd.device_frame import DeviceFrame, FrameBatch
what = sys.argv[1] if len(sys.argv) > 1 else "frame"
out = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/diag.bin"
ctx = _ffi.Context(0)
if what in ("disk", "diskkerr"):
    cams = [dict(origin=(30 * np.sin(i), 0.0, 30 * np.cos(i)), rotation_euler=(0.0, i, 0.0)) for i in np.radians([85.0, 80.0, 60.0, 30.0, 5.0])]
    fr = FrameBatch(ctx, cams, 1024, 1024, 1, fov_x=0.9, fov_y=0.9)
    p = _ffi.make_params(r_s=1.0, lambda_end=80.0, r_exit=40.0, disk_r_in=4.5, disk_r_out=10.5,
                         **(dict(rhs_form=2, spin=0.45) if what == "diskkerr" else {}))
elif what == "orbit":
    fr = DeviceFrame(ctx, 2048, 2048, 4, fov_x=0.6, fov_y=0.6)
    fr.set_objects([[8.0, 0.0, 0.0, 1.5]])
    p = _ffi.make_params(r_s=1.0, lambda_end=80.0, r_exit=40.0)
elif what == "exit":
    fr = DeviceFrame(ctx, 1024, 1024, 5, fov_x=0.6, fov_y=0.6)
    p = _ffi.make_params(r_s=1.0, lambda_end=80.0, r_exit=40.0)
elif what == "framesorted":
    # the plain frame with its tiles visited longest-first by the shadow-edge model (bench.py --visit cost)
    from blackhole_geodesic_calculator_amd import dist as bd
    def tile_cost(cx, cy):
        return -abs(np.hypot(0.6 * (cx - 512) / 1024, 0.6 * (cy - 512) / 1024) - 2.598 / 30.0)
    fr = DeviceFrame(ctx, 1024, 1024, 5, fov_x=0.6, fov_y=0.6, pixels=bd.rank_pixels(1024, 1024, 32, 0, 1, tile_cost=tile_cost))
    p = _ffi.make_params(r_s=1.0, lambda_end=50.0)
elif what.startswith("shard"):
    # rank 0's shard of a world-N dealing of the fixed frame (shard8, shard4 ...), tiles visited longest-first
    from blackhole_geodesic_calculator_amd import dist as bd
    def tile_cost(cx, cy):
        return -abs(np.hypot(0.6 * (cx - 512) / 1024, 0.6 * (cy - 512) / 1024) - 2.598 / 30.0)
    tile_cost.visit = "cost"
    fr = DeviceFrame(ctx, 1024, 1024, 5, fov_x=0.6, fov_y=0.6, pixels=bd.rank_pixels(1024, 1024, 32, 0, int(what[5:]), tile_cost=tile_cost))
    p = _ffi.make_params(r_s=1.0, lambda_end=50.0)
else:
    fr = DeviceFrame(ctx, 1024, 1024, 5, fov_x=0.6, fov_y=0.6)
    p = _ffi.make_params(r_s=1.0, lambda_end=50.0)
fr.generate_rays()
for _ in range(100): fr.trace(p)
torch.cuda.synchronize()
os.environ["BHGEO_DIAG_DUMP"] = out
fr.trace(p); torch.cuda.synchronize()
fr.trace(p); torch.cuda.synchronize()
