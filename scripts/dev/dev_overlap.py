"""Experiment: consecutive frames on two streams / two contexts (tails and small kernels overlap)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from blackhole_geodesic_calculator_amd import _ffi
from blackhole_geodesic_calculator_amd.device_frame import DeviceFrame, synthetic_sky
from blackhole_geodesic_calculator_amd.raygen import python_random_stream
from blackhole_geodesic_calculator_amd import dist as bdist
W = H = 1024; S = 5
jit = python_random_stream(42.0, 2 * S * W * H)
px = bdist.rank_pixels(W, H, 32, 0, 1)
sky = synthetic_sky(2048, 1024)
p = _ffi.make_params(r_s=1.0, lambda_end=50.0)
frames = []
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
for s in streams:
    ctx = _ffi.Context(0)
    with torch.cuda.stream(s):
        fr = DeviceFrame(ctx, W, H, S, fov_x=0.6, fov_y=0.6, pixels=px, jitter=jit)
        fr.set_sky(sky); fr.generate_rays()
    frames.append(fr)
torch.cuda.synchronize()
def run(nframes, nstreams):
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(nframes):
        k = i % nstreams
        with torch.cuda.stream(streams[k]):
            frames[k].trace(p); frames[k].shade()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / nframes * 1e3
for reps in range(3):
    a = run(40, 1); b = run(40, 2)
    print("serial %.3f ms/frame   two streams %.3f ms/frame   gain %.1f%%" % (a, b, (a / b - 1) * 100))
for w in (8, 10, 12):
    os.environ["BHGEO_WAVES_PER_CU"] = str(w)
    a = run(40, 1); b = run(40, 2)
    print("waves/CU", w, "serial %.3f  two streams %.3f  gain %.1f%%" % (a, b, (a / b - 1) * 100))
