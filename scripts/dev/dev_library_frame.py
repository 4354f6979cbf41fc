import sys, time, os
sys.path.insert(0, ".")
import numpy as np
from blackhole_geodesic_calculator_amd import _ffi
from blackhole_geodesic_calculator_amd.raygen import python_random_stream
from blackhole_geodesic_calculator_amd.sky import synthetic_sky
W = H = 1024; S = 5
jit = python_random_stream(42.0, 2 * S * W * H)
fo = _ffi.Frame([0], W, H, S, fov_x=0.6, fov_y=0.6, jitter=jit)
fo.set_scene(synthetic_sky(2048, 1024))
p = _ffi.make_params(r_s=1.0, lambda_end=50.0)
img = fo.render(p)
best = 1e9
for _ in range(12):
    t = time.perf_counter(); fo.render(p, out=img); best = min(best, time.perf_counter() - t)
print(os.environ.get("BHGEO_FRAME_PIECES"), os.environ.get("BHGEO_COPY_PIECE_KB"), "%.3f ms" % (best * 1e3))
