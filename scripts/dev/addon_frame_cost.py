"""Where the add-on's device path spends a frame (dev aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from blackhole_geodesic_calculator_amd import _ffi
from blackhole_geodesic_calculator_amd.raygen import euler_xyz_matrix, python_random_stream
from blackhole_geodesic_calculator_amd.sky import synthetic_sky
W = H = 1024; S = 5
sky = synthetic_sky(2048, 1024).astype(np.float32) if hasattr(synthetic_sky(8, 4), "astype") else None
p = _ffi.make_params(r_s=1.0, lambda_end=50.0)
for rep in range(3):
    t0 = time.perf_counter(); jit = python_random_stream(42.0, 2 * S * W * H); t1 = time.perf_counter()
    fr = _ffi.Frame([0], W, H, S, fov_x=0.6, fov_y=0.6, origin=np.array([1e-4, 0, 30.0]), rot=euler_xyz_matrix((0, 0, 0)), jitter=jit); t2 = time.perf_counter()
    fr.set_scene(sky); t3 = time.perf_counter()
    img = fr.render(p); t4 = time.perf_counter()
    img = fr.render(p); t5 = time.perf_counter()
    fr.close(); t6 = time.perf_counter()
    print("jitter %.1f ms | create %.1f | set_scene %.1f | first render %.1f | second render %.2f | close %.1f" % tuple(1e3 * v for v in (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5)))
