"""Two library-owned frames in flight on shard-sized frames: ms per frame, one frame object vs two alternating (dev aid).
Measured: 1024x128x5 0.250 -> 0.191 ms, 1024x256x5 0.402 -> 0.347, 1024x1024x5 1.198 -> 1.201 (priority streams: the same)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from blackhole_geodesic_calculator_amd import _ffi
from blackhole_geodesic_calculator_amd.raygen import python_random_stream
from blackhole_geodesic_calculator_amd.sky import synthetic_sky
sky = synthetic_sky(2048, 1024)
p = _ffi.make_params(r_s=1.0, lambda_end=50.0)
for W, H in ((1024, 128), (1024, 256), (1024, 1024)):
    S = 5
    jit = python_random_stream(42.0, 2 * S * W * H)
    mk = lambda: _ffi.Frame([0], W, H, S, fov_x=0.6, fov_y=0.6 * H / W, jitter=jit)
    fa, fb = mk(), mk()
    for f in (fa, fb):
        f.set_scene(sky); f.render(p, to_host=False); f.synchronize()
    def run(pair, n=400):
        for i in range(40): pair[i % len(pair)].render(p, to_host=False)
        for f in pair: f.synchronize()
        t = time.perf_counter()
        for i in range(n): pair[i % len(pair)].render(p, to_host=False)
        for f in pair: f.synchronize()
        return (time.perf_counter() - t) / n * 1e3
    a = min(run([fa]) for _ in range(3)); b = min(run([fa, fb]) for _ in range(3))
    print(f"{W}x{H}x{S}: one frame object {a:.4f} ms/frame | two objects alternating {b:.4f}")
    for f in (fa, fb): f.close()
