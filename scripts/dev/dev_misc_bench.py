"""Numbers for DESIGN.md: PCIe-inclusive host call, config-3 disk frames (five inclinations), RK4 / fine."""
import os, sys, time, math
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from blackhole_geodesic_calculator_amd import _ffi, camera_directions
from blackhole_geodesic_calculator_amd.raygen import euler_xyz_matrix
ctx = _ffi.Context(0)
cam = np.array([1e-4, 0.0, 30.0])
k0 = camera_directions(1024, 1024, 5, 0.6, 0.6, 42.0).reshape(-1, 3)
n = len(k0)
p = _ffi.make_params(r_s=1.0, lambda_end=50.0)
ctx.trace(k0[:1000], cam, p)
for _ in range(2):
    t = time.perf_counter(); ctx.trace(k0, cam, p); dt = time.perf_counter() - t
print("host-buffer bhg_trace (H2D + 3 passes + D2H), config 2: %.1f ms -> %.0f Mrays/s" % (dt * 1e3, n / dt / 1e6))
# config 3: 1024^2, S=1, camera on a circle r=30 at 5 inclinations, looking at the hole; disk 4.5..10.5 r_s
dk = torch.empty((1024 * 1024, 3), dtype=torch.float64, device="cuda")
dend = torch.empty((1024 * 1024, 6), dtype=torch.float64, device="cuda")
dfl = torch.empty(1024 * 1024, dtype=torch.uint8, device="cuda")
dst = torch.empty(1024 * 1024, dtype=torch.int32, device="cuda")
pd = _ffi.make_params(r_s=1.0, lambda_end=80.0, disk_r_in=4.5, disk_r_out=10.5)
for inc_deg in (85.0, 80.0, 60.0, 30.0, 5.0):
    inc = math.radians(inc_deg)
    c = np.array([30 * math.sin(inc), 0.0, 30 * math.cos(inc)])
    # camera looks down its -z axis: rotate about y by inc so that -z points at the origin
    d = camera_directions(1024, 1024, 1, 0.6, 0.6, 42.0, rotation_euler=(0.0, inc, 0.0)).reshape(-1, 3)
    dk.copy_(torch.from_numpy(d))
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        ctx.trace_device(pd, len(d), dk.data_ptr(), dend.data_ptr(), x0_shared=c, d_flags=dfl.data_ptr(), d_n_steps=dst.data_ptr(), stream=0)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    f = dfl.cpu().numpy()
    print("config 3 inclination %4.0f deg: %.2f ms/frame (1 Mray), passes %d, disk hits %.1f%%, horizon %.1f%%, steps/ray %.1f" %
          (inc_deg, min(ts) * 1e3, ctx.last_launch()["passes"], 100 * (f == 128).mean(), 100 * ((f & 1) != 0).mean(), dst.float().mean().item()))
