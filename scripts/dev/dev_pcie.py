"""PCIe / host-copy speeds of the box next to the host-buffer call's figure (development aid)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
n = 5242880
d = torch.empty(n * 6, dtype=torch.float64, device="cuda")
hp = torch.empty(n * 6, dtype=torch.float64).pin_memory()
for name, fn, nbytes in (("D2H pinned 252 MB", lambda: hp.copy_(d, non_blocking=True), n * 48),
                         ("H2D pinned 252 MB", lambda: d.copy_(hp, non_blocking=True), n * 48)):
    best = 1e9
    for _ in range(5):
        torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
    print(name, "%.2f ms  %.1f GB/s" % (best * 1e3, nbytes / best / 1e9))
a = np.random.default_rng(0).random(n * 3)
b = np.empty_like(a)
best = 1e9
for _ in range(5):
    t = time.perf_counter(); np.copyto(b, a); best = min(best, time.perf_counter() - t)
print("host memcpy 126 MB single thread %.2f ms %.1f GB/s" % (best * 1e3, a.nbytes / best / 1e9))
from blackhole_geodesic_calculator_amd import _ffi
ctx = _ffi.Context(0)
k = np.random.default_rng(1).normal(size=(n, 3)); k /= np.linalg.norm(k, axis=1)[:, None]; k[:, 2] = -abs(k[:, 2])
cam = np.array([1e-4, 0.0, 30.0]); p = _ffi.make_params(r_s=1.0, lambda_end=50.0)
for pinned in (True, False):
    for want in (True, False):
        ctx.trace(k, cam, p, pinned_results=pinned, want_steps=want)
        best = 1e9
        for _ in range(3):
            t = time.perf_counter(); r = ctx.trace(k, cam, p, pinned_results=pinned, want_steps=want); best = min(best, time.perf_counter() - t)
        print("bhg_trace pinned_results=%s want_steps=%s: %.2f ms  %.0f Mrays/s" % (pinned, want, best * 1e3, n / best / 1e6))
print("cpus:", len(os.sched_getaffinity(0)), open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "")
