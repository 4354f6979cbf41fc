"""Where the per-ray call's time goes (dev aid): the bare C call with preallocated arrays, Context.trajectory, calc_trajectory."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from blackhole_geodesic_calculator_amd import GeodesicIntegratorSchwarzschild, _ffi
gi = GeodesicIntegratorSchwarzschild(mass=0.5)
ctx = gi.context
lib = _ffi.load()
x0 = np.array([1e-4, 0.0, 30.0])
rng = np.random.default_rng(0)
K = np.array([0, 0, -1.0]) + rng.normal(size=(2000, 3)) * 0.15
p = gi.params(1e4, 50.0)
dp = C.POINTER(C.c_double)
for T in (50, 10000):
    traj = ctx.pinned.empty((1, 6, T), np.float64)
    nv = np.empty(1, np.uint32); end = np.empty((1, 6)); fl = np.empty(1, np.uint8)
    args = (ctx._h, C.byref(p), x0.ctypes.data_as(dp), 1, None, 1, T, traj.ctypes.data_as(dp), nv.ctypes.data_as(C.POINTER(C.c_uint32)),
            end.ctypes.data_as(dp), fl.ctypes.data_as(C.POINTER(C.c_uint8)))
    ks = [K[i].ctypes.data_as(dp) for i in range(500)]
    for rep in range(2):
        t = time.perf_counter()
        for i in range(500):
            lib.bhg_trajectory(args[0], args[1], args[2], 1, ks[i], 1, T, args[7], args[8], args[9], args[10])
        a = (time.perf_counter() - t) / 500
    for rep in range(2):
        t = time.perf_counter()
        for i in range(500):
            ctx.trajectory(K[i][None, :], x0, p, T)
        b = (time.perf_counter() - t) / 500
    for rep in range(2):
        t = time.perf_counter()
        for i in range(500):
            gi.calc_trajectory(K[i], x0, max_step=1e4, curve_end=50, nr_points_curve=T)
        c = (time.perf_counter() - t) / 500
    print(f"T={T}: bare C call {a*1e6:.1f} us, Context.trajectory {b*1e6:.1f} us, calc_trajectory {c*1e6:.1f} us")
ctx.set_profiling(False)
