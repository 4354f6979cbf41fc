"""ctypes loader for oracle/libgeodesic_oracle.so -- TEST INFRASTRUCTURE ONLY.

The C file's header states what it follows (reference call sites, README equations, scipy's
RK45) and that parity is unpinned.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# BHG_ORACLE_LIB: another build of the same source, e.g. the sanitizer build (`make -C oracle asan`)
_LIB_PATH = os.environ.get("BHG_ORACLE_LIB") or os.path.join(_HERE, "libgeodesic_oracle.so")

FLAG_HIT_HORIZON = 1
FLAG_START_INSIDE = 2
FLAG_REACHED_END = 4
FLAG_EXITED_SPHERE = 8
FLAG_MAX_STEPS = 16
FLAG_STEP_TOO_SMALL = 32
FLAG_NAN = 64
FLAG_HIT_DISK = 128
FLAG_HIT_OBJECT = 0x88
MAX_SPHERES = 8

METHOD_DP54 = 0
METHOD_RK4 = 1
RHS_CHRISTOFFEL = 0
RHS_REDUCED = 1
RHS_KERR_BL = 2
KERR_HORIZON_MARGIN = 1e-3


class Params(C.Structure):
    _fields_ = [
        ("r_s", C.c_double),
        ("lambda_end", C.c_double),
        ("max_step", C.c_double),
        ("rtol", C.c_double),
        ("atol", C.c_double),
        ("h_fixed", C.c_double),
        ("r_exit", C.c_double),
        ("method", C.c_int32),
        ("rhs_form", C.c_int32),
        ("max_steps", C.c_uint32),
        ("time_like", C.c_uint32),
        ("disk_r_in", C.c_double),
        ("disk_r_out", C.c_double),
        ("spin", C.c_double),
        ("n_spheres", C.c_int32),
        ("reserved2", C.c_int32),
        ("spheres", (C.c_double * 4) * 8),
    ]


def make_params(r_s=1.0, lambda_end=50.0, max_step=np.inf, rtol=1e-3, atol=1e-6, h_fixed=0.1,
                r_exit=0.0, method=METHOD_DP54, rhs_form=RHS_CHRISTOFFEL, max_steps=0, disk_r_in=0.0,
                disk_r_out=0.0, spin=0.0, spheres=None, time_like=0):
    p = Params(r_s, lambda_end, max_step, rtol, atol, h_fixed, r_exit, method, rhs_form,
               max_steps, int(time_like), disk_r_in, disk_r_out, spin)
    if spheres is not None:
        sp = np.asarray(spheres, dtype=np.float64).reshape(-1, 4)
        assert len(sp) <= MAX_SPHERES
        p.n_spheres = len(sp)
        for j, row in enumerate(sp):
            for c in range(4):
                p.spheres[j][c] = float(row[c])
    return p


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("geodesic_oracle.c", "kerr_rhs.inc")]
    if os.environ.get("BHG_ORACLE_LIB"):
        return _LIB_PATH          # (a build of the caller's own making)
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        dp = C.POINTER(C.c_double)
        L.bhgo_trace.restype = C.c_int
        L.bhgo_trace.argtypes = [C.POINTER(Params), dp, C.c_int, dp, C.c_size_t, dp,
                                 C.POINTER(C.c_uint8), C.POINTER(C.c_uint32),
                                 C.POINTER(C.c_uint32), dp, C.c_int]
        L.bhgo_acceleration.restype = C.c_int
        L.bhgo_acceleration.argtypes = [C.POINTER(Params), dp, dp, C.c_size_t, dp]
        L.bhgo_trace_objects.restype = C.c_int
        L.bhgo_trace_objects.argtypes = [C.POINTER(Params), dp, C.c_int, dp, C.c_size_t, dp, C.POINTER(C.c_uint8),
                                         C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_int8), C.c_int]
        L.bhgo_trajectory.restype = C.c_int
        L.bhgo_trajectory.argtypes = [C.POINTER(Params), dp, C.c_int, dp, C.c_size_t, C.c_uint32, dp,
                                      C.POINTER(C.c_uint32), C.POINTER(C.c_uint8)]
        L.bhgo_num_threads.restype = C.c_int
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def trace(k0, x0, n_threads=0, **kw):
    """k0[N,3], x0[3] or [N,3] -> dict(end[N,6], flags[N], n_attempted[N], n_accepted[N], t_end[N])."""
    p = kw.pop("params", None) or make_params(**kw)
    k0 = np.ascontiguousarray(np.atleast_2d(k0), dtype=np.float64)
    x0 = np.ascontiguousarray(x0, dtype=np.float64)
    n = k0.shape[0]
    shared = 1 if x0.ndim == 1 else 0
    if not shared:
        assert x0.shape == (n, 3)
    end = np.empty((n, 6))
    flags = np.empty(n, np.uint8)
    natt = np.empty(n, np.uint32)
    nacc = np.empty(n, np.uint32)
    tend = np.empty(n)
    rc = lib().bhgo_trace(C.byref(p), _dp(x0), shared, _dp(k0), n, _dp(end),
                          flags.ctypes.data_as(C.POINTER(C.c_uint8)),
                          natt.ctypes.data_as(C.POINTER(C.c_uint32)),
                          nacc.ctypes.data_as(C.POINTER(C.c_uint32)), _dp(tend), n_threads)
    if rc != 0:
        raise RuntimeError(f"bhgo_trace failed: {rc}")
    out = {"end": end, "flags": flags, "n_attempted": natt, "n_accepted": nacc, "t_end": tend}
    if p.n_spheres > 0:
        obj = np.empty(n, np.int8)
        rc = lib().bhgo_trace_objects(C.byref(p), _dp(x0), shared, _dp(k0), n, _dp(end),
                                      flags.ctypes.data_as(C.POINTER(C.c_uint8)),
                                      natt.ctypes.data_as(C.POINTER(C.c_uint32)),
                                      nacc.ctypes.data_as(C.POINTER(C.c_uint32)),
                                      obj.ctypes.data_as(C.POINTER(C.c_int8)), n_threads)
        if rc != 0:
            raise RuntimeError(f"bhgo_trace_objects failed: {rc}")
        out["object_id"] = obj
    return out


def trajectory(k0, x0, n_points, **kw):
    """Sampled curves: (traj[N,6,T], n_valid[N], flags[N]); t_eval = linspace(0, lambda_end, T)."""
    p = make_params(**kw)
    k0 = np.ascontiguousarray(np.atleast_2d(k0), dtype=np.float64)
    x0 = np.ascontiguousarray(x0, dtype=np.float64)
    n = k0.shape[0]
    traj = np.full((n, 6, n_points), np.nan)
    nv = np.zeros(n, np.uint32)
    flags = np.zeros(n, np.uint8)
    rc = lib().bhgo_trajectory(C.byref(p), _dp(x0), 1 if x0.ndim == 1 else 0, _dp(k0), n, n_points, _dp(traj),
                               nv.ctypes.data_as(C.POINTER(C.c_uint32)), flags.ctypes.data_as(C.POINTER(C.c_uint8)))
    if rc != 0:
        raise RuntimeError(f"bhgo_trajectory failed: {rc}")
    return traj, nv, flags


def acceleration(x, k, **kw):
    p = make_params(**kw)
    x = np.ascontiguousarray(np.atleast_2d(x), dtype=np.float64)
    k = np.ascontiguousarray(np.atleast_2d(k), dtype=np.float64)
    a = np.empty_like(x)
    lib().bhgo_acceleration(C.byref(p), _dp(x), _dp(k), x.shape[0], _dp(a))
    return a


def num_threads():
    return lib().bhgo_num_threads()
