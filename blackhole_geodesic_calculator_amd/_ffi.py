"""ctypes binding of libbhgeo.so (C ABI declared in include/bhgeo.h).

The library is the product: there is no Python or CPU fallback.  Importing this module never
touches the GPU; `load()` raises loudly when the shared library has not been built, and
`Context()` raises when no HIP device is usable.
"""
from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BHGEO_LIB") or os.path.join(_HERE, "libbhgeo.so")  # BHGEO_LIB: A/B builds

ABI_VERSION = 8
ABI_COMPAT_MIN = 7    # bhg_abi_check serves bindings from this ABI on (8 only added bhg_trajectory_objects)

OK = 0
E_INVALID, E_NO_DEVICE, E_HIP, E_NOMEM = -1, -2, -3, -4

FLAG_HIT_HORIZON = 1
FLAG_START_INSIDE = 2
FLAG_REACHED_END = 4
FLAG_EXITED_SPHERE = 8
FLAG_MAX_STEPS = 16
FLAG_STEP_TOO_SMALL = 32
FLAG_NAN = 64
FLAG_HIT_DISK = 128
FLAG_HIT_OBJECT = 0x88  # composite: test with (flags & 0x88) == 0x88
MAX_SPHERES = 8

METHOD_DP54 = 0
METHOD_RK4 = 1
RHS_CHRISTOFFEL = 0
RHS_REDUCED = 1
RHS_KERR_BL = 2

# every symbol include/bhgeo.h declares (tests check the built library exports all of them)
EXPORTS = (
    "bhg_version", "bhg_device_count", "bhg_last_error", "bhg_default_params", "bhg_create",
    "bhg_destroy", "bhg_device_name", "bhg_num_cus", "bhg_trace", "bhg_trace_device",
    "bhg_acceleration", "bhg_synchronize", "bhg_last_launch", "bhg_context_stream", "bhg_raygen_device",
    "bhg_shade_device", "bhg_set_profiling", "bhg_last_pass_ms", "bhg_trajectory", "bhg_trace_objects",
    "bhg_trace_objects_device", "bhg_shade_scene_device", "bhg_shade_scene_f32_device",
    "bhg_assemble_frame_f32_device", "bhg_host_alloc", "bhg_host_free", "bhg_rays_create", "bhg_rays_count",
    "bhg_rays_destroy", "bhg_rays_trace", "bhg_trace_dir_device", "bhg_shade_dir_device",
    "bhg_frame_create", "bhg_frame_destroy", "bhg_frame_set_scene", "bhg_frame_set_camera", "bhg_frame_render", "bhg_frame_synchronize",
    "bhg_frame_device_image", "bhg_frame_rebalance", "bhg_frame_stats", "bhg_frame_info", "bhg_frame_set_profiling",
    "bhg_frame_last_ms", "bhg_deal_tiles",
    "bhg_params_size", "bhg_camera_size", "bhg_scene_size", "bhg_frame_scene_size", "bhg_abi_check",
    "bhg_default_params_sized", "bhg_peak_probe", "bhg_trajectory_objects",
)
PROBE_FMA, PROBE_STEP_MIX = 0, 1

GATHER_AUTO, GATHER_COPY, GATHER_RCCL, GATHER_PEER, GATHER_COPY_PEERCALL = 0, 1, 2, 3, 4


class Camera(C.Structure):
    """struct bhg_camera (include/bhgeo.h)."""
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32), ("samples", C.c_int32), ("reserved", C.c_int32),
        ("fov_x", C.c_double), ("fov_y", C.c_double),
        ("rot", C.c_double * 9),
        ("origin", C.c_double * 3),
    ]


class Scene(C.Structure):
    """struct bhg_scene (include/bhgeo.h)."""
    _fields_ = [
        ("d_sky", C.c_void_p), ("sky_w", C.c_int32), ("sky_h", C.c_int32),
        ("d_disk_tex", C.c_void_p), ("disk_w", C.c_int32), ("disk_h", C.c_int32),
        ("disk_r_in", C.c_double), ("disk_r_out", C.c_double),
        ("disk_phase", C.c_double), ("disk_mean", C.c_double), ("disk_stddev", C.c_double),
        ("disk_intensity", C.c_double),
        ("n_spheres", C.c_int32), ("n_lamps", C.c_int32),
        ("spheres", (C.c_double * 4) * 8),
        ("sphere_rgb", (C.c_double * 3) * 8),
        ("lamps", (C.c_double * 4) * 4),
    ]


class FrameScene(C.Structure):
    """struct bhg_frame_scene (include/bhgeo.h): the scene of a library-owned frame, everything on the host."""
    _fields_ = [
        ("sky", C.c_void_p), ("sky_w", C.c_int32), ("sky_h", C.c_int32),
        ("disk_tex", C.c_void_p), ("disk_w", C.c_int32), ("disk_h", C.c_int32),
        ("disk_r_in", C.c_double), ("disk_r_out", C.c_double),
        ("disk_phase", C.c_double), ("disk_mean", C.c_double), ("disk_stddev", C.c_double),
        ("disk_intensity", C.c_double),
        ("n_spheres", C.c_int32), ("n_lamps", C.c_int32),
        ("spheres", (C.c_double * 4) * 8),
        ("sphere_rgb", (C.c_double * 3) * 8),
        ("lamps", (C.c_double * 4) * 4),
    ]


def make_scene(d_sky, sky_w, sky_h, *, d_disk_tex=0, disk_w=0, disk_h=0, disk=None, disk_phase=0.0, disk_mean=0.2,
               disk_stddev=0.3, disk_intensity=1.0, spheres=None, sphere_rgb=None, lamps=None):
    """disk=(R_in, R_out); spheres [[cx, cy, cz, radius]]; sphere_rgb [[r, g, b]] (default white);
    lamps [[x, y, z, intensity]].  Defaults of the disk profile: LimitedRelativisticRenderEngine.py:495-498."""
    sc = Scene()
    sc.d_sky, sc.sky_w, sc.sky_h = d_sky or None, int(sky_w), int(sky_h)
    sc.d_disk_tex, sc.disk_w, sc.disk_h = d_disk_tex or None, int(disk_w), int(disk_h)
    if disk is not None:
        sc.disk_r_in, sc.disk_r_out = float(disk[0]), float(disk[1])
    sc.disk_phase, sc.disk_mean, sc.disk_stddev, sc.disk_intensity = (float(disk_phase), float(disk_mean),
                                                                        float(disk_stddev), float(disk_intensity))
    sp = _spheres_array(spheres if spheres is not None else [])
    rgb = np.ones((len(sp), 3)) if sphere_rgb is None else np.asarray(sphere_rgb, dtype=np.float64).reshape(-1, 3)
    lm = np.zeros((0, 4)) if lamps is None else np.asarray(lamps, dtype=np.float64).reshape(-1, 4)
    if len(sp) > MAX_SPHERES or len(lm) > 4 or len(rgb) != len(sp):
        raise ValueError("at most 8 spheres (one colour each) and 4 lamps")
    sc.n_spheres, sc.n_lamps = len(sp), len(lm)
    for j in range(len(sp)):
        for q in range(4):
            sc.spheres[j][q] = float(sp[j, q])
        for q in range(3):
            sc.sphere_rgb[j][q] = float(rgb[j, q])
    for j in range(len(lm)):
        for q in range(4):
            sc.lamps[j][q] = float(lm[j, q])
    return sc


def _spheres_array(spheres):
    sp = np.ascontiguousarray(spheres, dtype=np.float64).reshape(-1, 4)
    return sp if len(sp) else np.zeros((1, 4))[:0]


class BhgError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libbhgeo error {code}: {msg}")
        self.code = code


class Params(C.Structure):
    """struct bhg_params (include/bhgeo.h)."""
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
        ("order_blocks", C.c_uint32),
        ("disk_r_in", C.c_double),
        ("disk_r_out", C.c_double),
        ("spin", C.c_double),
        ("time_like", C.c_int32),
        ("reserved0", C.c_int32),
    ]


_lib = None
_dp = C.POINTER(C.c_double)
_u8p = C.POINTER(C.c_uint8)
_u32p = C.POINTER(C.c_uint32)


def load():
    """dlopen libbhgeo.so and declare the prototypes.  Raises if the library is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C blackhole_geodesic_calculator_amd/csrc` (hipcc, --offload-arch=gfx950). "
            "There is no CPU fallback.")
    # One HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64 (SONAME
    # libamdhip64.so.7).  If torch is importable, import it first so that libbhgeo binds to that
    # already-loaded runtime; two runtimes in one process leave the second without a GPU.
    if "torch" not in sys.modules and not os.environ.get("BHGEO_NO_TORCH_PRELOAD"):
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    L = C.CDLL(LIB_PATH)
    L.bhg_version.restype = C.c_int
    L.bhg_device_count.restype = C.c_int
    L.bhg_last_error.restype = C.c_char_p
    L.bhg_default_params.restype = None
    L.bhg_default_params.argtypes = [C.POINTER(Params)]
    L.bhg_create.restype = C.c_int
    L.bhg_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    L.bhg_destroy.restype = None
    L.bhg_destroy.argtypes = [C.c_void_p]
    L.bhg_device_name.restype = C.c_int
    L.bhg_device_name.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    L.bhg_num_cus.restype = C.c_int
    L.bhg_num_cus.argtypes = [C.c_void_p]
    L.bhg_trace.restype = C.c_int
    L.bhg_trace.argtypes = [C.c_void_p, C.POINTER(Params), _dp, C.c_int, _dp, C.c_size_t, _dp, _u8p,
                            _u32p, _u32p]
    L.bhg_trace_device.restype = C.c_int
    L.bhg_trace_device.argtypes = [C.c_void_p, C.POINTER(Params), _dp, C.c_void_p, C.c_void_p,
                                   C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p]
    L.bhg_trace_dir_device.restype = C.c_int
    L.bhg_trace_dir_device.argtypes = L.bhg_trace_device.argtypes
    L.bhg_shade_dir_device.restype = C.c_int
    L.bhg_shade_dir_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int32, C.c_void_p,
                                       C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.bhg_trace_objects.restype = C.c_int
    L.bhg_trace_objects.argtypes = [C.c_void_p, C.POINTER(Params), _dp, C.c_int32, _dp, C.c_int, _dp, C.c_size_t, _dp,
                                    _u8p, _u32p, _u32p, C.POINTER(C.c_int8)]
    L.bhg_trace_objects_device.restype = C.c_int
    L.bhg_trace_objects_device.argtypes = [C.c_void_p, C.POINTER(Params), _dp, C.c_int32, _dp, C.c_void_p, C.c_void_p,
                                           C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p]
    L.bhg_shade_scene_device.restype = C.c_int
    L.bhg_shade_scene_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int32,
                                         C.POINTER(Scene), C.c_void_p, C.c_void_p]
    L.bhg_shade_scene_f32_device.restype = C.c_int
    L.bhg_shade_scene_f32_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int32,
                                             C.POINTER(Scene), C.c_void_p, C.c_void_p, C.c_void_p]
    L.bhg_assemble_frame_f32_device.restype = C.c_int
    L.bhg_assemble_frame_f32_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    L.bhg_raygen_device.restype = C.c_int
    L.bhg_raygen_device.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_double, _dp,
                                    C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    L.bhg_shade_device.restype = C.c_int
    L.bhg_shade_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int32, C.c_void_p,
                                   C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    L.bhg_trajectory.restype = C.c_int
    # (raw addresses: building a typed ctypes pointer costs ~2 us per array, and the engine's literal call is one ray long)
    L.bhg_trajectory.argtypes = [C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_void_p]
    L.bhg_trajectory_objects.restype = C.c_int
    L.bhg_trajectory_objects.argtypes = [C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_int32, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t,
                                         C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.bhg_set_profiling.restype = C.c_int
    L.bhg_set_profiling.argtypes = [C.c_void_p, C.c_int]
    L.bhg_last_pass_ms.restype = C.c_int
    L.bhg_last_pass_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    L.bhg_acceleration.restype = C.c_int
    L.bhg_acceleration.argtypes = [C.c_void_p, C.POINTER(Params), _dp, _dp, C.c_size_t, _dp]
    L.bhg_rays_create.restype = C.c_int
    L.bhg_rays_create.argtypes = [C.c_void_p, C.POINTER(Camera), _dp, C.c_int, C.POINTER(C.c_int64), C.c_size_t,
                                  C.POINTER(C.c_void_p)]
    L.bhg_rays_count.restype = C.c_size_t
    L.bhg_rays_count.argtypes = [C.c_void_p]
    L.bhg_rays_destroy.restype = None
    L.bhg_rays_destroy.argtypes = [C.c_void_p]
    L.bhg_rays_trace.restype = C.c_int
    L.bhg_rays_trace.argtypes = [C.c_void_p, C.POINTER(Params), _dp, C.c_int32, C.c_size_t, C.c_size_t, _dp, _dp, _dp, _u8p,
                                 _u32p, _u32p, C.POINTER(C.c_int8)]
    L.bhg_host_alloc.restype = C.c_int
    L.bhg_host_alloc.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
    L.bhg_host_free.restype = C.c_int
    L.bhg_host_free.argtypes = [C.c_void_p, C.c_void_p]
    L.bhg_synchronize.restype = C.c_int
    L.bhg_synchronize.argtypes = [C.c_void_p]
    L.bhg_context_stream.restype = C.c_void_p
    L.bhg_context_stream.argtypes = [C.c_void_p]
    L.bhg_last_launch.restype = C.c_int
    L.bhg_last_launch.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
    L.bhg_frame_create.restype = C.c_int
    L.bhg_frame_create.argtypes = [C.POINTER(C.c_int32), C.c_int32, C.POINTER(Camera), _dp, C.c_int32, C.c_int32,
                                   C.POINTER(C.c_void_p)]
    L.bhg_frame_destroy.restype = None
    L.bhg_frame_destroy.argtypes = [C.c_void_p]
    L.bhg_frame_set_scene.restype = C.c_int
    L.bhg_frame_set_scene.argtypes = [C.c_void_p, C.POINTER(FrameScene)]
    L.bhg_frame_set_camera.restype = C.c_int
    L.bhg_frame_set_camera.argtypes = [C.c_void_p, C.POINTER(Camera)]
    L.bhg_frame_render.restype = C.c_int
    L.bhg_frame_render.argtypes = [C.c_void_p, C.POINTER(Params), C.c_void_p]
    L.bhg_frame_synchronize.restype = C.c_int
    L.bhg_frame_synchronize.argtypes = [C.c_void_p]
    L.bhg_frame_device_image.restype = C.c_void_p
    L.bhg_frame_device_image.argtypes = [C.c_void_p]
    L.bhg_frame_rebalance.restype = C.c_int
    L.bhg_frame_rebalance.argtypes = [C.c_void_p, C.c_double]
    L.bhg_frame_stats.restype = C.c_int
    L.bhg_frame_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    L.bhg_frame_info.restype = C.c_int
    L.bhg_frame_info.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    L.bhg_frame_set_profiling.restype = C.c_int
    L.bhg_frame_set_profiling.argtypes = [C.c_void_p, C.c_int]
    L.bhg_frame_last_ms.restype = C.c_int
    L.bhg_frame_last_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.bhg_deal_tiles.restype = C.c_int
    L.bhg_deal_tiles.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, _dp, C.c_int32, C.c_double, C.c_int32,
                                 C.POINTER(C.c_int64), C.c_size_t, C.POINTER(C.c_size_t)]
    if L.bhg_version() != ABI_VERSION or not hasattr(L, "bhg_abi_check"):
        raise ImportError(f"libbhgeo ABI {L.bhg_version()} != expected {ABI_VERSION}: rebuild {LIB_PATH}")
    for name in ("bhg_params_size", "bhg_camera_size", "bhg_scene_size", "bhg_frame_scene_size"):
        getattr(L, name).restype = C.c_size_t
        getattr(L, name).argtypes = []
    L.bhg_abi_check.restype = C.c_int
    L.bhg_abi_check.argtypes = [C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t]
    L.bhg_default_params_sized.restype = C.c_int
    L.bhg_default_params_sized.argtypes = [C.POINTER(Params), C.c_size_t]
    L.bhg_peak_probe.restype = C.c_int
    L.bhg_peak_probe.argtypes = [C.c_void_p, C.c_int32, C.c_double, _dp]
    # the handshake include/bhgeo.h asks of every binding: ABI version and the layout of every struct declared above
    if L.bhg_abi_check(ABI_VERSION, C.sizeof(Params), C.sizeof(Camera), C.sizeof(Scene), C.sizeof(FrameScene)) != OK:
        raise ImportError("libbhgeo: " + L.bhg_last_error().decode())
    _lib = L
    return L


def _check(rc):
    if rc != OK:
        raise BhgError(rc, load().bhg_last_error().decode())


def default_params() -> Params:
    p = Params()
    _check(load().bhg_default_params_sized(C.byref(p), C.sizeof(p)))
    return p


def make_params(r_s=1.0, lambda_end=50.0, max_step=np.inf, rtol=1e-3, atol=1e-6, h_fixed=0.1,
                r_exit=0.0, method=METHOD_DP54, rhs_form=RHS_CHRISTOFFEL, max_steps=0, disk_r_in=0.0,
                disk_r_out=0.0, spin=0.0, order_blocks=0, time_like=0) -> Params:
    return Params(float(r_s), float(lambda_end), float(max_step), float(rtol), float(atol),
                  float(h_fixed), float(r_exit), int(method), int(rhs_form), int(max_steps), int(order_blocks),
                  float(disk_r_in), float(disk_r_out), float(spin), int(bool(time_like)), 0)


def device_count() -> int:
    return load().bhg_device_count()


def _np_dp(a):
    return a.ctypes.data_as(_dp)


def _addr(a):
    """The address of a numpy array's first element (for void* parameters)."""
    return a.__array_interface__["data"][0]


class _PinnedBlock:
    """One bhg_host_alloc block.  Page-locking is slow (tens of ms for a few hundred MB), so blocks are pooled:
    when the last numpy view of a block dies, the block goes back to its pool's free list instead of to the OS."""

    def __init__(self, nbytes, pool):
        p = C.c_void_p()
        # (the owning context: the allocation is made with ITS device current, not device 0)
        _check(load().bhg_host_alloc(pool.ctx_handle() if pool is not None else None, int(nbytes), C.byref(p)))
        self.ptr, self.nbytes, self.pool = p.value, int(nbytes), pool

    def release(self):
        if self.ptr:
            load().bhg_host_free(None, C.c_void_p(self.ptr))
            self.ptr = None


class PinnedPool:
    """Page-locked result arrays for the host-buffer calls (Context.trace): numpy arrays over bhg_host_alloc
    memory, so the device's copy engines write results straight into what the caller gets -- no staging copy."""

    def __init__(self, max_free_bytes=2 << 30, ctx=None):
        self._free = {}          # nbytes -> [blocks]
        self._free_bytes = 0
        self.max_free_bytes = int(max_free_bytes)
        self._ctx = ctx          # the owning Context (None: allocations are made on the current device)
        self._closed = False

    def ctx_handle(self):
        return getattr(self._ctx, "_h", None) if self._ctx is not None else None

    def _give_back(self, blk):
        if blk.ptr is None:
            return
        if self._closed or self._free_bytes + blk.nbytes > self.max_free_bytes:
            blk.release()        # (a block that comes home after Context.close() is freed, not pooled for nobody)
            return
        self._free.setdefault(blk.nbytes, []).append(blk)
        self._free_bytes += blk.nbytes

    def empty(self, shape, dtype):
        import weakref
        dtype = np.dtype(dtype)
        n = int(np.prod(shape)) * dtype.itemsize
        if n == 0:
            return np.empty(shape, dtype)
        size = (n + 4095) & ~4095
        lst = self._free.get(size)
        if lst:
            blk = lst.pop()
            self._free_bytes -= blk.nbytes
        else:
            try:
                blk = _PinnedBlock(size, self)
            except BhgError:
                # page-locking failed (pin limit, a 67-M-ray frame): a pageable array still works, it crosses the
                # library's staging ring instead of being written by the copy engines directly
                return np.empty(shape, dtype)
        buf = (C.c_char * n).from_address(blk.ptr)
        weakref.finalize(buf, PinnedPool._give_back, self, blk)   # the array's base chain holds `buf`
        return np.frombuffer(buf, dtype=dtype).reshape(shape)

    def clear(self):
        for lst in self._free.values():
            for blk in lst:
                blk.release()
        self._free.clear()
        self._free_bytes = 0

    def close(self):
        self.clear()
        self._closed = True


class RaySet:
    """bhg_rays: the camera rays of one frame, generated on the device and resident there (include/bhgeo.h).
    Ray s * n_pixels + p is sample s of pixel p."""

    def __init__(self, ctx: "Context", width, height, samples, fov_x, fov_y, origin, rot=None, jitter=None,
                 jitter_is_compact=False, pixels=None):
        cam = Camera()
        cam.width, cam.height, cam.samples = int(width), int(height), int(samples)
        cam.fov_x, cam.fov_y = float(fov_x), float(fov_y)
        r = np.eye(3) if rot is None else np.asarray(rot, dtype=np.float64).reshape(3, 3)
        cam.rot[:] = [float(v) for v in r.reshape(9)]
        cam.origin[:] = [float(v) for v in np.asarray(origin, dtype=np.float64).reshape(3)]
        jit = None if jitter is None else np.ascontiguousarray(jitter, dtype=np.float64).reshape(-1)
        pix = None if pixels is None else np.ascontiguousarray(pixels, dtype=np.int64).reshape(-1)
        n_pixels = int(width) * int(height) if pix is None else len(pix)
        need = 2 * int(samples) * (n_pixels if jitter_is_compact else int(width) * int(height))
        if jit is not None and len(jit) < need:
            raise ValueError(f"jitter stream too short: {len(jit)} < {need}")
        h = C.c_void_p()
        _check(load().bhg_rays_create(ctx._h, C.byref(cam), None if jit is None else _np_dp(jit), 1 if jitter_is_compact else 0,
                                      None if pix is None else pix.ctypes.data_as(C.POINTER(C.c_int64)), n_pixels, C.byref(h)))
        self._h, self.ctx = h, ctx
        self.n_pixels, self.samples = n_pixels, int(samples)
        self.n = int(load().bhg_rays_count(h))

    def close(self):
        if getattr(self, "_h", None):
            if getattr(self.ctx, "_h", None):    # the context frees nothing of ours; but never touch a dead one
                load().bhg_rays_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def trace(self, params: "Params", first=0, n=None, want=("end", "flags", "n_steps", "n_accepted"), spheres=None):
        """Trace rays [first, first + n); returns a dict of the arrays named in `want` (of: end [n,6], end_loc [n,3],
        end_dir [n,3], flags, n_steps, n_accepted, object_id), page-locked from the context's pool."""
        n = self.n - int(first) if n is None else int(n)
        pool = self.ctx.pinned
        new = pool.empty if n >= 65536 else (lambda shape, dt: np.empty(shape, dt))
        spec = {"end": ((n, 6), np.float64), "end_loc": ((n, 3), np.float64), "end_dir": ((n, 3), np.float64),
                "flags": ((n,), np.uint8), "n_steps": ((n,), np.uint32), "n_accepted": ((n,), np.uint32),
                "object_id": ((n,), np.int8)}
        out = {k: new(*spec[k]) for k in want}

        def ptr(k, typ):
            return out[k].ctypes.data_as(typ) if k in out else None
        sp = None if spheres is None else _spheres_array(spheres)
        _check(load().bhg_rays_trace(self._h, C.byref(params), None if sp is None else _np_dp(sp), 0 if sp is None else len(sp),
                                     int(first), n, ptr("end", _dp), ptr("end_loc", _dp), ptr("end_dir", _dp), ptr("flags", _u8p),
                                     ptr("n_steps", _u32p), ptr("n_accepted", _u32p), ptr("object_id", C.POINTER(C.c_int8))))
        return out


def deal_tiles(width, height, tile, world, rank, tile_cost=None, visit_by_cost=True, root_share=1.0):
    """bhg_deal_tiles: the pixel list of device `rank` of a library-owned frame (host logic, no GPU needed).
    tile_cost: None or one figure per tile, row-major over the tile grid; root_share: rank 0's part of an equal share."""
    n = C.c_size_t()
    tc = None if tile_cost is None else np.ascontiguousarray(tile_cost, dtype=np.float64).reshape(-1)
    args = (int(width), int(height), int(tile), int(world), None if tc is None else _np_dp(tc), 1 if visit_by_cost else 0,
            float(root_share), int(rank))
    _check(load().bhg_deal_tiles(*args, None, 0, C.byref(n)))
    px = np.empty(n.value, dtype=np.int64)
    _check(load().bhg_deal_tiles(*args, px.ctypes.data_as(C.POINTER(C.c_int64)), px.size, C.byref(n)))
    return px


class Frame:
    """bhg_frame: a whole frame owned by the library -- jitter stream -> rays -> geodesics -> shaded, sample-averaged
    float RGBA pixels in frame order -- on one or several GPUs of this one process (include/bhgeo.h).  No torch.

    devices: device indices; an index may be repeated ({0, 0}: several contexts of one GPU, the N > 1 code path on a
    one-GPU machine).  jitter: the full-frame stream [S*H*W*2] of random.random() draws, or None = pixel centres.
    origin is BH-centred; rot a 3x3 rotation matrix (None = identity)."""

    def __init__(self, devices, width, height, samples, *, fov_x=1.0, fov_y=1.0, origin=(1e-4, 0.0, 30.0), rot=None,
                 jitter=None, tile=32, gather=GATHER_AUTO):
        cam = self._camera(width, height, samples, fov_x, fov_y, origin, rot)
        devs = [int(d) for d in (devices if hasattr(devices, "__len__") else [devices])]
        jit = None if jitter is None else np.ascontiguousarray(jitter, dtype=np.float64).reshape(-1)
        need = 2 * cam.samples * cam.width * cam.height
        if jit is not None and len(jit) < need:
            raise ValueError(f"jitter stream too short: {len(jit)} < {need}")
        h = C.c_void_p()
        _check(load().bhg_frame_create((C.c_int32 * len(devs))(*devs), len(devs), C.byref(cam),
                                       None if jit is None else _np_dp(jit), int(tile), int(gather), C.byref(h)))
        self._h = h
        self.devices, self.W, self.H, self.S = devs, cam.width, cam.height, cam.samples
        self._scene = FrameScene()
        self._scene.disk_mean, self._scene.disk_stddev, self._scene.disk_intensity = 0.2, 0.3, 1.0

    @staticmethod
    def _camera(width, height, samples, fov_x, fov_y, origin, rot):
        cam = Camera()
        cam.width, cam.height, cam.samples = int(width), int(height), int(samples)
        cam.fov_x, cam.fov_y = float(fov_x), float(fov_y)
        r = np.eye(3) if rot is None else np.asarray(rot, dtype=np.float64).reshape(3, 3)
        cam.rot[:] = [float(v) for v in r.reshape(9)]
        cam.origin[:] = [float(v) for v in np.asarray(origin, dtype=np.float64).reshape(3)]
        return cam

    def set_camera(self, *, fov_x, fov_y, origin, rot=None):
        """Move the camera (bhg_frame_set_camera): the frame, its jitter stream and its buffers stay.  A new origin is free;
        a new rotation / field of view regenerates the rays on the devices at the next render."""
        cam = self._camera(self.W, self.H, self.S, fov_x, fov_y, origin, rot)
        _check(load().bhg_frame_set_camera(self._h, C.byref(cam)))

    def close(self):
        if getattr(self, "_h", None):
            load().bhg_frame_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_scene(self, sky=None, *, disk=None, disk_tex=None, disk_phase=0.0, disk_mean=0.2, disk_stddev=0.3,
                  disk_intensity=1.0, spheres=None, sphere_rgb=None, lamps=None):
        """sky [h, w, 4] float32 equirectangular (None: keep the current one); disk = (R_in, R_out) or None; disk_tex
        [h, w, 4] float32 or None; spheres [[cx, cy, cz, radius]] BH-centred, sphere_rgb (default white), lamps
        [[x, y, z, intensity]].  The whole scene is replaced by each call (images are kept when not given)."""
        sc = FrameScene()
        keep = []
        if sky is not None:
            a = np.ascontiguousarray(sky, dtype=np.float32)
            assert a.ndim == 3 and a.shape[2] == 4
            sc.sky, sc.sky_w, sc.sky_h = a.ctypes.data, a.shape[1], a.shape[0]
            keep.append(a)
        if disk_tex is not None:
            t = np.ascontiguousarray(disk_tex, dtype=np.float32)
            assert t.ndim == 3 and t.shape[2] == 4
            sc.disk_tex, sc.disk_w, sc.disk_h = t.ctypes.data, t.shape[1], t.shape[0]
            keep.append(t)
        if disk is not None:
            sc.disk_r_in, sc.disk_r_out = float(disk[0]), float(disk[1])
        sc.disk_phase, sc.disk_mean, sc.disk_stddev, sc.disk_intensity = (float(disk_phase), float(disk_mean),
                                                                            float(disk_stddev), float(disk_intensity))
        sp = _spheres_array(spheres if spheres is not None else [])
        rgb = np.ones((len(sp), 3)) if sphere_rgb is None else np.asarray(sphere_rgb, dtype=np.float64).reshape(-1, 3)
        lm = np.zeros((0, 4)) if lamps is None else np.asarray(lamps, dtype=np.float64).reshape(-1, 4)
        if len(sp) > MAX_SPHERES or len(lm) > 4 or len(rgb) != len(sp):
            raise ValueError("at most 8 spheres (one colour each) and 4 lamps")
        sc.n_spheres, sc.n_lamps = len(sp), len(lm)
        for j in range(len(sp)):
            for q in range(4):
                sc.spheres[j][q] = float(sp[j, q])
            for q in range(3):
                sc.sphere_rgb[j][q] = float(rgb[j, q])
        for j in range(len(lm)):
            for q in range(4):
                sc.lamps[j][q] = float(lm[j, q])
        _check(load().bhg_frame_set_scene(self._h, C.byref(sc)))

    def render(self, params: "Params", out=None, to_host=True):
        """One frame: float32 [H, W, 4] (a new array, or `out`).  to_host=False: only enqueue; the image stays on the
        first device (device_image(), synchronize())."""
        if not to_host:
            _check(load().bhg_frame_render(self._h, C.byref(params), None))
            return None
        if out is None:
            out = np.empty((self.H, self.W, 4), dtype=np.float32)
        assert out.dtype == np.float32 and out.flags["C_CONTIGUOUS"] and out.size == self.H * self.W * 4
        _check(load().bhg_frame_render(self._h, C.byref(params), C.c_void_p(out.ctypes.data)))
        return out

    def synchronize(self):
        _check(load().bhg_frame_synchronize(self._h))

    def device_image(self) -> int:
        return load().bhg_frame_device_image(self._h) or 0

    def rebalance(self, root_share=1.0):
        """Re-deal the tiles by the last render's measured cost; root_share < 1 gives the first device -- which also
        receives the gather and assembles the frame -- that part of an equal share."""
        _check(load().bhg_frame_rebalance(self._h, float(root_share)))

    def stats(self):
        out = (C.c_uint64 * 4)()
        _check(load().bhg_frame_stats(self._h, out))
        return {"rays": int(out[0]), "attempted_steps": int(out[1]), "accepted_steps": int(out[2]), "horizon_rays": int(out[3])}

    def info(self):
        out = (C.c_int64 * 8)()
        _check(load().bhg_frame_info(self._h, out))
        return {"n_devices": int(out[0]), "gather": {GATHER_COPY: "copy", GATHER_RCCL: "rccl", GATHER_PEER: "peer"}.get(int(out[1]), str(out[1])),
                "largest_shard_pixels": int(out[2]), "smallest_shard_pixels": int(out[3]), "tile": int(out[4]),
                "dealt_by_measured_cost": bool(out[5]), "renders": int(out[6]), "directions_only": bool(out[7])}

    def set_profiling(self, enable=True):
        _check(load().bhg_frame_set_profiling(self._h, 1 if enable else 0))

    def last_ms(self):
        """(trace kernel ms per listed device, root gather + assembly ms) of the last profiled render."""
        tr = (C.c_float * len(self.devices))()
        root = C.c_float()
        _check(load().bhg_frame_last_ms(self._h, tr, C.byref(root)))
        return [float(v) for v in tr], float(root.value)


class Context:
    """One bhg_context = one device + stream.  Not thread-safe; one call at a time."""

    def __init__(self, device: int = 0):
        L = load()
        h = C.c_void_p()
        _check(L.bhg_create(int(device), C.byref(h)))
        self._h = h
        self.device = int(device)
        self.pinned = PinnedPool(ctx=self)

    def close(self):
        if getattr(self, "_h", None):
            self.pinned.close()
            load().bhg_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def name(self) -> str:
        buf = C.create_string_buffer(256)
        _check(load().bhg_device_name(self._h, buf, 256))
        return buf.value.decode()

    @property
    def num_cus(self) -> int:
        return load().bhg_num_cus(self._h)

    def last_launch(self):
        out = (C.c_int32 * 4)()
        _check(load().bhg_last_launch(self._h, out))
        return {"workgroups": out[0], "threads": out[1], "waves_per_cu": out[2], "passes": out[3]}

    @property
    def stream(self) -> int:
        """The context's own hipStream_t as an integer handle."""
        return load().bhg_context_stream(self._h) or 0

    def set_profiling(self, enable=True):
        _check(load().bhg_set_profiling(self._h, 1 if enable else 0))

    def last_pass_ms(self):
        """{prepare, trace, post} milliseconds of the last profiled trace call (waits for it); post = Kerr's finalize
        pass, 0 otherwise ("resolve" is the slot's round-1 name, kept as an alias)."""
        out = (C.c_float * 3)()
        _check(load().bhg_last_pass_ms(self._h, out))
        return {"prepare": out[0], "trace": out[1], "post": out[2], "resolve": out[2]}

    def synchronize(self):
        _check(load().bhg_synchronize(self._h))

    def peak_probe(self, kind=PROBE_FMA, target_ms=1.0):
        """bhg_peak_probe: the fp64 VALU rate THIS device sustains, in the trace kernels' launch geometry.  kind
        PROBE_FMA: nothing but v_fma_f64; PROBE_STEP_MIX: the DP5(4) step loop's mix (16 quarter-rate ops per 503)."""
        out = (C.c_double * 6)()
        _check(load().bhg_peak_probe(self._h, int(kind), float(target_ms), out))
        return {"tflops": out[0], "ms": out[1], "valu_wave_insts": out[2], "quarter_rate_wave_insts": out[3],
                "ms_fastest": out[4], "fp64_full_rate_clock_mhz": out[5]}

    # -- host buffers -------------------------------------------------------------------
    def trace(self, k0, x0, params: Params, want_accepted=True, spheres=None, want_steps=True, pinned_results=True):
        """k0[N,3], x0[3] (shared) or [N,3] -> (end[N,6], flags[N] u8, n_steps[N] u32, n_accepted[N] u32);
        with spheres [[cx, cy, cz, radius], ...] a fifth array object_id[N] i8 is appended.  want_steps /
        want_accepted = False: that array is not brought back (None in its place).  The result arrays are
        page-locked (self.pinned, a pool) unless pinned_results=False."""
        k0 = np.ascontiguousarray(k0, dtype=np.float64)
        if k0.ndim != 2 or k0.shape[1] != 3:
            raise ValueError("k0 must have shape [N, 3]")
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        n = k0.shape[0]
        shared = x0.ndim == 1
        if shared:
            if x0.shape != (3,):
                raise ValueError("x0 must have shape [3] or [N, 3]")
        elif x0.shape != (n, 3):
            raise ValueError("x0 must have shape [3] or [N, 3]")
        new = self.pinned.empty if (pinned_results and n >= 65536) else (lambda shape, dt: np.empty(shape, dt))
        end = new((n, 6), np.float64)
        flags = new((n,), np.uint8)
        steps = new((n,), np.uint32) if want_steps else None
        acc = new((n,), np.uint32) if want_accepted else None
        p_steps = steps.ctypes.data_as(_u32p) if steps is not None else None
        p_acc = acc.ctypes.data_as(_u32p) if acc is not None else None
        if spheres is not None:
            sp = _spheres_array(spheres)
            obj = new((n,), np.int8)
            _check(load().bhg_trace_objects(self._h, C.byref(params), _np_dp(sp), len(sp), _np_dp(x0), 1 if shared else 0,
                                            _np_dp(k0), n, _np_dp(end), flags.ctypes.data_as(_u8p), p_steps, p_acc,
                                            obj.ctypes.data_as(C.POINTER(C.c_int8))))
            return end, flags, steps, acc, obj
        _check(load().bhg_trace(self._h, C.byref(params), _np_dp(x0), 1 if shared else 0, _np_dp(k0), n,
                                _np_dp(end), flags.ctypes.data_as(_u8p), p_steps, p_acc))
        return end, flags, steps, acc

    def trajectory(self, k0, x0, params: Params, n_points, spheres=None):
        """Sampled curves: (traj[N,6,T], n_valid[N], end[N,6], flags[N]); with spheres= (object spheres in the curved region,
        [[cx, cy, cz, radius], ...] BH-centred): (..., object_id[N]) -- bhg_trajectory_objects."""
        k0 = np.ascontiguousarray(k0, dtype=np.float64).reshape(-1, 3)
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        n = k0.shape[0]
        shared = x0.ndim == 1
        # small calls (the engine's literal one ray x 10,000 samples): a page-locked sample array from the context's pool -- the
        # kernel then writes the samples straight into it (no device-to-host copy, no host-side split; include/bhgeo.h)
        small = n <= 2048 and n * 6 * int(n_points) * 8 <= (4 << 20)
        traj = (self.pinned.empty if small else np.empty)((n, 6, int(n_points)), np.float64)
        nv = np.empty(n, np.uint32)
        end = np.empty((n, 6), np.float64)
        flags = np.empty(n, np.uint8)
        if spheres is not None:
            sp = _spheres_array(spheres)
            obj = np.empty(n, np.int8)
            _check(load().bhg_trajectory_objects(self._h, C.byref(params), _addr(sp) if len(sp) else None, len(sp), _addr(x0),
                                                 1 if shared else 0, _addr(k0), n, int(n_points), _addr(traj), _addr(nv), _addr(end),
                                                 _addr(flags), _addr(obj)))
            return traj, nv, end, flags, obj
        _check(load().bhg_trajectory(self._h, C.byref(params), _addr(x0), 1 if shared else 0, _addr(k0), n,
                                     int(n_points), _addr(traj), _addr(nv), _addr(end), _addr(flags)))
        return traj, nv, end, flags

    # -- device buffers (raw addresses, e.g. torch.Tensor.data_ptr()) -------------------
    def trace_device(self, params: Params, n, d_k0, d_end, x0_shared=None, d_x0=0, d_flags=0,
                     d_n_steps=0, d_n_accepted=0, stream=0, spheres=None, d_object_id=0):
        xs = None
        if x0_shared is not None:
            xs = (C.c_double * 3)(*[float(v) for v in x0_shared])
        if spheres is not None:
            sp = _spheres_array(spheres)
            _check(load().bhg_trace_objects_device(self._h, C.byref(params), _np_dp(sp), len(sp), xs,
                                                   C.c_void_p(d_x0 or None), C.c_void_p(d_k0), int(n), C.c_void_p(d_end),
                                                   C.c_void_p(d_flags or None), C.c_void_p(d_n_steps or None),
                                                   C.c_void_p(d_n_accepted or None), C.c_void_p(d_object_id or None),
                                                   C.c_void_p(stream or None)))
            return
        _check(load().bhg_trace_device(self._h, C.byref(params), xs, C.c_void_p(d_x0 or None),
                                       C.c_void_p(d_k0), int(n), C.c_void_p(d_end),
                                       C.c_void_p(d_flags or None), C.c_void_p(d_n_steps or None),
                                       C.c_void_p(d_n_accepted or None), C.c_void_p(stream or None)))

    def trace_dir_device(self, params: Params, n, d_k0, d_end_dir, x0_shared=None, d_x0=0, d_flags=0,
                         d_n_steps=0, d_n_accepted=0, stream=0):
        """bhg_trace_dir_device: like trace_device, but only the direction half of the end states is written
        (d_end_dir [n][3]) -- what a sky frame consumes."""
        xs = None
        if x0_shared is not None:
            xs = (C.c_double * 3)(*[float(v) for v in x0_shared])
        _check(load().bhg_trace_dir_device(self._h, C.byref(params), xs, C.c_void_p(d_x0 or None),
                                           C.c_void_p(d_k0), int(n), C.c_void_p(d_end_dir),
                                           C.c_void_p(d_flags or None), C.c_void_p(d_n_steps or None),
                                           C.c_void_p(d_n_accepted or None), C.c_void_p(stream or None)))

    def shade_dir_device(self, d_end_dir, d_flags, n_pixels, samples, d_sky, sky_w, sky_h, d_rgba=0, d_rgba_f32=0,
                         d_scatter=0, stream=0):
        _check(load().bhg_shade_dir_device(self._h, C.c_void_p(d_end_dir), C.c_void_p(d_flags), int(n_pixels),
                                           int(samples), C.c_void_p(d_sky), int(sky_w), int(sky_h),
                                           C.c_void_p(d_rgba or None), C.c_void_p(d_rgba_f32 or None),
                                           C.c_void_p(d_scatter or None), C.c_void_p(stream or None)))

    def raygen_device(self, width, height, samples, fov_x, fov_y, d_jitter, d_k0, n_pixels, d_pixels=0,
                      rot=None, stream=0):
        r9 = None
        if rot is not None:
            r9 = (C.c_double * 9)(*[float(v) for v in np.asarray(rot, dtype=np.float64).reshape(9)])
        _check(load().bhg_raygen_device(self._h, int(width), int(height), int(samples), float(fov_x), float(fov_y),
                                        r9, C.c_void_p(d_jitter), C.c_void_p(d_pixels or None), int(n_pixels),
                                        C.c_void_p(d_k0), C.c_void_p(stream or None)))

    def shade_device(self, d_end, d_flags, n_pixels, samples, d_sky, sky_w, sky_h, d_rgba, stream=0):
        _check(load().bhg_shade_device(self._h, C.c_void_p(d_end), C.c_void_p(d_flags), int(n_pixels), int(samples),
                                       C.c_void_p(d_sky), int(sky_w), int(sky_h), C.c_void_p(d_rgba),
                                       C.c_void_p(stream or None)))

    def shade_scene_device(self, d_end, d_flags, n_pixels, samples, scene: "Scene", d_rgba, d_object_id=0, stream=0):
        _check(load().bhg_shade_scene_device(self._h, C.c_void_p(d_end), C.c_void_p(d_flags),
                                             C.c_void_p(d_object_id or None), int(n_pixels), int(samples),
                                             C.byref(scene), C.c_void_p(d_rgba), C.c_void_p(stream or None)))

    def shade_scene_f32_device(self, d_end, d_flags, n_pixels, samples, scene: "Scene", d_rgba_f32, d_object_id=0,
                               d_scatter=0, stream=0):
        _check(load().bhg_shade_scene_f32_device(self._h, C.c_void_p(d_end), C.c_void_p(d_flags),
                                                 C.c_void_p(d_object_id or None), int(n_pixels), int(samples),
                                                 C.byref(scene), C.c_void_p(d_rgba_f32), C.c_void_p(d_scatter or None),
                                                 C.c_void_p(stream or None)))

    def assemble_frame_f32_device(self, d_slabs, d_index, n_pixels, d_frame, stream=0):
        _check(load().bhg_assemble_frame_f32_device(self._h, C.c_void_p(d_slabs), C.c_void_p(d_index), int(n_pixels),
                                                    C.c_void_p(d_frame), C.c_void_p(stream or None)))

    def acceleration(self, x, k, params: Params):
        x = np.ascontiguousarray(np.atleast_2d(x), dtype=np.float64)
        k = np.ascontiguousarray(np.atleast_2d(k), dtype=np.float64)
        a = np.empty_like(x)
        _check(load().bhg_acceleration(self._h, C.byref(params), _np_dp(x), _np_dp(k), x.shape[0], _np_dp(a)))
        return a
