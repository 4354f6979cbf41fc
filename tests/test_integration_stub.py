"""The ctypes stubs INTEGRATION.md section 2 prints are EXECUTED here, so that they cannot drift from include/bhgeo.h.

Round 4's copy of the first stub still declared the 96-byte bhg_params of ABI 5 against a 104-byte ABI-6 library:
bhg_default_params wrote 8 bytes past the struct and no test noticed, because no test ran the text a maintainer of the
reference (raytracer/RelativisticRenderEngine.py:134, :293-294) would paste.  CPU: declarations + handshake against the
built library, with a canary behind the struct.  GPU: the stubs' trace() and frame sequence against `_ffi`."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from blackhole_geodesic_calculator_amd import _ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GPU_MARK = "# --- from here on a GPU is needed"


def _blocks():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("## 2. C-ABI level"):text.index("## 3. Multi-GPU")]
    blocks = re.findall(r"```python\n(.*?)```", sec, re.S)
    assert len(blocks) == 2, "INTEGRATION.md section 2 is expected to hold the trace stub and the frame stub"
    trace_stub, frame_stub = blocks
    assert 'C.CDLL("libbhgeo.so")' in trace_stub and GPU_MARK in trace_stub
    # the one substitution: where the library lives in this tree
    trace_stub = trace_stub.replace('C.CDLL("libbhgeo.so")', f"C.CDLL({_ffi.LIB_PATH!r})")
    return trace_stub, frame_stub


def test_stub_declarations_match_the_library_layout():
    """The CPU half of the trace stub: struct declaration, handshake, bhg_default_params_sized -- and a canary."""
    _ffi.load()                      # (the one HIP runtime of the process, as every other test loads it)
    trace_stub, frame_stub = _blocks()
    ns = {}
    exec(trace_stub[:trace_stub.index(GPU_MARK)], ns)
    lib, P = ns["lib"], ns["bhg_params"]
    lib.bhg_params_size.restype = C.c_size_t
    assert C.sizeof(P) == lib.bhg_params_size() == C.sizeof(_ffi.Params)
    assert [f[0] for f in P._fields_] == [f[0] for f in _ffi.Params._fields_]
    assert [f[1] for f in P._fields_] == [f[1] for f in _ffi.Params._fields_]
    assert ns["p"].r_s == 1.0 and ns["p"].lambda_end == 50.0 and ns["p"].rtol == 1e-3 and ns["p"].time_like == 0

    # a canary directly behind the stub's struct: the library's default writer must not touch it
    class Guarded(C.Structure):
        _fields_ = [("p", P), ("canary", C.c_uint8 * 64)]
    g = Guarded()
    C.memset(C.byref(g), 0xA5, C.sizeof(g))
    lib.bhg_default_params.restype = None
    lib.bhg_default_params(C.byref(g.p))
    assert bytes(g.canary) == b"\xa5" * 64
    assert g.p.r_s == 1.0 and g.p.reserved0 == 0

    # the frame stub's structs (declarations only: everything up to the first call that creates something)
    decl = frame_stub[:frame_stub.index("cam = bhg_camera(")]
    exec(decl, ns)
    for name, fn in (("bhg_camera", "bhg_camera_size"), ("bhg_frame_scene", "bhg_frame_scene_size")):
        getattr(lib, fn).restype = C.c_size_t
        assert C.sizeof(ns[name]) == getattr(lib, fn)()
    assert C.sizeof(ns["bhg_camera"]) == C.sizeof(_ffi.Camera) and C.sizeof(ns["bhg_frame_scene"]) == C.sizeof(_ffi.FrameScene)


def test_handshake_names_both_sizes_on_a_mismatch():
    """What round 4's stale stub would have met: a 96-byte bhg_params against this library."""
    lib = _ffi.load()
    assert lib.bhg_abi_check(_ffi.ABI_VERSION, 96, 0, 0, 0) == _ffi.E_INVALID
    msg = lib.bhg_last_error().decode()
    assert "bhg_params" in msg and "96" in msg and str(C.sizeof(_ffi.Params)) in msg
    # (ABI 8 only added an entry point: a binding written for 7 is still served; 6 and anything newer than the library are not)
    assert lib.bhg_abi_check(_ffi.ABI_COMPAT_MIN, 0, 0, 0, 0) == _ffi.OK
    for bad in (_ffi.ABI_COMPAT_MIN - 1, _ffi.ABI_VERSION + 1):
        assert lib.bhg_abi_check(bad, 0, 0, 0, 0) == _ffi.E_INVALID
        assert f"ABI {bad}" in lib.bhg_last_error().decode() and f"ABI {_ffi.ABI_VERSION}" in lib.bhg_last_error().decode()
    assert lib.bhg_abi_check(_ffi.ABI_VERSION, 0, C.sizeof(_ffi.Camera) + 8, 0, 0) == _ffi.E_INVALID
    assert "bhg_camera" in lib.bhg_last_error().decode()
    assert lib.bhg_abi_check(_ffi.ABI_VERSION, 0, 0, 0, 0) == _ffi.OK
    # the sized default writer refuses a struct of another size and writes nothing

    class Old(C.Structure):
        _fields_ = _ffi.Params._fields_[:-2]
    assert C.sizeof(Old) == 96
    buf = (C.c_uint8 * 160)()
    C.memset(buf, 0x5A, 160)
    lib.bhg_default_params_sized.argtypes = [C.c_void_p, C.c_size_t]
    try:
        assert lib.bhg_default_params_sized(C.addressof(buf), C.sizeof(Old)) == _ffi.E_INVALID
    finally:
        lib.bhg_default_params_sized.argtypes = [C.POINTER(_ffi.Params), C.c_size_t]
    assert bytes(buf) == b"\x5a" * 160 and "96" in lib.bhg_last_error().decode()


@pytest.mark.gpu
def test_stub_trace_and_frame_run_on_the_gpu_and_agree_with_the_package():
    """The whole text of both stubs, run as printed, against `_ffi` on the smoke rays (a 64x64x1 frame, config 1)."""
    from blackhole_geodesic_calculator_amd import camera_directions
    from blackhole_geodesic_calculator_amd.device_frame import synthetic_sky
    from blackhole_geodesic_calculator_amd.raygen import python_random_stream
    trace_stub, frame_stub = _blocks()
    ns = {"mass": 0.5, "curve_end": 50.0, "max_step": np.inf}
    exec(trace_stub, ns)
    cam = np.array([1e-4, 0.0, 30.0])
    k0 = camera_directions(64, 64, 1, 0.6, 0.6, 42.0).reshape(-1, 3)
    end_loc, end_dir, hit = ns["trace"](k0, cam)
    ctx = _ffi.Context(0)
    end, flags, _, _ = ctx.trace(k0, cam, _ffi.make_params(r_s=1.0, lambda_end=50.0))
    assert np.array_equal(end_loc, end[:, 0:3]) and np.array_equal(end_dir, end[:, 3:6])
    assert np.array_equal(hit, (flags & 1).astype(bool)) and 100 < hit.sum() < 1000

    W, H, S = 96, 64, 2
    sky = synthetic_sky(128, 64)
    jit = python_random_stream(42.0, 2 * W * H * S)
    ns.update(W=W, H=H, S=S, sky=np.ascontiguousarray(sky, np.float32), jitter=np.ascontiguousarray(jit, np.float64),
              frame_devices=(0, 0))
    exec(frame_stub, ns)
    for origin, img in (((1e-4, 0.0, 30.0), ns["first"]), ((0.5, -1.0, 28.0), ns["rgba"])):
        fr = _ffi.Frame([0], W, H, S, fov_x=0.6, fov_y=0.6, origin=origin, jitter=jit)
        fr.set_scene(sky)
        want = fr.render(_ffi.make_params(r_s=1.0, lambda_end=50.0))
        fr.close()
        assert np.array_equal(img, want)
    assert not np.array_equal(ns["first"], ns["rgba"])
