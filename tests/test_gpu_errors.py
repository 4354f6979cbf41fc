"""Failure paths on the device: a refused allocation comes back as BHG_E_NOMEM with a message and leaves the context
usable, and an error ANOTHER caller of the HIP runtime left behind on this thread (hipGetLastError is per thread and
sticky) is not reported as the status of the library's next launch (csrc/geodesic_kernels.h, BHG_LAUNCH)."""
import ctypes as C

import numpy as np
import pytest

from conftest import CAM, frame_rays

pytestmark = pytest.mark.gpu

HIP_SUCCESS, HIP_ERROR_INVALID_VALUE, HIP_ERROR_OUT_OF_MEMORY = 0, 1, 2


def _hip():
    h = C.CDLL("libamdhip64.so")
    h.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    h.hipFree.argtypes = [C.c_void_p]
    h.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    h.hipSetDevice.argtypes = [C.c_int]
    return h


def test_a_foreign_sticky_error_is_not_reported_by_the_next_launch(ctx, oracle):
    from blackhole_geodesic_calculator_amd import _ffi as ffi
    h = _hip()
    k = frame_rays(3000, seed=5)
    p = ffi.make_params(r_s=1.0, lambda_end=50.0)
    ref = ctx.trace(k, CAM, p)
    assert h.hipSetDevice(0) == HIP_SUCCESS
    for make_error in (lambda: h.hipFree(C.c_void_p(0x1234)),                         # invalid value
                       lambda: h.hipMalloc(C.byref(C.c_void_p()), C.c_size_t(1 << 50))):  # out of memory
        assert make_error() != HIP_SUCCESS           # ... and nobody reads hipGetLastError() after it
        got = ctx.trace(k, CAM, p)                   # launches on this thread: must not inherit that error
        for a, b in zip(got, ref):
            assert np.array_equal(a, b, equal_nan=True)
        assert make_error() != HIP_SUCCESS
        traj = ctx.trajectory(k[:2], CAM, p, 100)    # another launcher
        assert np.isfinite(traj[0][:, :, 0]).all()
        assert make_error() != HIP_SUCCESS
        acc = ctx.acceleration(np.array([[3.0, 1.0, 2.0]]), np.array([[0.1, 0.9, 0.2]]), p)
        assert np.isfinite(acc).all()
    o = oracle.trace(k, CAM, r_s=1.0, lambda_end=50.0)
    assert np.array_equal(ref[1], o["flags"]) and np.array_equal(ref[2], o["n_attempted"])


def test_a_refused_allocation_is_nomem_and_the_context_carries_on():
    from blackhole_geodesic_calculator_amd import _ffi as ffi
    h = _hip()
    c = ffi.Context(0)
    hog = C.c_void_p()
    try:
        k = frame_rays(20_000, seed=6)
        p = ffi.make_params(r_s=1.0, lambda_end=50.0)
        ref = c.trace(k, CAM, p)
        free, total = C.c_size_t(), C.c_size_t()
        assert h.hipSetDevice(0) == HIP_SUCCESS and h.hipMemGetInfo(C.byref(free), C.byref(total)) == HIP_SUCCESS
        assert free.value > (8 << 30)
        # take all but 2 GB of the device (an allocation nobody touches), then ask the library for 26 GB of rays
        assert h.hipMalloc(C.byref(hog), C.c_size_t(free.value - (2 << 30))) == HIP_SUCCESS
        with pytest.raises(ffi.BhgError) as e:
            ffi.RaySet(c, 8192, 8192, 16, 0.6, 0.6, CAM)
        assert e.value.code == ffi.E_NOMEM and "memory" in str(e.value).lower(), str(e.value)
        # sampled curves that need 9.6 GB of device memory for the samples: refused the same way (the context's own block is
        # released before the larger one is asked for -- and is simply absent afterwards)
        with pytest.raises(ffi.BhgError) as e2:
            c.trajectory(k[:2000], CAM, p, 100_000)
        assert e2.value.code == ffi.E_NOMEM, str(e2.value)
        # nothing is broken: the same small call still gives the same bits, with the device still full ...
        got = c.trace(k, CAM, p)
        for a, b in zip(got, ref):
            assert np.array_equal(a, b, equal_nan=True)
        assert h.hipFree(hog) == HIP_SUCCESS
        hog = C.c_void_p()
        # ... and with the memory back the ray set is made
        rs = ffi.RaySet(c, 512, 512, 2, 0.6, 0.6, CAM)
        assert rs.n == 512 * 512 * 2
        rs.close()
    finally:
        if hog.value:
            h.hipFree(hog)
        c.close()


def test_device_buffers_need_only_their_element_alignment(ctx):
    """The device-buffer calls take whatever address a caller's array view has: buffers that start one ELEMENT into an
    allocation (8 bytes for the doubles, 4 for the counts, 1 for the flags -- not the 16 bytes of the kernels' widest
    stores) give the same bits as aligned ones, for the full records, the directions-only form and the Kerr kernel."""
    import torch
    from blackhole_geodesic_calculator_amd import _ffi as ffi
    n = 70_001
    k = frame_rays(n, seed=8)
    st = torch.cuda.current_stream().cuda_stream
    for kw in (dict(r_s=1.0, lambda_end=50.0), dict(r_s=1.0, lambda_end=50.0, rhs_form=2, spin=0.45),
               dict(r_s=1.0, lambda_end=80.0, r_exit=40.0, disk_r_in=4.5, disk_r_out=10.5)):
        p = ffi.make_params(**kw)
        ref = ctx.trace(k, CAM, p)
        dk = torch.zeros(n * 3 + 1, dtype=torch.float64, device="cuda")
        dk[1:] = torch.from_numpy(k.reshape(-1)).cuda()
        dend = torch.full((n * 6 + 1,), -7.0, dtype=torch.float64, device="cuda")
        ddir = torch.full((n * 3 + 1,), -7.0, dtype=torch.float64, device="cuda")
        dfl = torch.full((n + 1,), 255, dtype=torch.uint8, device="cuda")
        dst = torch.full((n + 1,), -1, dtype=torch.int32, device="cuda")
        dac = torch.full((n + 1,), -1, dtype=torch.int32, device="cuda")
        ctx.trace_device(p, n, dk.data_ptr() + 8, dend.data_ptr() + 8, x0_shared=CAM, d_flags=dfl.data_ptr() + 1,
                         d_n_steps=dst.data_ptr() + 4, d_n_accepted=dac.data_ptr() + 4, stream=st)
        torch.cuda.synchronize()
        assert np.array_equal(dend[1:].cpu().numpy().reshape(n, 6), ref[0], equal_nan=True)
        assert np.array_equal(dfl[1:].cpu().numpy(), ref[1])
        assert np.array_equal(dst[1:].cpu().numpy().astype(np.uint32), ref[2])
        assert np.array_equal(dac[1:].cpu().numpy().astype(np.uint32), ref[3])
        # (the element in front of each buffer is untouched)
        assert dend[0].item() == -7.0 and dfl[0].item() == 255 and dst[0].item() == -1 and dac[0].item() == -1
        if "disk_r_out" not in kw:
            ctx.trace_dir_device(p, n, dk.data_ptr() + 8, ddir.data_ptr() + 8, x0_shared=CAM, d_flags=dfl.data_ptr() + 1, stream=st)
            torch.cuda.synchronize()
            assert np.array_equal(ddir[1:].cpu().numpy().reshape(n, 3), ref[0][:, 3:6], equal_nan=True) and ddir[0].item() == -7.0
