"""The threading contract of include/bhgeo.h ("one bhg_context per device; calls on one context must not overlap" --
so calls on DIFFERENT contexts may): several host threads, each with a context of its own on the one GPU, run the
host-buffer entry points at the same time (ctypes drops the GIL for the duration of a call) and every thread gets,
bit for bit, what the calls return when they run one after another.  The library-wide pieces the threads share are the
host-copy worker pool and the thread-local error message (bhgeo_capi.hip)."""
import threading

import numpy as np
import pytest

from conftest import CAM, frame_rays

pytestmark = pytest.mark.gpu

N_THREADS = 4
ROUNDS = 6


def _work(ffi, c, seed):
    """One thread's calls: a pageable-buffer trace big enough for the staging ring and the copy pool (600 k rays), a
    small trace, a sampled trajectory and a call that must fail.  Returns everything as arrays."""
    p = ffi.make_params(r_s=1.0, lambda_end=50.0)
    k_big = frame_rays(600_000 + 1000 * seed, seed=100 + seed)
    k_small = frame_rays(777 + seed, seed=200 + seed)
    out = []
    out += [np.array(a) for a in c.trace(k_big, CAM, p, pinned_results=False)]      # end, flags, n_steps, n_accepted
    out += [np.array(a) for a in c.trace(k_small, CAM, p)]
    out += [np.array(a) for a in c.trajectory(k_small[:5], CAM, p, 300)]               # traj, n_valid, end, flags
    bad = ffi.make_params(r_s=1.0, lambda_end=50.0, rtol=-1.0 - seed)
    with pytest.raises(ffi.BhgError) as e:
        c.trace(k_small, CAM, bad)
    out.append(np.frombuffer(str(e.value).encode(), np.uint8).copy())
    return out


def test_contexts_on_separate_threads_run_concurrently_and_agree_with_serial_calls():
    from blackhole_geodesic_calculator_amd import _ffi as ffi
    ctxs = [ffi.Context(0) for _ in range(N_THREADS)]
    try:
        serial = [_work(ffi, ctxs[i], i) for i in range(N_THREADS)]
        assert all(len(s[-1]) > 0 for s in serial)
        results, errors = [None] * N_THREADS, []
        start = threading.Barrier(N_THREADS)

        def run(i):
            try:
                start.wait(timeout=60)
                for _ in range(ROUNDS):
                    results[i] = _work(ffi, ctxs[i], i)
            except BaseException as ex:     # noqa: BLE001 -- reported by the main thread
                errors.append((i, repr(ex)))

        th = [threading.Thread(target=run, args=(i,)) for i in range(N_THREADS)]
        for t in th:
            t.start()
        for t in th:
            t.join(timeout=600)
        assert not any(t.is_alive() for t in th), "a thread is still inside the library after 10 minutes"
        assert not errors, errors
        for i in range(N_THREADS):
            assert len(results[i]) == len(serial[i])
            for a, b in zip(results[i], serial[i]):
                assert a.shape == b.shape and a.dtype == b.dtype
                assert np.array_equal(a, b, equal_nan=a.dtype.kind == "f"), f"thread {i}: concurrent call differs from the serial one"
    finally:
        for c in ctxs:
            c.close()


def test_last_error_is_per_thread():
    """bhg_last_error() is thread-local (include/bhgeo.h): a failure on one thread leaves the other thread's message alone."""
    from blackhole_geodesic_calculator_amd import _ffi as ffi
    lib = ffi.load()
    c = ffi.Context(0)
    try:
        k = frame_rays(10, seed=1)
        with pytest.raises(ffi.BhgError):
            c.trace(k, CAM, ffi.make_params(r_s=-1.0))
        mine = lib.bhg_last_error()
        assert mine
        seen = {}

        def other():
            seen["before"] = lib.bhg_last_error()
            c2 = ffi.Context(0)
            try:
                with pytest.raises(ffi.BhgError):
                    c2.trace(k, CAM, ffi.make_params(r_s=1.0, lambda_end=-3.0))
                seen["after"] = lib.bhg_last_error()
            finally:
                c2.close()

        t = threading.Thread(target=other)
        t.start()
        t.join(timeout=120)
        assert seen["before"] in (b"", None) or seen["before"] != mine
        assert seen["after"] and seen["after"] != mine
        assert lib.bhg_last_error() == mine
    finally:
        c.close()
