"""Create / use / destroy, many times: every object of include/bhgeo.h gives back what it took -- device memory
(hipMemGetInfo through torch.cuda.mem_get_info) and host memory (page-locked blocks, worker threads, staging rings: the
process's resident set) return to where they were after the first cycle.  A render loop that builds a context per frame
(the reference builds its solver object per frame, RelativisticRenderEngine.py:134) must be able to run for days."""
import gc
import os

import numpy as np
import pytest

from conftest import CAM, frame_rays

pytestmark = pytest.mark.gpu

CYCLES = 25


def _cycle(ffi, sky, seed):
    c = ffi.Context(0)
    try:
        p = ffi.make_params(r_s=1.0, lambda_end=50.0)
        k = frame_rays(300_000, seed=seed)
        c.trace(k, CAM, p)                               # page-locked result arrays from the context's pool
        c.trace(k, CAM, p, pinned_results=False)         # the staging ring + the copy threads
        c.trace(k[:5000], CAM, ffi.make_params(r_s=1.0, lambda_end=80.0, r_exit=40.0, disk_r_in=4.5, disk_r_out=10.5))
        c.trace(k[:5000], CAM, ffi.make_params(r_s=1.0, lambda_end=50.0, rhs_form=2, spin=0.45))
        c.trajectory(k[:3], CAM, p, 2000)
        rs = ffi.RaySet(c, 128, 96, 2, 0.6, 0.6, CAM)
        rs.trace(p, want=("end_dir", "flags"))
        rs.close()
        f = ffi.Frame([0, 0], 160, 128, 3, fov_x=0.6, fov_y=0.6, origin=CAM)
        f.set_scene(sky)
        f.render(p)
        f.close()
        with pytest.raises(ffi.BhgError):                # an error path leaves nothing behind either
            c.trace(k[:10], CAM, ffi.make_params(r_s=-1.0))
    finally:
        c.close()
    gc.collect()


def test_objects_give_back_device_and_host_memory():
    import psutil
    import torch
    from blackhole_geodesic_calculator_amd import _ffi as ffi
    from blackhole_geodesic_calculator_amd.device_frame import synthetic_sky
    sky = synthetic_sky(256, 128)
    proc = psutil.Process(os.getpid())
    for i in range(3):                                   # first cycles: code objects, the runtime's own pools, allocator arenas
        _cycle(ffi, sky, i)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info(0)[0]
    rss0 = proc.memory_info().rss
    thr0 = proc.num_threads()
    for i in range(CYCLES):
        _cycle(ffi, sky, 10 + i)
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info(0)[0]
    rss1 = proc.memory_info().rss
    thr1 = proc.num_threads()
    print(f"{CYCLES} cycles: device memory in use {(free0 - free1) / 2**20:+.1f} MiB, resident set {(rss1 - rss0) / 2**20:+.1f} MiB, "
          f"threads {thr0} -> {thr1}")
    # one cycle holds ~60 MB of device memory and ~50 MB of page-locked host memory while it runs
    assert free0 - free1 < 32 << 20, f"device memory: {(free0 - free1) / 2**20:.1f} MiB more in use after {CYCLES} cycles"
    assert rss1 - rss0 < 96 << 20, f"resident set grew by {(rss1 - rss0) / 2**20:.1f} MiB over {CYCLES} cycles"
    assert thr1 <= thr0 + 2, f"threads {thr0} -> {thr1}: a context's copy threads outlive it"
