"""GPU tests of the device-resident frame pipeline (ray generation, shading, multisample mean) and
of the batched adaptors, through the C ABI."""
import numpy as np
import pytest

from conftest import CAM

pytestmark = pytest.mark.gpu


def _params(**kw):
    from blackhole_geodesic_calculator_amd import _ffi
    return _ffi.make_params(**kw)


@pytest.mark.parametrize("euler", [(0.0, 0.0, 0.0), (0.3, -0.2, 1.1)])
def test_device_raygen_matches_host(ctx, euler):
    import torch
    from blackhole_geodesic_calculator_amd import camera_directions
    from blackhole_geodesic_calculator_amd.device_frame import DeviceFrame
    W, H, S = 48, 40, 3
    want = camera_directions(W, H, S, 0.6, 0.45, 42.0, rotation_euler=euler).reshape(S, H * W, 3)
    fr = DeviceFrame(ctx, W, H, S, fov_x=0.6, fov_y=0.45, sampling_seed=42.0, rotation_euler=euler)
    fr.generate_rays()
    torch.cuda.synchronize()
    got = fr.d_k0.cpu().numpy().reshape(S, H * W, 3)
    if euler == (0.0, 0.0, 0.0):
        assert np.array_equal(got, want)  # bit-identical to the reference's loop
    else:
        assert np.abs(got - want).max() < 5e-16
    # a tile shard: arbitrary pixel subset
    pix = np.random.default_rng(0).permutation(W * H)[:777]
    fr2 = DeviceFrame(ctx, W, H, S, fov_x=0.6, fov_y=0.45, sampling_seed=42.0, rotation_euler=euler, pixels=pix)
    fr2.generate_rays()
    torch.cuda.synchronize()
    got2 = fr2.d_k0.cpu().numpy().reshape(S, len(pix), 3)
    assert np.array_equal(got2, got[:, pix, :])


def test_device_shade_and_mean_match_numpy(ctx, oracle):
    import torch
    from blackhole_geodesic_calculator_amd.device_frame import DeviceFrame, synthetic_sky
    from oracle import shade_reference as sh
    W, H, S = 96, 80, 4
    sky = synthetic_sky(512, 256)
    fr = DeviceFrame(ctx, W, H, S, fov_x=0.6, fov_y=0.6, sampling_seed=42.0, origin=CAM)
    fr.set_sky(sky)
    rgba = fr.render(_params(r_s=1.0, lambda_end=50.0)).cpu().numpy()
    torch.cuda.synchronize()
    end, flags = fr.d_end.cpu().numpy(), fr.d_flags.cpu().numpy()
    want = sh.shade_reduce(end, flags, W * H, S, sky)
    assert np.abs(rgba - want).max() < 1e-12
    assert np.all(rgba[:, 3] == 1.0)
    # and the whole pipeline against the oracle + numpy shade on the same rays
    k0 = fr.d_k0.cpu().numpy()
    o = oracle.trace(k0, CAM, r_s=1.0, lambda_end=50.0)
    assert np.array_equal(flags, o["flags"])
    want2 = sh.shade_reduce(o["end"], o["flags"], W * H, S, sky)
    assert np.abs(rgba - want2).max() < 1e-6  # bilinear sky gradient x end-direction tolerance
    # pixels whose every sample ended on the horizon are exactly black (:242-244)
    all_hit = ((flags.reshape(S, H * W) & 1) != 0).all(0)
    assert all_hit.sum() > 50
    assert np.array_equal((rgba[:, :3] == 0.0).all(1), all_hit)


@pytest.mark.parametrize("W,H,S", [(37, 23, 1), (50, 31, 7), (9, 5, 300), (64, 3, 256), (3, 64, 257)])
def test_shade_and_mean_on_odd_frame_shapes_and_sample_counts(ctx, W, H, S):
    """Frame shapes that are multiples of nothing, one sample, a sample count that does not divide the shade kernel's 256-thread
    block, exactly 256 samples per pixel, and MORE than 256 (the kernel's serial fallback): rays, trace, shade + sample mean
    against the numpy restatement, in all three output forms (fp64, float32, float32 scattered by pixel id) and from
    directions alone."""
    import torch
    from blackhole_geodesic_calculator_amd.device_frame import DeviceFrame, synthetic_sky
    from oracle import shade_reference as sh
    sky = synthetic_sky(128, 64)
    p = _params(r_s=1.0, lambda_end=50.0)
    fr = DeviceFrame(ctx, W, H, S, fov_x=0.6, fov_y=0.6, sampling_seed=42.0, origin=CAM)
    fr.set_sky(sky)
    rgba = fr.render(p).cpu().numpy()
    end, flags = fr.d_end.cpu().numpy(), fr.d_flags.cpu().numpy()
    want = sh.shade_reduce(end, flags, W * H, S, sky)
    assert rgba.shape == (W * H, 4) and np.abs(rgba - want).max() < 1e-12
    f32 = torch.zeros((W * H, 4), dtype=torch.float32, device="cuda")
    fr.shade_f32(f32)
    assert np.abs(f32.cpu().numpy() - want).max() < 1e-6
    perm = torch.randperm(W * H, device="cuda")
    sc = torch.zeros((W * H, 4), dtype=torch.float32, device="cuda")
    fr.shade_f32(sc, perm)
    torch.cuda.synchronize()
    assert torch.equal(sc[perm], f32)
    fd = DeviceFrame(ctx, W, H, S, fov_x=0.6, fov_y=0.6, sampling_seed=42.0, origin=CAM, directions_only=True)
    fd.set_sky(sky)
    assert np.array_equal(fd.render(p).cpu().numpy(), rgba)


def test_device_scene_shade_disk_and_objects(ctx, oracle):
    """Disk colour (Limited engine's checkHitDisk profile) and object Lambert shading in the device shade
    kernel against the numpy restatement, on a frame that holds horizon, sky, disk and object pixels."""
    import torch
    from blackhole_geodesic_calculator_amd.device_frame import DeviceFrame, synthetic_sky
    from oracle import shade_reference as sh
    W, H, S = 128, 96, 3
    sky = synthetic_sky(512, 256)
    disk_tex = synthetic_sky(256, 64, seed=3)
    inc = np.radians(75.0)
    cam = np.array([30 * np.sin(inc), 0.0, 30 * np.cos(inc)])
    fr = DeviceFrame(ctx, W, H, S, fov_x=0.9, fov_y=0.9, sampling_seed=42.0, origin=cam, rotation_euler=(0.0, inc, 0.0))
    fr.set_sky(sky)
    prof = dict(disk_phase=0.4, disk_mean=0.3, disk_stddev=0.25, disk_intensity=2.0)
    fr.set_disk(3.0, 9.0, disk_tex, **prof)
    sph = [[6.0, 3.0, 2.5, 1.5], [7.0, -4.0, 3.0, 1.0], [2.0, 6.0, -1.0, 1.2]]
    rgb = [[1.0, 0.8, 0.6], [0.2, 0.9, 0.3], [0.5, 0.5, 1.0]]
    lamps = [[20.0, 0.0, 20.0, 10.0], [10.0, -15.0, 5.0, 6.0]]
    fr.set_objects(sph, rgb, lamps)
    kw = dict(r_s=1.0, lambda_end=80.0, r_exit=40.0, disk_r_in=3.0, disk_r_out=9.0)
    rgba = fr.render(_params(**kw)).cpu().numpy()
    torch.cuda.synchronize()
    end, flags, obj = fr.d_end.cpu().numpy(), fr.d_flags.cpu().numpy(), fr.d_obj.cpu().numpy()
    kinds = {int(f): int((flags == f).sum()) for f in np.unique(flags)}
    assert kinds.get(1, 0) > 100 and kinds.get(8, 0) > 1000 and kinds.get(128, 0) > 1000 and kinds.get(0x88, 0) > 300, kinds
    want = sh.shade_scene(end, flags, obj, W * H, S, sky, disk=(3.0, 9.0), disk_tex=disk_tex, disk_profile=dict(
        phase=0.4, mean=0.3, stddev=0.25, intensity=2.0), spheres=sph, sphere_rgb=np.array(rgb), lamps=lamps)
    assert np.abs(rgba - want).max() < 1e-11
    # trace parity of the same frame (ids included), and the whole pipeline against oracle + numpy shade
    k0 = fr.d_k0.cpu().numpy()
    o = oracle.trace(k0, cam, spheres=sph, **kw)
    assert np.array_equal(flags, o["flags"]) and np.array_equal(obj, o["object_id"])
    want2 = sh.shade_scene(o["end"], o["flags"], o["object_id"], W * H, S, sky, disk=(3.0, 9.0), disk_tex=disk_tex,
                           disk_profile=dict(phase=0.4, mean=0.3, stddev=0.25, intensity=2.0), spheres=sph,
                           sphere_rgb=np.array(rgb), lamps=lamps)
    assert np.abs(rgba - want2).max() < 1e-5
    # some object pixels are lit, some lie in shadow or face away
    lit = want[:, :3].sum(1)
    assert (lit > 0).any()


def test_device_buffers_need_only_8_byte_alignment(ctx):
    """k0 / end handed over at an address that is 8 but not 16 bytes aligned (a view into a larger
    allocation): same results as the aligned call (the kernels use 16-byte accesses on 48-byte records)."""
    import torch
    from blackhole_geodesic_calculator_amd.device_frame import synthetic_sky
    from conftest import frame_rays
    n, S = 5000, 2
    k = torch.as_tensor(frame_rays(n, seed=5)).cuda()
    p = _params(r_s=1.0, lambda_end=50.0)
    outs = []
    for off in (0, 1):
        kb = torch.empty(n * 3 + 2, dtype=torch.float64, device="cuda")
        eb = torch.empty(n * 6 + 2, dtype=torch.float64, device="cuda")
        kv, ev = kb[off:off + n * 3], eb[off:off + n * 6]
        kv.copy_(k.reshape(-1))
        fl = torch.empty(n, dtype=torch.uint8, device="cuda")
        assert kv.data_ptr() % 16 == 8 * off and ev.data_ptr() % 16 == 8 * off
        ctx.trace_device(p, n, kv.data_ptr(), ev.data_ptr(), x0_shared=CAM, d_flags=fl.data_ptr(),
                         stream=torch.cuda.current_stream().cuda_stream)
        sky = torch.as_tensor(synthetic_sky(128, 64)).cuda()
        rgba = torch.empty((n // S, 4), dtype=torch.float64, device="cuda")
        ctx.shade_device(ev.data_ptr(), fl.data_ptr(), n // S, S, sky.data_ptr(), 128, 64, rgba.data_ptr(),
                         stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        outs.append((ev.cpu().numpy().copy(), fl.cpu().numpy().copy(), rgba.cpu().numpy().copy()))
    for a, b in zip(outs[0], outs[1]):
        assert np.array_equal(a, b, equal_nan=True)


def test_shade_f32_scatter_matches_fp64_path(ctx):
    """The float32 / scattered output of the shade kernel (what the bench's frame end uses) equals the fp64
    output cast to float32 and scattered by pixel id."""
    import torch
    from blackhole_geodesic_calculator_amd import dist as bdist
    from blackhole_geodesic_calculator_amd.device_frame import DeviceFrame, synthetic_sky
    W, H, S = 96, 64, 2
    pix = bdist.rank_pixels(W, H, 32, 0, 1)
    fr = DeviceFrame(ctx, W, H, S, fov_x=0.6, fov_y=0.6, sampling_seed=42.0, origin=CAM, pixels=pix)
    fr.set_sky(synthetic_sky(256, 128))
    rgba = fr.render(_params(r_s=1.0, lambda_end=50.0)).clone()
    d_pix = torch.as_tensor(pix).cuda()
    frame = torch.zeros((H * W, 4), dtype=torch.float32, device="cuda")
    fr.shade_f32(frame, d_pix)
    slab = torch.zeros((fr.P, 4), dtype=torch.float32, device="cuda")
    fr.shade_f32(slab)
    torch.cuda.synchronize()
    want = torch.zeros_like(frame)
    want[d_pix] = rgba.to(torch.float32)
    assert torch.equal(frame, want) and torch.equal(slab, rgba.to(torch.float32))
    g = bdist.FrameGatherer(W, H, 32, channels=4, dtype=torch.float32, device="cuda")
    g.submit_with(0, fr.shade_f32)
    g.drain()
    assert torch.equal(g.image().reshape(-1, 4), want) and g.frames_done == 1


def test_work_order_hint_never_changes_results(ctx):
    """params.order_blocks only permutes the order in which 64-ray batches are started."""
    import torch
    from conftest import frame_rays
    n = 64 * 50 * 4
    k = torch.as_tensor(frame_rays(n, seed=9)).cuda()
    outs = []
    for ob in (0, 4, 7, 50):      # 7: n / 7 is not whole -> ignored; 50: blocks of 4 batches
        p = _params(r_s=1.0, lambda_end=50.0, r_exit=40.0, disk_r_in=3.0, disk_r_out=8.0, order_blocks=ob)
        end = torch.empty((n, 6), dtype=torch.float64, device="cuda")
        fl = torch.empty(n, dtype=torch.uint8, device="cuda")
        st = torch.empty(n, dtype=torch.int32, device="cuda")
        ctx.trace_device(p, n, k.data_ptr(), end.data_ptr(), x0_shared=CAM, d_flags=fl.data_ptr(), d_n_steps=st.data_ptr(),
                         stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        outs.append((end.cpu().numpy(), fl.cpu().numpy(), st.cpu().numpy()))
    for o in outs[1:]:
        for a, b in zip(outs[0], o):
            assert np.array_equal(a, b, equal_nan=True)


def test_profiled_passes_and_pixel_cost(ctx):
    """bhg_set_profiling / bhg_last_pass_ms: {prepare, trace, post} -- the start records are worked out inside the trace kernel
    (prepare reads ~0 for every form), Kerr has the finalize pass after it; DeviceFrame.pixel_cost() = the steps of a pixel's rays summed over its samples."""
    import torch
    from blackhole_geodesic_calculator_amd.device_frame import DeviceFrame
    fr = DeviceFrame(ctx, 64, 48, 3, fov_x=0.6, fov_y=0.6)
    fr.generate_rays()
    ctx.set_profiling(True)
    try:
        fr.trace(_params(r_s=1.0, lambda_end=50.0))
        t = ctx.last_pass_ms()
        assert t["trace"] > 0.0 and t["prepare"] >= 0.0 and t["post"] == 0.0 and t["prepare"] < t["trace"]
        cost = fr.pixel_cost().cpu().numpy()
        st = fr.d_steps.cpu().numpy().astype(np.int64)
        assert cost.shape == (64 * 48,) and np.array_equal(cost, st.reshape(3, -1).sum(0)) and cost.min() > 0
        fr.trace(_params(r_s=1.0, lambda_end=50.0, rhs_form=2, spin=0.45))
        t = ctx.last_pass_ms()
        assert t["trace"] > 0.0 and 0.0 <= t["prepare"] < 0.1 * t["trace"] and t["post"] > 0.0
    finally:
        ctx.set_profiling(False)
    torch.cuda.synchronize()


def test_frame_batch_equals_single_frames(ctx):
    """Several cameras in ONE trace call (per-ray origins) give bit-identical rays, end states and pixels."""
    import torch
    from blackhole_geodesic_calculator_amd.device_frame import DeviceFrame, FrameBatch, synthetic_sky
    W, H, S = 64, 48, 2
    sky = synthetic_sky(256, 128)
    cams = [dict(origin=(30 * np.sin(i), 0.0, 30 * np.cos(i)), rotation_euler=(0.0, i, 0.0)) for i in (1.4, 0.6, 0.1)]
    p = _params(r_s=1.0, lambda_end=80.0, r_exit=40.0, disk_r_in=4.5, disk_r_out=10.5)
    fb = FrameBatch(ctx, cams, W, H, S, fov_x=0.9, fov_y=0.9, sampling_seed=42.0)
    for f in fb.frames:
        f.set_sky(sky)
        f.set_disk(4.5, 10.5)
    fb.generate_rays()
    fb.trace(p)
    imgs = [x.cpu().numpy().copy() for x in fb.shade()]
    torch.cuda.synchronize()
    for j, cam in enumerate(cams):
        one = DeviceFrame(ctx, W, H, S, fov_x=0.9, fov_y=0.9, sampling_seed=42.0, **cam)
        one.set_sky(sky)
        one.set_disk(4.5, 10.5)
        img = one.render(p).cpu().numpy()
        f = fb.frames[j]
        assert np.array_equal(one.d_k0.cpu().numpy(), f.d_k0.cpu().numpy())
        assert np.array_equal(one.d_flags.cpu().numpy(), f.d_flags.cpu().numpy())
        assert np.array_equal(one.d_end.cpu().numpy(), f.d_end.cpu().numpy(), equal_nan=True)
        assert np.array_equal(img, imgs[j])
        assert (f.d_flags == 128).sum().item() > 50


def test_frame_tracer_on_gpu_matches_oracle(ctx, oracle):
    """frame.FrameTracer (the batched ray_trace generator) with the real integrator."""
    from blackhole_geodesic_calculator_amd import GeodesicIntegratorSchwarzschild, camera_directions
    from blackhole_geodesic_calculator_amd.frame import FrameTracer, equirect_uv
    W, H, S = 40, 32, 2
    gi = GeodesicIntegratorSchwarzschild(mass=0.5, context=ctx)

    def sky(d):
        u, v = equirect_uv(d)
        return np.stack([0.5 + 0.5 * np.sin(np.pi * u), 0.5 + 0.5 * v, 0.25 + 0.25 * np.cos(2 * np.pi * u)], -1)

    ft = FrameTracer(gi, W, H, S, fov_x=0.6, fov_y=0.6, sampling_seed=42.0, origin=CAM + np.array([1.0, 2.0, 3.0]),
                     bh_loc=np.array([1.0, 2.0, 3.0]))
    buf = np.ones((H, W, 4))
    prog = list(ft.ray_trace(buf, sky))
    assert len(prog) == S * H and prog[-1] == (S * W * H - W) / (S * W * H)
    d = camera_directions(W, H, S, 0.6, 0.6, 42.0)
    o = oracle.trace(d.reshape(-1, 3), CAM, r_s=1.0, lambda_end=50.0)
    col = sky(o["end"][:, 3:6])
    col[(o["flags"] & 1) != 0] = 0.0
    want = col.reshape(S, H, W, 3).sum(0) / S
    assert np.abs(buf[..., :3] - want).max() < 1e-7


def test_relativistic_camera_on_gpu(ctx, oracle, tmp_path):
    from blackhole_geodesic_calculator_amd import GeodesicIntegratorSchwarzschild
    from blackhole_geodesic_calculator_amd.camera import RelativisticCamera
    gi = GeodesicIntegratorSchwarzschild(mass=0.5, context=ctx)
    cam = RelativisticCamera(resolution=[48, 64], field_of_view=[0.6, 0.6], camera_location=CAM, integrator=gi)
    cam.run()
    o = oracle.trace(cam.pixel_directions().reshape(-1, 3), CAM, r_s=1.0, lambda_end=50.0)
    assert np.array_equal(cam.ray_blackhole_hit.reshape(-1), (o["flags"] & 1).astype(np.uint8))
    assert np.abs(cam.ray_end.reshape(-1, 6) - o["end"]).max() < 1e-7
    assert cam.ray_blackhole_hit.sum() > 20 and cam.ray_end[0, 0, 3:6].shape == (3,)
    cam.save(tmp_path / "c.pkl")
    assert np.array_equal(RelativisticCamera().load(tmp_path / "c.pkl").ray_end, cam.ray_end)


def test_c_example_runs_on_the_gpu(tmp_path):
    """examples/trace_frame.c: the C ABI from plain C, end to end."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "blackhole_geodesic_calculator_amd")
    exe = tmp_path / "trace_frame"
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-I", os.path.join(root, "include"),
                           os.path.join(root, "examples", "trace_frame.c"), "-L", libdir, "-lbhgeo",
                           "-Wl,-rpath," + libdir, "-lm", "-o", str(exe)])
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    head = r.stdout.splitlines()[0]
    assert "6144 rays" in head and " 0 left the region" in head   # curve_end 60 ends the far-side rays before r = 40
    assert "#" in r.stdout and "o" in r.stdout.split("\n", 1)[1]


@pytest.mark.parametrize("kw", [dict(r_s=1.0, lambda_end=50.0), dict(r_s=1.0, lambda_end=80.0, r_exit=35.0),
                                dict(r_s=1.0, lambda_end=50.0, rhs_form=2, spin=0.45),
                                dict(r_s=1.0, lambda_end=40.0, method=1, h_fixed=0.1)])
def test_direction_only_trace_and_shade_match_whole_records(ctx, kw):
    """bhg_trace_dir_device writes the direction half of the end states alone and bhg_shade_dir_device reads it: the
    directions, flags, step counts and the shaded frame (fp64 and float32) are bit-for-bit those of the whole-record
    calls -- with the exit sphere (events located in the kernel), Kerr (split off after the finalize pass) and RK4."""
    import torch
    from blackhole_geodesic_calculator_amd import _ffi
    from blackhole_geodesic_calculator_amd.device_frame import DeviceFrame, synthetic_sky
    W, H, S = 96, 64, 3
    full = DeviceFrame(ctx, W, H, S, fov_x=0.6, fov_y=0.6)
    dirs = DeviceFrame(ctx, W, H, S, fov_x=0.6, fov_y=0.6, directions_only=True)
    sky = synthetic_sky(256, 128, seed=2)
    p = _ffi.make_params(**kw)
    out = []
    for fr in (full, dirs):
        fr.set_sky(sky)
        fr.generate_rays()
        if fr.d_end is not None:
            fr.d_end.fill_(float("nan"))
        fr.trace(p)
        rgba = fr.shade().clone()
        f32 = torch.zeros((fr.P, 4), dtype=torch.float32, device="cuda")
        fr.shade_f32(f32)
        torch.cuda.synchronize()
        out.append((rgba, f32, fr.d_flags.clone(), fr.d_steps.clone(), fr.d_acc.clone()))
    assert dirs._dir_traced and not full._dir_traced
    assert dirs.d_end is None                                  # a direction-only frame allocates no record array at all
    fin = ~torch.isnan(full.d_end[:, 3:6]).any(1)
    assert torch.equal(dirs.d_dir[fin], full.d_end[:, 3:6][fin]) and torch.equal(torch.isnan(dirs.d_dir), torch.isnan(full.d_end[:, 3:6]))
    for a, b in zip(out[0], out[1]):
        assert torch.equal(a, b)
    assert int((out[0][2] & 1).sum()) > 0 and int((out[0][2] & 1).sum()) < full.n
    # a frame with a disk needs the end locations: the option falls back to whole records by itself
    dirs.set_disk(4.5, 10.5)
    with pytest.raises(RuntimeError):      # the scene changed after the last trace: both shade paths refuse, neither reads stale data
        dirs.shade()
    with pytest.raises(RuntimeError):
        dirs.shade_f32(torch.zeros((dirs.P, 4), dtype=torch.float32, device="cuda"))
    dirs.trace(_ffi.make_params(r_s=1.0, lambda_end=80.0, r_exit=35.0, disk_r_in=4.5, disk_r_out=10.5))
    assert not dirs._dir_traced and dirs.d_end is not None
    dirs.shade()


@pytest.mark.parametrize("n", [0, 1, 65, 4099])
def test_direction_only_entry_point_ragged_sizes_and_per_ray_origins(ctx, n):
    """bhg_trace_dir_device on ragged sizes with per-ray origins: the directions of bhg_trace_device, bit for bit;
    n = 0 is a no-op; a NULL direction array is refused."""
    import torch
    from blackhole_geodesic_calculator_amd import _ffi
    rng = np.random.default_rng(n)
    x0 = rng.normal(size=(max(n, 1), 3)) * 6.0 + np.array([0.0, 0.0, 25.0])
    k0 = -x0 / np.linalg.norm(x0, axis=1)[:, None] + rng.normal(size=x0.shape) * 0.15
    k0 /= np.linalg.norm(k0, axis=1)[:, None]
    d_x0, d_k0 = torch.as_tensor(x0).cuda(), torch.as_tensor(k0).cuda()
    p = _ffi.make_params(r_s=1.0, lambda_end=60.0, r_exit=45.0)
    end = torch.full((max(n, 1), 6), float("nan"), dtype=torch.float64, device="cuda")
    dirs = torch.full((max(n, 1), 3), float("nan"), dtype=torch.float64, device="cuda")
    fl = [torch.full((max(n, 1),), 255, dtype=torch.uint8, device="cuda") for _ in range(2)]
    ctx.trace_device(p, n, d_k0.data_ptr(), end.data_ptr(), d_x0=d_x0.data_ptr(), d_flags=fl[0].data_ptr())
    ctx.trace_dir_device(p, n, d_k0.data_ptr(), dirs.data_ptr(), d_x0=d_x0.data_ptr(), d_flags=fl[1].data_ptr())
    torch.cuda.synchronize()
    if n == 0:
        assert torch.isnan(dirs).all() and int(fl[1][0]) == 255
    else:
        assert torch.equal(fl[0], fl[1]) and torch.equal(dirs, end[:, 3:6])
        with pytest.raises(_ffi.BhgError):
            ctx.trace_dir_device(p, n, d_k0.data_ptr(), 0, d_x0=d_x0.data_ptr())
