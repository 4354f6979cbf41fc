"""The library-owned frame (bhg_frame_*, include/bhgeo.h) on the GPU: one process, one or several device contexts, no
PyTorch in the product path.  What it replaces: the body of the reference's frame loop as Blender calls it,
raytracer/RelativisticRenderEngine.py:50 -> :152-168 -> :172-267.

Parity: bit-identical to device_frame.DeviceFrame (which the rest of the suite pins against the oracle and the numpy
shade restatement) on the same frame -- sky only, disk, objects; a frame sharded over {0, 0} / {0, 0, 0} (several
contexts of ONE GPU: the only way to run N > 1 on this box) bit-identical to the one-device frame, cyclic dealing and
dealing by measured cost alike.  torch is used HERE only to drive the DeviceFrame the frame is compared with.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import CAM

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _params(**kw):
    from blackhole_geodesic_calculator_amd import _ffi
    return _ffi.make_params(**kw)


def _device_frame_image(ctx, W, H, S, params, sky, cam=CAM, euler=(0.0, 0.0, 0.0), fov=0.6, disk=None, disk_tex=None, profile=None,
                        objects=None):
    import torch
    from blackhole_geodesic_calculator_amd.device_frame import DeviceFrame
    fr = DeviceFrame(ctx, W, H, S, fov_x=fov, fov_y=fov, sampling_seed=42.0, origin=cam, rotation_euler=euler,
                     directions_only=(disk is None and objects is None))
    fr.set_sky(sky)
    if disk is not None:
        fr.set_disk(disk[0], disk[1], disk_tex, **(profile or {}))
    if objects is not None:
        fr.set_objects(*objects)
    fr.generate_rays()
    fr.trace(params)
    out = torch.empty((W * H, 4), dtype=torch.float32, device=fr.dev)
    fr.shade_f32(out)
    torch.cuda.synchronize()
    return out.cpu().numpy().reshape(H, W, 4), int(fr.d_steps.to(torch.int64).sum().item())


def _frame(devices, W, H, S, cam=CAM, euler=(0.0, 0.0, 0.0), fov=0.6, **kw):
    from blackhole_geodesic_calculator_amd import _ffi
    from blackhole_geodesic_calculator_amd.raygen import euler_xyz_matrix, python_random_stream
    return _ffi.Frame(devices, W, H, S, fov_x=fov, fov_y=fov, origin=cam, rot=euler_xyz_matrix(euler),
                      jitter=python_random_stream(42.0, 2 * S * W * H), **kw)


@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0]])
def test_sky_frame_is_bit_identical_to_device_frame(ctx, devices):
    from blackhole_geodesic_calculator_amd.device_frame import synthetic_sky
    W, H, S = 160, 96, 3          # (tiles of 32: a ragged right / top edge)
    sky = synthetic_sky(256, 128)
    p = _params(r_s=1.0, lambda_end=50.0)
    want, steps = _device_frame_image(ctx, W, H, S, p, sky)
    fr = _frame(devices, W, H, S)
    fr.set_scene(sky)
    got = fr.render(p)
    assert np.array_equal(got, want)
    st, info = fr.stats(), fr.info()
    assert st["rays"] == W * H * S and st["attempted_steps"] == steps and st["horizon_rays"] > 100
    assert info["n_devices"] == len(devices) and info["gather"] == "copy" and info["directions_only"]
    # re-dealt by the measured cost of that render: another pixel order on every device, the same image
    fr.rebalance()
    assert fr.info()["dealt_by_measured_cost"]
    assert np.array_equal(fr.render(p), want)
    if len(devices) > 1:        # the first device dealt a smaller part (it also assembles the frame): the same image
        eq = fr.info()["largest_shard_pixels"]
        fr.rebalance(root_share=0.6)
        assert np.array_equal(fr.render(p), want) and fr.info()["largest_shard_pixels"] > eq
        fr.rebalance()
    # an enqueue-only render leaves the image on the first device
    fr.render(p, to_host=False)
    fr.synchronize()
    assert fr.device_image() != 0 and fr.info()["renders"] == (4 if len(devices) > 1 else 3)
    # page-locked destination: written by the copy engine directly
    pinned = ctx.pinned.empty((H, W, 4), np.float32)
    assert np.array_equal(fr.render(p, out=pinned), want)
    fr.close()


@pytest.mark.parametrize("devices", [[0], [0, 0]])
def test_disk_and_object_frames_are_bit_identical_to_device_frame(ctx, devices):
    from blackhole_geodesic_calculator_amd.device_frame import synthetic_sky
    W, H, S = 128, 96, 2
    sky, disk_tex = synthetic_sky(256, 128), synthetic_sky(128, 32, seed=3)
    inc = np.radians(75.0)
    cam = np.array([30 * np.sin(inc), 0.0, 30 * np.cos(inc)])
    euler = (0.0, inc, 0.0)
    prof = dict(disk_phase=0.4, disk_mean=0.3, disk_stddev=0.25, disk_intensity=2.0)
    sph = [[6.0, 3.0, 2.5, 1.5], [7.0, -4.0, 3.0, 1.0]]
    rgb = [[1.0, 0.8, 0.6], [0.2, 0.9, 0.3]]
    lamps = [[20.0, 0.0, 20.0, 10.0], [10.0, -15.0, 5.0, 6.0]]
    fr = _frame(devices, W, H, S, cam=cam, euler=euler, fov=0.9)
    # disk only
    p = _params(r_s=1.0, lambda_end=80.0, r_exit=40.0, disk_r_in=3.0, disk_r_out=9.0)
    want, _ = _device_frame_image(ctx, W, H, S, p, sky, cam, euler, 0.9, disk=(3.0, 9.0), disk_tex=disk_tex, profile=prof)
    fr.set_scene(sky, disk=(3.0, 9.0), disk_tex=disk_tex, **prof)
    got = fr.render(p)
    assert np.array_equal(got, want) and not fr.info()["directions_only"]
    # disk + objects (the scene is replaced; the images are kept)
    want2, _ = _device_frame_image(ctx, W, H, S, p, sky, cam, euler, 0.9, disk=(3.0, 9.0), disk_tex=disk_tex, profile=prof,
                                   objects=(sph, rgb, lamps))
    fr.set_scene(None, disk=(3.0, 9.0), spheres=sph, sphere_rgb=rgb, lamps=lamps, **prof)
    got2 = fr.render(p)
    assert np.array_equal(got2, want2) and not np.array_equal(got2, got)
    # objects only, Kerr sky: the other trace variants behind the same object
    p3 = _params(r_s=1.0, lambda_end=80.0, r_exit=40.0)
    want3, _ = _device_frame_image(ctx, W, H, S, p3, sky, cam, euler, 0.9, objects=(sph, rgb, lamps))
    fr.set_scene(None, spheres=sph, sphere_rgb=rgb, lamps=lamps)
    assert np.array_equal(fr.render(p3), want3)
    p4 = _params(r_s=1.0, lambda_end=60.0, rhs_form=2, spin=0.45)
    want4, _ = _device_frame_image(ctx, W, H, S, p4, sky, cam, euler, 0.9)
    fr.set_scene(None)
    assert np.array_equal(fr.render(p4), want4) and fr.info()["directions_only"]
    # the trace parameters and the scene must agree about the disk
    from blackhole_geodesic_calculator_amd import _ffi
    with pytest.raises(_ffi.BhgError) as ei:
        fr.render(p)
    assert ei.value.code == _ffi.E_INVALID
    fr.close()


@pytest.mark.parametrize("devices", [[0], [0, 0]])
def test_set_camera_moves_an_existing_frame(ctx, devices):
    """bhg_frame_set_camera: an animation keeps ONE frame object and moves its camera.  After a move (origin only: the rays
    stay; rotation + field of view: the rays are regenerated from the kept jitter stream) the frame renders bit for bit
    what a frame created at that camera renders; the image size is fixed at creation."""
    from blackhole_geodesic_calculator_amd import _ffi
    from blackhole_geodesic_calculator_amd.device_frame import synthetic_sky
    from blackhole_geodesic_calculator_amd.raygen import euler_xyz_matrix
    W, H, S = 96, 64, 2
    sky = synthetic_sky(128, 64)
    p = _params(r_s=1.0, lambda_end=60.0)
    fr = _frame(devices, W, H, S)
    fr.set_scene(sky, spheres=[[1.0, 0.5, 9.0, 1.2]], lamps=[[5.0, 5.0, 30.0, 1.0]])
    first = fr.render(p)
    for cam, euler, fov in (((0.5, -1.0, 28.0), (0.0, 0.0, 0.0), 0.6),          # origin only
                            ((3.0, -20.0, 18.0), (0.9, 0.0, 0.1), 0.6),          # origin + rotation
                            ((3.0, -20.0, 18.0), (0.9, 0.0, 0.1), 0.45)):        # field of view
        fr.set_camera(fov_x=fov, fov_y=fov, origin=np.array(cam), rot=euler_xyz_matrix(euler))
        moved = fr.render(p)
        fresh = _frame(devices, W, H, S, cam=np.array(cam), euler=euler, fov=fov)
        fresh.set_scene(sky, spheres=[[1.0, 0.5, 9.0, 1.2]], lamps=[[5.0, 5.0, 30.0, 1.0]])
        want = fresh.render(p)
        fresh.close()
        assert np.array_equal(moved, want) and not np.array_equal(moved, first)
    # back where it started: the first image again
    fr.set_camera(fov_x=0.6, fov_y=0.6, origin=CAM, rot=None)
    assert np.array_equal(fr.render(p), first)
    cam = _ffi.Frame._camera(W + 1, H, S, 0.6, 0.6, CAM, None)
    assert _ffi.load().bhg_frame_set_camera(fr._h, __import__("ctypes").byref(cam)) != 0       # the size is fixed at creation
    fr.close()


def test_two_frame_objects_in_flight_render_alike(ctx):
    """An animation that alternates between TWO frame objects keeps two frames in flight (each object has its own contexts,
    streams and work counters: the second frame's first waves take the slots the first one's last waves leave -- 0.250 ->
    0.191 ms per frame on a 1/8-frame-sized frame).  Renders enqueued alternately on both without waiting in between come
    out right."""
    from blackhole_geodesic_calculator_amd.device_frame import synthetic_sky
    W, H, S = 128, 96, 2
    sky = synthetic_sky(128, 64)
    p = _params(r_s=1.0, lambda_end=50.0)
    fa, fb = _frame([0, 0], W, H, S), _frame([0, 0], W, H, S)
    for f in (fa, fb):
        f.set_scene(sky)
    want = fa.render(p)
    assert np.array_equal(fb.render(p), want)
    for i in range(6):                      # enqueue only, alternating objects, no synchronisation in between
        (fa if i % 2 == 0 else fb).render(p, to_host=False)
    fa.synchronize()
    fb.synchronize()
    assert np.array_equal(fa.render(p), want) and np.array_equal(fb.render(p), want)
    fa.close()
    fb.close()


def test_peer_store_frame_end_is_bit_identical(ctx):
    """BHG_FRAME_GATHER_PEER: every context's shade kernel stores its pixels straight into the first device's image (no
    slab, no gather, no assembly) -- here with three contexts of the one GPU; sky and scene frames, the same images."""
    from blackhole_geodesic_calculator_amd import _ffi
    from blackhole_geodesic_calculator_amd.device_frame import synthetic_sky
    W, H, S = 160, 96, 2
    sky = synthetic_sky(256, 128)
    p = _params(r_s=1.0, lambda_end=60.0, r_exit=40.0)
    sph, rgb, lamps = [[2.5, 1.0, 10.0, 1.5]], [[1.0, 0.8, 0.6]], [[10.0, 10.0, 30.0, 30.0]]
    imgs = {}
    for mode in (_ffi.GATHER_COPY, _ffi.GATHER_PEER):
        fr = _frame([0, 0, 0], W, H, S, gather=mode)
        fr.set_scene(sky)
        a = fr.render(p)
        fr.set_scene(None, spheres=sph, sphere_rgb=rgb, lamps=lamps)
        b = fr.render(p)
        fr.rebalance(root_share=0.8)
        c = fr.render(p)
        assert fr.info()["gather"] == ("peer" if mode == _ffi.GATHER_PEER else "copy")
        imgs[mode] = (a, b, c)
        fr.close()
    for x, y in zip(imgs[_ffi.GATHER_COPY], imgs[_ffi.GATHER_PEER]):
        assert np.array_equal(x, y)
    assert np.array_equal(imgs[_ffi.GATHER_PEER][1], imgs[_ffi.GATHER_PEER][2]) and not np.array_equal(imgs[_ffi.GATHER_PEER][0], imgs[_ffi.GATHER_PEER][1])


def test_gather_copies_through_the_peer_copy_call_give_the_same_image(ctx):
    """The copy gather between DISTINCT devices is hipMemcpyPeerAsync on the sending device's stream; contexts of one device
    use a plain device-to-device copy.  The gather mode BHG_FRAME_GATHER_COPY_PEERCALL (an explicit argument of
    bhg_frame_create; an environment variable up to round 5) sends the one-GPU frame through the peer call too -- same
    device at both ends -- so that the call, its arguments and its stream order run on this box."""
    from blackhole_geodesic_calculator_amd.device_frame import synthetic_sky
    from blackhole_geodesic_calculator_amd import _ffi
    W, H, S = 160, 96, 2
    sky = synthetic_sky(256, 128)
    p = _params(r_s=1.0, lambda_end=50.0)
    imgs = []
    for mode in (_ffi.GATHER_COPY, _ffi.GATHER_COPY_PEERCALL):
        fr = _frame([0, 0, 0, 0], W, H, S, gather=mode)
        assert fr.info()["gather"] in (_ffi.GATHER_COPY, "copy")
        fr.set_scene(sky)
        imgs.append([fr.render(p).copy() for _ in range(3)])     # (three frames: the receive block is reused, the copies wait for the assembly)
        fr.close()
    one = _frame([0], W, H, S)
    one.set_scene(sky)
    ref = one.render(p)
    one.close()
    for a, b in zip(*imgs):
        assert np.array_equal(a, b) and np.array_equal(a, ref)


def test_more_devices_than_tiles_leaves_empty_shards_and_the_same_image(ctx):
    """160 x 96 in 32-pixel tiles = 15 tiles over EIGHT contexts: one shard is empty (no pixels, no rays).  Sky, then disk +
    objects + exit sphere; before and after a re-deal by measured cost: the one-device image, bit for bit."""
    from blackhole_geodesic_calculator_amd.device_frame import synthetic_sky
    W, H, S = 160, 96, 2
    sky = synthetic_sky(64, 32)
    sph, rgb, lamps = [[3.0, 2.0, 9.0, 1.5]], [[1.0, 0.8, 0.6]], [[10.0, 10.0, 30.0, 25.0]]
    p_sky = _params(r_s=1.0, lambda_end=50.0)
    p_scn = _params(r_s=1.0, lambda_end=60.0, r_exit=40.0, disk_r_in=3.0, disk_r_out=8.0)
    imgs = {}
    for devs in ([0], [0] * 8):
        fr = _frame(devs, W, H, S)
        fr.set_scene(sky)
        a = fr.render(p_sky)
        fr.set_scene(None, disk=(3.0, 8.0), spheres=sph, sphere_rgb=rgb, lamps=lamps)
        b = fr.render(p_scn)
        info = fr.info()
        if len(devs) > 1:
            assert info["smallest_shard_pixels"] == 0 and info["n_devices"] == 8
            fr.rebalance(root_share=0.7)
            assert np.array_equal(fr.render(p_scn), b)
        st = fr.stats()
        assert st["rays"] == W * H * S
        imgs[len(devs)] = (a, b)
        fr.close()
    assert np.array_equal(imgs[1][0], imgs[8][0]) and np.array_equal(imgs[1][1], imgs[8][1])


def test_frame_argument_checks(ctx):
    from blackhole_geodesic_calculator_amd import _ffi
    from blackhole_geodesic_calculator_amd.device_frame import synthetic_sky
    for bad in (dict(devices=[]), dict(devices=[99]), dict(devices=[0, 0], gather=_ffi.GATHER_RCCL), dict(devices=[0], gather=7)):
        with pytest.raises(_ffi.BhgError):
            _ffi.Frame(bad["devices"], 16, 16, 1, gather=bad.get("gather", _ffi.GATHER_AUTO))
    fr = _ffi.Frame([0], 16, 16, 1)          # pixel centres
    with pytest.raises(_ffi.BhgError):
        fr.render(_params())                  # no sky yet
    with pytest.raises(_ffi.BhgError):
        fr.set_scene(None)                    # the first scene must bring one
    with pytest.raises(_ffi.BhgError):
        fr.stats()
    fr.set_scene(synthetic_sky(32, 16))
    img = fr.render(_params())
    assert img.shape == (16, 16, 4) and np.isfinite(img).all() and (img[..., 3] == 1.0).all()
    fr.close()


def test_stats_and_rebalance_between_a_redeal_and_the_next_render_are_refused(ctx):
    """after a re-deal the shards' pixel lists are new while the devices' steps / flags arrays still hold
    the PREVIOUS lists' rays (and may be smaller than the new shards, root_share < 1 grows the others) -- a second
    bhg_frame_stats / bhg_frame_rebalance before the next render read out of bounds and binned old steps by new pixels.  Both
    now return BHG_E_INVALID until the frame has been rendered again; a rotating camera regenerates its rays from the
    shard's RESIDENT draws of the jitter stream (no host-side gather, no upload) and gives the image a new frame gives."""
    from blackhole_geodesic_calculator_amd import _ffi
    from blackhole_geodesic_calculator_amd.device_frame import synthetic_sky
    from blackhole_geodesic_calculator_amd.raygen import euler_xyz_matrix
    W, H, S = 160, 96, 2
    sky = synthetic_sky(64, 32)
    p = _params(r_s=1.0, lambda_end=50.0)
    fr = _frame([0, 0, 0], W, H, S)
    fr.set_scene(sky)
    a = fr.render(p)
    st = fr.stats()
    fr.rebalance(root_share=0.5)              # the first device's shard shrinks, the others' grow
    for call in (fr.stats, fr.rebalance):
        with pytest.raises(_ffi.BhgError) as ei:
            call()
        assert ei.value.code == _ffi.E_INVALID and "render" in str(ei.value)
    b = fr.render(p)
    st2 = fr.stats()
    assert np.array_equal(a, b) and all(st2[k] == st[k] for k in ("rays", "attempted_steps", "accepted_steps", "horizon_rays") if k in st)
    fr.rebalance()                            # fine again after a render
    assert np.array_equal(fr.render(p), a)
    # a camera that turns: rays regenerated on the devices from the kept draws == a frame created at that camera
    rot = euler_xyz_matrix((0.03, -0.02, 0.2))
    fr.set_camera(fov_x=0.6, fov_y=0.6, origin=(1e-4, 0.0, 30.0), rot=rot)
    turned = fr.render(p)
    fr.close()
    f2 = _frame([0], W, H, S, euler=(0.03, -0.02, 0.2))
    f2.set_scene(sky)
    assert np.array_equal(f2.render(p), turned) and not np.array_equal(turned, a)
    f2.close()


def test_frame_profiling_reports_per_device_trace_times(ctx):
    from blackhole_geodesic_calculator_amd.device_frame import synthetic_sky
    fr = _frame([0, 0], 256, 256, 2)
    fr.set_scene(synthetic_sky(64, 32))
    fr.set_profiling(True)
    fr.render(_params(), to_host=False)
    tr, root = fr.last_ms()
    assert len(tr) == 2 and all(0.0 < t < 50.0 for t in tr) and 0.0 < root < 50.0
    fr.close()


_RCCL_LOOPBACK = r"""
import sys
sys.modules['torch'] = None                      # the frame object must not need it
import numpy as np
sys.path.insert(0, %(root)r)
from blackhole_geodesic_calculator_amd import _ffi
from blackhole_geodesic_calculator_amd.raygen import python_random_stream
W, H, S = 96, 64, 2
sky = np.random.default_rng(1).random((32, 64, 4)).astype(np.float32)
jit = python_random_stream(42.0, 2 * S * W * H)
p = _ffi.make_params(r_s=1.0, lambda_end=50.0)
imgs = {}
for name, g in (("copy", _ffi.GATHER_COPY), ("rccl", _ffi.GATHER_RCCL)):
    fr = _ffi.Frame([0], W, H, S, fov_x=0.6, fov_y=0.6, jitter=jit, gather=g)
    fr.set_scene(sky)
    imgs[name] = fr.render(p)
    imgs[name + "2"] = fr.render(p)
    assert fr.info()["gather"] == name, fr.info()
    fr.close()
assert np.array_equal(imgs["copy"], imgs["rccl"]) and np.array_equal(imgs["rccl"], imgs["rccl2"])
print("RCCL loopback ok", float(imgs["rccl"].sum()))
"""


def test_rccl_gather_path_on_one_gpu():
    """BHG_FRAME_GATHER_RCCL with ONE device: librccl is loaded at run time, ncclCommInitAll builds a one-rank group and
    the frame's slab goes through a grouped ncclSend / ncclRecv to itself before the assembly kernel -- the whole RCCL
    gather path of the N-GPU frame on the single GPU of this box.  In a child process (a hang must not take the suite
    down), without torch."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", _RCCL_LOOPBACK % dict(root=ROOT)], capture_output=True, text=True, timeout=150, env=env)
    assert r.returncode == 0 and "RCCL loopback ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


_ADDON_NO_TORCH = r"""
import sys, importlib, warnings
sys.modules['torch'] = None                      # Blender's bundled Python has no torch
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import numpy as np
import fake_bpy
out = {}
for devs in ("0", "0,0"):
    import os
    os.environ["BHGEO_DEVICES"] = devs
    bpy, depsgraph = fake_bpy.install(width=64, height=64, samples=2, device_shading=1.0, render_devices=float(len(devs.split(","))))
    addon = importlib.import_module("blackhole_geodesic_calculator_amd.blender_addon")
    addon.register()
    eng = addon.RelativisticRenderEngine()
    eng.render(depsgraph)
    assert eng.ended == 1 and len(eng.progress) == 64 * 2
    out[devs] = np.array(eng.result.layers[0].passes["Combined"].rect, dtype=np.float64).reshape(64, 64, 4)
    assert eng.last_device_frame["n_devices"] == len(devs.split(",")) and eng.last_device_frame["directions_only"]
    addon.unregister()
assert sys.modules.get("torch") is None
assert np.array_equal(out["0"], out["0,0"])
np.save(%(out)r, out["0"])
print("addon without torch ok")
"""


def test_addon_renders_on_the_device_without_torch(ctx, oracle, tmp_path):
    """scene.device_shading = 1 through the fake bpy with `sys.modules['torch'] = None`: the add-on's device path is the
    library-owned frame (ctypes only).  One device and two contexts of it (scene.render_devices = 2) give the same
    image, and that image is the oracle's end states shaded by the numpy restatement of the library's lookup."""
    from blackhole_geodesic_calculator_amd import camera_directions
    from oracle import shade_reference as sh
    import fake_bpy
    out = tmp_path / "img.npy"
    r = subprocess.run([sys.executable, "-c", _ADDON_NO_TORCH % dict(root=ROOT, out=str(out))], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0 and "addon without torch ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    rect = np.load(out)
    sky = fake_bpy.FakeImage("/tmp/sky.png").array      # the deterministic image the fake registry loads
    d = camera_directions(64, 64, 2, 0.6, 0.6, 42.0).reshape(-1, 3)
    o = oracle.trace(d, CAM, r_s=1.0, lambda_end=50.0)
    want = sh.shade_reduce(o["end"], o["flags"], 64 * 64, 2, sky).reshape(64, 64, 4)
    assert np.abs(rect - want).max() < 1e-6


def test_c_frame_example_runs_on_the_gpu(tmp_path):
    """examples/render_frame.c: the library-owned frame from plain C, one device and two contexts of it."""
    libdir = os.path.join(ROOT, "blackhole_geodesic_calculator_amd")
    exe = tmp_path / "render_frame"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-O2", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "render_frame.c"), "-L", libdir, "-lbhgeo",
                           "-Wl,-rpath," + libdir, "-lm", "-o", str(exe)])
    outs = []
    for devs in ("0", "0,0"):
        r = subprocess.run([str(exe), devs], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
        head, picture = r.stdout.split("\n", 1)
        assert f"{len(devs.split(','))} device(s)" in head and "30720 rays" in head
        assert "#" in picture and "o" in picture
        outs.append((head.split(":", 1)[1], picture))
    assert outs[0] == outs[1]     # same rays, steps, checksum and picture, however the frame is sharded
