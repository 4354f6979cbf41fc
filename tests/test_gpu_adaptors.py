"""GPU tests of the host-side mirrors of the reference's interface on the resident-ray path: the camera rays are
generated on the device (bhg_rays_create) and only what the caller reads comes back (bhg_rays_trace) -- frame driver,
pre-traced camera, and the Blender add-on end to end through the fake-bpy harness with the REAL integrator
(raytracer/RelativisticRenderEngine.py:50-168: render -> render_scene -> ray_trace -> layer.rect), against an image
built from the oracle.  Plus Kerr physics checked on what the device returns."""
import importlib

import numpy as np
import pytest

import fake_bpy
from conftest import CAM

pytestmark = pytest.mark.gpu


def _params(**kw):
    from blackhole_geodesic_calculator_amd import _ffi
    return _ffi.make_params(**kw)


def test_resident_rays_equal_host_generated_rays_bit_for_bit(ctx):
    """Unrotated camera: device ray generation reproduces the host restatement of :224-230 exactly, so tracing the
    resident set gives the same bits as uploading camera_directions() -- full frame, a mark window (compact jitter
    stream: the engine only draws inside the window, :219) and pixel centres."""
    from blackhole_geodesic_calculator_amd import GeodesicIntegratorSchwarzschild, camera_directions
    from blackhole_geodesic_calculator_amd.camera import RelativisticCamera
    from blackhole_geodesic_calculator_amd.raygen import python_random_stream
    gi = GeodesicIntegratorSchwarzschild(mass=0.5, context=ctx)
    W, H, S = 96, 80, 3
    d = camera_directions(W, H, S, 0.6, 0.6, 42.0)
    ref = ctx.trace(d.reshape(-1, 3), CAM, _params(r_s=1.0, lambda_end=50.0))
    rs = gi.ray_set(W, H, S, 0.6, 0.6, CAM, jitter=python_random_stream(42.0, 2 * S * W * H))
    assert rs.n == S * W * H
    o = gi.trace_rays(rs, want=("end", "end_loc", "end_dir", "flags", "n_steps", "n_accepted"))
    assert np.array_equal(o["end"], ref[0]) and np.array_equal(o["flags"], ref[1])
    assert np.array_equal(o["n_steps"], ref[2]) and np.array_equal(o["n_accepted"], ref[3])
    assert np.array_equal(o["end_loc"], ref[0][:, 0:3]) and np.array_equal(o["end_dir"], ref[0][:, 3:6])
    # a sub-range (one sample) with only direction + flags coming back
    P = W * H
    o1 = gi.trace_rays(rs, first=P, n=P)
    assert set(o1) == {"end_dir", "flags"} and np.array_equal(o1["end_dir"], ref[0][P:2 * P, 3:6])
    rs.close()
    # mark window
    mark = (10, 49, 20, 70)
    dm = camera_directions(W, H, S, 0.6, 0.6, 42.0, mark=mark)
    rows, cols = np.arange(10, 50), np.arange(20, 71)
    dw = dm[:, rows][:, :, cols]
    refw = ctx.trace(dw.reshape(-1, 3), CAM, _params(r_s=1.0, lambda_end=50.0))
    pix = (rows[:, None] * W + cols[None, :]).reshape(-1)
    rsw = gi.ray_set(W, H, S, 0.6, 0.6, CAM, jitter=python_random_stream(42.0, 2 * S * len(pix)), jitter_is_compact=True, pixels=pix)
    ow = gi.trace_rays(rsw, want=("end", "flags"))
    assert np.array_equal(ow["end"], refw[0]) and np.array_equal(ow["flags"], refw[1])
    rsw.close()
    # pixel centres: the pre-traced camera
    cam = RelativisticCamera(resolution=[H, W], field_of_view=[0.6, 0.6], camera_location=CAM, integrator=gi)
    cam.run()
    refc = ctx.trace(cam.pixel_directions().reshape(-1, 3), CAM, _params(r_s=1.0, lambda_end=50.0))
    assert np.array_equal(cam.ray_end.reshape(-1, 6), refc[0]) and np.array_equal(cam.results["flags"].reshape(-1), refc[1])
    with pytest.raises(Exception):
        gi.trace_rays(gi.ray_set(8, 8, 1, 0.6, 0.6, CAM), first=60, n=10)      # range beyond the set


def test_frame_tracer_resident_path_equals_upload_path(ctx):
    from blackhole_geodesic_calculator_amd import GeodesicIntegratorSchwarzschild
    from blackhole_geodesic_calculator_amd.frame import FrameTracer, equirect_uv

    def sky(d):
        u, v = equirect_uv(d)
        return np.stack([0.5 + 0.5 * np.sin(np.pi * u), 0.5 + 0.5 * v, 0.25 + 0.25 * np.cos(2 * np.pi * u)], -1)

    gi = GeodesicIntegratorSchwarzschild(mass=0.5, context=ctx)
    bh = np.array([1.0, -2.0, 0.5])
    sph = [[1.0 + 3.0, -2.0 + 2.0, 0.5 + 9.0, 1.5]]
    hits = []

    def object_hit(loc, normal, index):
        hits.append(len(index))
        return np.stack([0.2 + 0.0 * index, 0.4 + 0.0 * index, 0.9 + 0.0 * index], -1)

    for kw, exact in ((dict(), True), (dict(mark=(4, 30, 8, 50)), True), (dict(disk=(3.0, 8.0)), True),
                      (dict(spheres=sph, object_hit=object_hit, disk=(3.0, 8.0)), True),
                      (dict(rotation_euler=(0.02, -0.01, 0.3)), False)):
        bufs = []
        for resident in (True, False):
            ft = FrameTracer(gi, 64, 48, 2, fov_x=0.6, fov_y=0.6, sampling_seed=42.0, origin=CAM + bh, bh_loc=bh, curve_end=60.0,
                             device_rays=resident, **kw)
            buf = np.ones((48, 64, 4))
            prog = list(ft.ray_trace(buf, sky))
            assert len(prog) > 0
            bufs.append(buf)
            if resident:   # a second frame reuses the resident set (static camera) and gives the same image
                assert ft._rays is not None
                rs = ft._rays
                buf2 = np.ones((48, 64, 4))
                list(ft.ray_trace(buf2, sky))
                assert ft._rays is rs and np.array_equal(buf2, buf)
        if exact:
            assert np.array_equal(bufs[0], bufs[1]), kw
        else:   # rotated camera: the device applies the rotation in a different operation order than numpy's matmul
            assert np.abs(bufs[0] - bufs[1]).max() < 1e-9
    assert hits and min(hits) > 0


def test_addon_render_on_the_gpu_matches_oracle_image(ctx, oracle):
    """config 1 geometry through the plugin surface: 64 x 64 x 1, camera (1e-4, 0, 30), fov 0.6, mass 0.5."""
    bpy, depsgraph = fake_bpy.install(width=64, height=64, samples=1)
    addon = importlib.import_module("blackhole_geodesic_calculator_amd.blender_addon")
    addon.register()
    eng = addon.RelativisticRenderEngine()
    eng.render(depsgraph)                       # builds its own GeodesicIntegratorSchwarzschild (:134) on the GPU
    assert eng.ended == 1 and len(eng.progress) == 64
    rect = np.array(eng.result.layers[0].passes["Combined"].rect).reshape(64, 64, 4)
    assert eng.GeoInt.context.last_launch()["workgroups"] > 0      # the HIP kernel ran
    # the same image from the oracle: rays by the reference's formula, black for horizon rays, Blender's texture
    # lookup (the fake's analytic sky) for the rest, one sample
    from blackhole_geodesic_calculator_amd import camera_directions
    from blackhole_geodesic_calculator_amd.frame import equirect_uv
    d = camera_directions(64, 64, 1, 0.6, 0.6, 42.0).reshape(-1, 3)
    o = oracle.trace(d, CAM, r_s=1.0, lambda_end=50.0)
    u, v = equirect_uv(o["end"][:, 3:6])
    tex = fake_bpy.FakeTexture("t", "IMAGE")
    col = np.array([tex.evaluate((float(a), float(b), 0)).xyz for a, b in zip(u, v)])
    col[(o["flags"] & 1) != 0] = 0.0
    want = col.reshape(64, 64, 3)
    assert (o["flags"] & 1).sum() > 50
    assert np.abs(rect[..., 0:3] - want).max() < 1e-7 and np.all(rect[..., 3] == 1.0)
    addon.unregister()


def test_kerr_constants_of_motion_on_device_trajectories(ctx, oracle):
    """E, L_z, Carter's Q and the null norm along curves the DEVICE integrated (bhg_trajectory samples), recomputed
    from each sample's Cartesian state through the metric itself -- independent of the generated right-hand side."""
    from blackhole_geodesic_calculator_amd import GeodesicIntegratorKerr
    from oracle import scipy_reference as sr
    M, a = 0.5, 0.45
    gi = GeodesicIntegratorKerr(mass=M, a=a / M, rtol=1e-10, atol=1e-12, context=ctx)
    cam = np.array([0.0, -25.0, 12.0])
    rng = np.random.default_rng(12)
    k = (-cam / np.linalg.norm(cam))[None, :] + rng.normal(size=(24, 3)) * 0.1
    k /= np.linalg.norm(k, axis=1)[:, None]

    def carter(q, u, E, L):
        Sig = q[0] ** 2 + a * a * np.cos(q[1]) ** 2
        return (Sig * u[1]) ** 2 + np.cos(q[1]) ** 2 * (L * L / np.sin(q[1]) ** 2 - a * a * E * E)

    escaped = 0
    for i in range(len(k)):
        k_xyz, x_xyz, res = gi.calc_trajectory(k[i], cam, curve_end=60.0, nr_points_curve=25)
        if res["hit_blackhole"]:
            continue
        escaped += 1
        q0, u0 = sr.cart_to_bl(cam, k[i], a)
        E0, L0, _ = sr.kerr_constants(q0, u0, M, a)
        Q0 = carter(q0, u0, E0, L0)
        for j in range(1, x_xyz.shape[1]):
            q, u = sr.cart_to_bl(x_xyz[:, j], k_xyz[:, j], a)
            E, L, kt = sr.kerr_constants(q, u, M, a)          # null condition re-solved at the sample
            assert abs(E - E0) < 1e-7 and abs(L - L0) < 1e-6 and abs(carter(q, u, E, L) - Q0) < 1e-5
            gtt, gtp, grr, gthth, gpp = sr.kerr_metric(q[0], q[1], M, a)
            norm = gtt * kt * kt + 2 * gtp * kt * u[2] + grr * u[0] ** 2 + gthth * u[1] ** 2 + gpp * u[2] ** 2
            assert abs(norm) < 1e-9
    assert escaped > 8


def test_kerr_kernel_at_vanishing_spin_equals_schwarzschild_kernel(ctx):
    """The Boyer-Lindquist kernel with a -> 0 against the (Cartesian, reduced-form) Schwarzschild kernel."""
    rng = np.random.default_rng(13)
    cam = np.array([3.0, -20.0, 14.0])
    k = (-cam / np.linalg.norm(cam))[None, :] + rng.normal(size=(3000, 3)) * 0.15
    k /= np.linalg.norm(k, axis=1)[:, None]
    kw = dict(r_s=1.0, lambda_end=50.0, rtol=1e-11, atol=1e-13)
    ea, fa, _, _ = ctx.trace(k, cam, _params(rhs_form=2, spin=1e-12, **kw))
    eb, fb, _, _ = ctx.trace(k, cam, _params(rhs_form=1, **kw))
    esc = (fa == 4) & (fb == 4)
    assert esc.sum() > 2000 and (fa & 1).sum() > 30
    # horizon flags: the BL event sits at r_plus (1 + 1e-3), a hair outside r_s -- the same rays are captured
    assert np.array_equal(fa & 1, fb & 1)
    assert np.abs(ea - eb)[esc].max() < 1e-6


def test_addon_with_device_shading_renders_the_frame_on_the_gpu(ctx, oracle):
    """scene.device_shading = 1: the add-on reads the sky image's pixels once and ray generation, trace, sky lookup
    and the sample mean all run on the device (the library-owned frame, bhg_frame_*) -- config 1 geometry, 64 x 64 x 1 and x 3, against an
    image built from the oracle's end states and the numpy restatement of the library's bilinear lookup."""
    from oracle import shade_reference as sh
    from blackhole_geodesic_calculator_amd import camera_directions
    for S in (1, 3):
        bpy, depsgraph = fake_bpy.install(width=64, height=64, samples=S, device_shading=1.0)
        addon = importlib.import_module("blackhole_geodesic_calculator_amd.blender_addon")
        addon.register()
        eng = addon.RelativisticRenderEngine()
        eng.render(depsgraph)
        assert eng.ended == 1 and len(eng.progress) == 64 * S and abs(eng.progress[-1] - 1.0) < 1e-12
        rect = np.array(eng.result.layers[0].passes["Combined"].rect).reshape(64, 64, 4)
        sky = bpy.data.images["sky.png"].array
        d = camera_directions(64, 64, S, 0.6, 0.6, 42.0).reshape(-1, 3)
        o = oracle.trace(d, CAM, r_s=1.0, lambda_end=50.0)
        want = sh.shade_reduce(o["end"], o["flags"], 64 * 64, S, sky).reshape(64, 64, 4)
        assert np.abs(rect - want).max() < 1e-6 and (o["flags"] & 1).sum() > 50 * S
        assert eng.last_device_frame["directions_only"]       # a sky-only frame traces exit directions alone
        addon.unregister()


def test_addon_keeps_its_device_frame_across_renders(ctx, oracle):
    """Blender makes a new engine instance for every frame of an animation; the add-on's device path keeps the library-owned
    frame between them and only moves its camera and scene (creating it costs 100 ms, rendering 2.4).  Three renders of an
    'animation' -- the camera moves, then turns -- each against an image built from the oracle; a changed resolution makes a
    new frame; unregister() frees it."""
    from oracle import shade_reference as sh
    from blackhole_geodesic_calculator_amd import camera_directions
    reused, uploaded = [], []
    bpy, depsgraph = fake_bpy.install(width=48, height=48, samples=2, device_shading=1.0)
    addon = importlib.import_module("blackhole_geodesic_calculator_amd.blender_addon")
    addon.register()
    for cam, euler, size in (((1e-4, 0.0, 30.0), (0.0, 0.0, 0.0), 48), ((2.0, -1.0, 27.0), (0.0, 0.0, 0.0), 48),
                             ((2.0, -1.0, 27.0), (0.05, -0.04, 0.3), 48), ((2.0, -1.0, 27.0), (0.05, -0.04, 0.3), 32)):
        depsgraph.scene.camera.matrix_world = fake_bpy._Matrix(cam, euler)
        depsgraph.scene.render.resolution_x = depsgraph.scene.render.resolution_y = size
        eng = addon.RelativisticRenderEngine()          # (a new engine instance per frame, as Blender does)
        eng.render(depsgraph)
        reused.append(eng.device_frame_reused)
        uploaded.append(eng.device_sky_uploaded)
        rect = np.array(eng.result.layers[0].passes["Combined"].rect).reshape(size, size, 4)
        d = camera_directions(size, size, 2, 0.6, 0.6, 42.0, rotation_euler=euler).reshape(-1, 3)
        o = oracle.trace(d, np.array(cam), r_s=1.0, lambda_end=50.0)
        want = sh.shade_reduce(o["end"], o["flags"], size * size, 2, bpy.data.images["sky.png"].array).reshape(size, size, 4)
        assert np.abs(rect - want).max() < 1e-6
        assert len(addon._DEVICE_FRAMES) == 1
    assert reused == [False, True, True, False]
    # the sky image crosses PCIe once per frame OBJECT, not once per render (every render re-uploaded 33 MB per device)
    assert uploaded == [True, False, False, True]
    # ... unless it has unsaved edits: another image under the same name is read and uploaded again
    img = bpy.data.images["sky.png"]
    img.array = np.ascontiguousarray(img.array[:, ::-1])
    img.pixels = img.array.reshape(-1).tolist()
    img.is_dirty = True
    eng = addon.RelativisticRenderEngine()
    eng.render(depsgraph)
    assert eng.device_frame_reused and eng.device_sky_uploaded
    rect2 = np.array(eng.result.layers[0].passes["Combined"].rect).reshape(size, size, 4)
    want2 = sh.shade_reduce(o["end"], o["flags"], size * size, 2, img.array).reshape(size, size, 4)
    assert np.abs(rect2 - want2).max() < 1e-6 and np.abs(rect2 - rect).max() > 1e-3
    # ... or was changed behind Blender's back (reloaded from an edited file: same name, size, path, is_dirty False -- ADVICE
    # r05): the identity's checksum sees other pixels; the render after that one finds the same image and uploads nothing;
    # invalidate_device_sky() forces one upload
    img.is_dirty = False
    img.array = np.ascontiguousarray(img.array[::-1])
    img.pixels = img.array.reshape(-1).tolist()
    ups = []
    for hook in (False, False, True):
        if hook:
            addon.invalidate_device_sky()
        eng = addon.RelativisticRenderEngine()
        eng.render(depsgraph)
        ups.append(eng.device_sky_uploaded)
        rect3 = np.array(eng.result.layers[0].passes["Combined"].rect).reshape(size, size, 4)
        want3 = sh.shade_reduce(o["end"], o["flags"], size * size, 2, img.array).reshape(size, size, 4)
        assert np.abs(rect3 - want3).max() < 1e-6
    assert ups == [True, False, True] and np.abs(rect3 - rect2).max() > 1e-3
    addon.unregister()
    assert len(addon._DEVICE_FRAMES) == 0


def test_addon_device_and_host_paths_light_objects_alike(ctx):
    """One lighting contract for object hits (Lambert lamps with shadow rays against the other spheres): the pixels
    whose rays end on a sphere come out the same from the host path (Python, spacetime_hit_many) and from the device
    path (object_colour in the shade kernel)."""
    import types
    out = {}
    for dev in (0.0, 1.0):
        bpy, depsgraph = fake_bpy.install(width=48, height=48, samples=1, device_shading=dev, curved_space_objects=1.0,
                                          integration_depth=70)
        addon = importlib.import_module("blackhole_geodesic_calculator_amd.blender_addon")
        addon.register()
        depsgraph.scene.objects[:] = [
            types.SimpleNamespace(type="MESH", location=(1.5, 1.0, 10.0), dimensions=(3.0, 3.0, 3.0)),
            types.SimpleNamespace(type="MESH", location=(2.3, 1.9, 13.0), dimensions=(1.4, 1.4, 1.4)),
            types.SimpleNamespace(type="LIGHT", location=(5.0, 5.0, 30.0))]
        eng = addon.RelativisticRenderEngine()
        eng.render(depsgraph)
        out[dev] = np.array(eng.result.layers[0].passes["Combined"].rect).reshape(48, 48, 4)
        if dev:
            assert not eng.last_device_frame["directions_only"]
            from blackhole_geodesic_calculator_amd import camera_directions
            d = camera_directions(48, 48, 1, 0.6, 0.6, 42.0).reshape(-1, 3)
            flags = ctx.trace(d, CAM, _params(r_s=1.0, lambda_end=70.0), spheres=[[1.5, 1.0, 10.0, 1.5], [2.3, 1.9, 13.0, 0.7]])[1]
            hit = (flags == 0x88).reshape(48, 48)
        addon.unregister()
    assert hit.sum() > 30
    # (the device path hands back float RGBA -- what layer.rect holds, :163-164 -- the host path float64)
    assert np.abs(out[0.0][hit] - out[1.0][hit]).max() < 2e-7
    assert out[1.0][hit][:, :3].max() > 0 and (out[1.0][hit][:, :3].sum(1) == 0).any()    # lit and shadowed / far-side points


def test_limited_engine_adaptor_against_oracle(ctx, oracle):
    """blackhole_geodesic_calculator_amd.limited.SchwarzschildGeodesic -- the solver call of the Limited engine
    (raytracer/LimitedRelativisticRenderEngine.py:90, :273-279, :308-314): rays started on the object sphere at
    ratio 30 against oracle.trace(r_exit=...), per ray and batched, with the disk and in flat space."""
    from blackhole_geodesic_calculator_amd.limited import SchwarzschildGeodesic
    sw = SchwarzschildGeodesic(metric="schwarzschild", context=ctx)
    ratio = 30.0
    assert sw.approximateCurveEnd(ratio) == 50 + 2 * 50 * (ratio / 20 - 1) == 100.0
    rng = np.random.default_rng(12)
    n = 4000
    # hit points on the sphere r = ratio (a mesh hit lies on a facet: up to 1 % inside), directions aimed near the hole
    p = rng.normal(size=(n, 3))
    p = ratio * (1.0 - 0.01 * rng.random(n))[:, None] * p / np.linalg.norm(p, axis=1)[:, None]
    d = rng.normal(size=(n, 3)) * 6.0 - p
    d /= np.linalg.norm(d, axis=1)[:, None]
    end_loc, end_dir, mes = sw.ray_trace_many(d, p, exit_tolerance=0.2, ratio_obj_to_blackhole=ratio)
    o = oracle.trace(d, p, r_s=1.0, lambda_end=100.0, r_exit=ratio)
    assert np.array_equal(mes["flags"], o["flags"]) and np.array_equal(mes["n_steps"], o["n_attempted"])
    assert np.array_equal(mes["hit_blackhole"], (o["flags"] & 1) != 0) and not mes["outside"].any()
    esc = o["flags"] == 8
    assert esc.sum() > 0.8 * n and mes["hit_blackhole"].sum() > 50
    assert np.abs(np.linalg.norm(end_loc[esc], axis=1) - ratio).max() < 1e-9           # they end ON the exit sphere
    assert np.abs(end_loc[esc] - o["end"][esc, 0:3]).max() < 1e-7 and np.abs(end_dir[esc] - o["end"][esc, 3:6]).max() < 1e-7
    # the per-ray form: same end state, a sampled path that starts at the hit point and ends inside the sphere
    for i in np.nonzero(esc)[0][:5]:
        x, y, z, el, ed, m = sw.ray_trace(d[i], loc_hit=p[i], exit_tolerance=0.2, ratio_obj_to_blackhole=ratio,
                                          curve_end=sw.approximateCurveEnd(ratio), max_step=np.inf)
        assert not m["hit_blackhole"] and "error" not in m
        # (the sampled-path kernel locates the exit with Brent, the frame kernel with the certified Newton search: one root, to rounding)
        assert np.abs(el - end_loc[i]).max() < 1e-10 and np.abs(ed - end_dir[i]).max() < 1e-10
        assert abs(x[0] - p[i, 0]) < 1e-12 and len(x) == len(y) == len(z) > 5
        assert np.sqrt(x * x + y * y + z * z).max() <= ratio * (1 + 1e-9)
    i = int(np.nonzero(mes["hit_blackhole"])[0][0])
    assert sw.ray_trace(d[i], loc_hit=p[i], ratio_obj_to_blackhole=ratio)[5]["hit_blackhole"]
    # a start beyond the tolerated radius is reported, not integrated (the engine paints it red, :311-314)
    far = p[0] * 1.25
    m = sw.ray_trace(d[0], loc_hit=far, exit_tolerance=0.2, ratio_obj_to_blackhole=ratio)[5]
    assert m["error"] == "Outside" and not m["hit_blackhole"]
    _, _, mm = sw.ray_trace_many(d[:2], np.stack([far, p[1]]), exit_tolerance=0.2, ratio_obj_to_blackhole=ratio)
    assert mm["outside"].tolist() == [True, False]
    # ... while one within the slack is integrated as it stands: it crosses the sphere inward first, and ends on the way out
    near = p[1] * (ratio * 1.1 / np.linalg.norm(p[1]))
    el, ed, mm = sw.ray_trace_many(d[1:2], near[None, :], exit_tolerance=0.2, ratio_obj_to_blackhole=ratio)
    assert not mm["outside"][0] and (mm["hit_blackhole"][0] or abs(np.linalg.norm(el[0]) - ratio) < 1e-9)
    # the "approximate" solver object of the same engine (:97-101, :269) is the exact solve here
    from blackhole_geodesic_calculator_amd.limited import ApproxSchwarzschildGeodesic
    asw = ApproxSchwarzschildGeodesic(ratio_obj_to_blackhole=ratio, exit_tolerance=0.2, context=ctx)
    assert round(asw.exit_tolerance, 4) == 0.2 and round(asw.ratio_obj_to_blackhole, 4) == 30.0       # what :98 compares
    j = int(np.nonzero(esc)[0][0])
    el1, ed1, m1 = asw.generatedRayTracer(p[j], d[j])
    assert np.array_equal(el1, end_loc[j]) and np.array_equal(ed1, end_dir[j]) and m1 == {"hit_blackhole": False}
    assert asw.generatedRayTracer(far, d[0])[2]["error"] == "Outside"
    ela, eda, ma = asw.generatedRayTracer_many(p, d)
    assert np.array_equal(ela, end_loc) and np.array_equal(ma["flags"], mes["flags"])
    # the disk, with the engine's radii (disk_R_in * ratio, :285): against the oracle's disk event
    disk = (0.15 * ratio, 0.35 * ratio)
    el, ed, md = sw.ray_trace_many(d, p, ratio_obj_to_blackhole=ratio, disk=disk)
    od = oracle.trace(d, p, r_s=1.0, lambda_end=100.0, r_exit=ratio, disk_r_in=disk[0], disk_r_out=disk[1])
    assert np.array_equal(md["flags"], od["flags"]) and md["hit_disk"].sum() > 100
    assert np.abs(el[md["hit_disk"], 2]).max() < 1e-12
    # metric='flat' (README.md:233): straight lines through the sphere, nothing hits a hole
    fl = SchwarzschildGeodesic(metric="flat", context=ctx)
    el, ed, mf = fl.ray_trace_many(d, p, ratio_obj_to_blackhole=ratio)
    assert not mf["hit_blackhole"].any() and np.abs(ed - d).max() < 1e-12
    chord = -2.0 * (p * d).sum(1)                                     # exit point of the straight line p + t d on |x| = ratio ...
    t_exit = 0.5 * (chord + np.sqrt(chord * chord - 4.0 * ((p * p).sum(1) - ratio * ratio)))
    through = t_exit < 100.0                                          # ... where curve_end lets it get there
    assert through.sum() > 0.9 * n and np.abs(el[through] - (p + t_exit[:, None] * d)[through]).max() < 1e-7
