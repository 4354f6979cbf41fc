"""CPU tests of the frame driver and the Blender plugin surface with a fake bpy and a STUB tracer
(flat-space straight lines computed in the test): checks the generator protocol, accumulation,
mark-window and registration logic.  No geodesic is computed here; the GPU counterpart is
tests/test_gpu_frame.py."""
import importlib
import numpy as np
import pytest

import fake_bpy
from conftest import CAM


class StubIntegrator:
    """Flat space: straight rays; 'horizon' = rays aimed inside a small cone around -z."""
    def __init__(self, cone=0.05):
        self.cone, self.calls = cone, 0

    def trace(self, k0, x0, max_step=np.inf, curve_end=50.0, r_exit=0.0, spheres=None, disk=None):
        self.calls += 1
        k0 = np.asarray(k0, float)
        end = np.concatenate([np.asarray(x0) + curve_end * k0, k0], axis=-1)
        hit = (np.hypot(k0[..., 0], k0[..., 1]) < self.cone).astype(np.uint8)
        flags = hit.copy()
        if disk is not None:   # straight lines through the plane z = 0
            t = -np.asarray(x0)[2] / k0[..., 2]
            loc = np.asarray(x0) + t[..., None] * k0
            R = np.hypot(loc[..., 0], loc[..., 1])
            on = (t > 0) & (t < curve_end) & (R >= disk[0]) & (R <= disk[1]) & (hit == 0)
            end[on, 0:3] = loc[on]
            flags = np.where(on, np.uint8(128), flags).astype(np.uint8)
        out = {"ray_end": end, "ray_blackhole_hit": hit, "flags": flags, "n_steps": hit * 0, "n_accepted": hit * 0}
        if spheres is not None:   # closed-form straight-line entry points
            t_best = np.full(k0.shape[:-1], np.inf)
            idx = np.full(k0.shape[:-1], -1, np.int8)
            for j, (cx, cy, cz, rho) in enumerate(np.asarray(spheres, float)):
                oc = np.asarray(x0) - np.array([cx, cy, cz])
                b = (k0 * oc).sum(-1)
                disc = b * b - (oc @ oc - rho * rho)
                t = np.where(disc > 0, -b - np.sqrt(np.maximum(disc, 0)), np.inf)
                upd = (t > 0) & (t < t_best) & (t < curve_end) & (hit == 0)
                t_best = np.where(upd, t, t_best)
                idx = np.where(upd, j, idx).astype(np.int8)
            h = idx >= 0
            end[h, 0:3] = np.asarray(x0) + t_best[h, None] * k0[h]
            out["object_id"] = idx
        return out


def sky(d):
    d = np.asarray(d)
    return np.stack([0.5 + 0.5 * d[..., 0], 0.5 + 0.5 * d[..., 1], np.abs(d[..., 2])], -1)


def reference_accumulate(W, H, S, fov, seed, stub, mark=None):
    """Pixel-by-pixel restatement of RelativisticRenderEngine.py:195-261 on the same stub."""
    from blackhole_geodesic_calculator_amd import camera_directions
    d = camera_directions(W, H, S, fov, fov, seed, mark=mark)
    buf = np.ones((H, W, 4))
    sbuf = np.zeros((H, W, 4))
    y0, y1, x0, x1 = mark if mark else (0, H, 0, W)
    prog = []
    for s in range(S):
        for y in range(H):
            if y0 <= y <= y1:
                for x in range(W):
                    if x0 <= x <= x1:
                        out = stub.trace(d[s, y, x][None], CAM)
                        if not out["ray_blackhole_hit"][0]:
                            sbuf[y, x, 0:3] += sky(out["ray_end"][0, 3:6])
                buf[y, :, 0:3] = sbuf[y, :, 0:3] / (s + 1)
                if y < H - 1:
                    buf[y + 1, :, 0:3] = 1 - buf[y + 1, :, 0:3]
                prog.append((s * W * H + W * y) / (S * W * H))
    return buf, prog


@pytest.mark.parametrize("mark", [None, (2, 6, 3, 9)])
def test_frame_tracer_matches_per_pixel_loop(mark):
    from blackhole_geodesic_calculator_amd.frame import FrameTracer
    W, H, S = 14, 10, 3
    stub = StubIntegrator()
    ft = FrameTracer(stub, W, H, S, fov_x=0.6, fov_y=0.6, sampling_seed=42.0, origin=CAM, mark=mark)
    buf = np.ones((H, W, 4))
    prog = list(ft.ray_trace(buf, sky))
    assert stub.calls == S  # one batched trace per sample
    want, wprog = reference_accumulate(W, H, S, 0.6, 42.0, StubIntegrator(), mark)
    assert prog == wprog
    assert np.array_equal(buf, want)
    assert np.all(buf[..., 3] == 1.0)


def test_bh_location_is_subtracted():
    from blackhole_geodesic_calculator_amd.frame import spacetime_ray_cast_batch
    stub = StubIntegrator()
    d = np.array([[0.0, 0.0, -1.0], [0.6, 0.0, -0.8]])
    hit, hit_bh, end_dir, end_loc = spacetime_ray_cast_batch(stub, [1.0, 2.0, 33.0], d, bh_loc=[1.0, 2.0, 3.0], curve_end=10.0)
    assert not hit.any() and hit_bh.tolist() == [True, False]
    assert np.allclose(end_loc, np.array([0.0, 0.0, 30.0]) + 10.0 * d) and np.array_equal(end_dir, d)


def test_equirect_uv_matches_reference_formula():
    from blackhole_geodesic_calculator_amd.frame import equirect_uv
    rng = np.random.default_rng(0)
    d = rng.normal(size=(100, 3))
    d /= np.linalg.norm(d, axis=1)[:, None]
    u, v = equirect_uv(d, normalise=False)
    theta = 1 - np.arccos(d[:, 2]) / np.pi
    phi = np.arctan2(d[:, 1], d[:, 0]) / np.pi
    assert np.array_equal(u, -phi) and np.array_equal(v, 2 * theta - 1)
    u2, v2 = equirect_uv(3.7 * d)
    assert np.allclose(u2, u) and np.allclose(v2, v)


def test_addon_plugin_surface_with_fake_bpy(monkeypatch):
    bpy, depsgraph = fake_bpy.install(width=12, height=8, samples=2)
    import importlib
    addon = importlib.import_module("blackhole_geodesic_calculator_amd.blender_addon")
    # the surface the reference exposes (RelativisticRenderEngine.py:40-45, :466-468, :504-517)
    assert addon.RelativisticRenderEngine.bl_idname == "RelRenEn"
    assert addon.RelativisticRenderEngine.bl_label == "Relativistic" and addon.RelativisticRenderEngine.bl_use_preview
    assert addon.CUSTOM_RENDER_PT_blackhole.bl_label == "Blackhole Settings"
    names = [n for n, _ in addon.PROPS]
    assert names == ["blackhole_obj", "mass", "max_integration_step", "integration_depth", "sampling_seed",
                     "field_of_view_x", "field_of_view_y", "sky_image", "mark_y_min", "mark_y_max", "mark_x_min", "mark_x_max"]
    defaults = {n: p[1].get("default") for n, p in addon.PROPS}
    assert defaults["mass"] == 0.5 and defaults["max_integration_step"] == 10000 and defaults["integration_depth"] == 50
    assert defaults["sampling_seed"] == 42 and defaults["mark_x_min"] == -1.0

    addon.register()
    assert addon.RelativisticRenderEngine in bpy._registered and hasattr(bpy.types.Scene, "mass")
    assert "RelRenEn" in addon.CUSTOM_RENDER_PT_blackhole.COMPAT_ENGINES
    import bl_ui
    assert "RelRenEn" in bl_ui.properties_render.RENDER_PT_eevee_sampling.COMPAT_ENGINES

    # render() end to end with the solver replaced by the stub (no GPU in this test)
    stub = StubIntegrator()
    monkeypatch.setattr(addon, "GeodesicIntegratorSchwarzschild", lambda **kw: stub)
    eng = addon.RelativisticRenderEngine()
    eng.render(depsgraph)
    assert stub.calls == 2 and eng.ended == 1 and eng.updates >= 2
    assert len(eng.progress) == 2 * 8 and eng.progress == sorted(eng.progress) and eng.progress[-1] < 1.0
    rect = np.array(eng.result.layers[0].passes["Combined"].rect)
    assert rect.shape == (12 * 8, 4) and np.all(rect[:, 3] == 1.0)
    assert eng.max_integration_step == np.inf  # -1 -> inf (:59-60)

    # a mesh object and a lamp in the scene: the object is traced as its bounding sphere and lit (the stub at
    # :304-305 filled in); the black-hole marker object and non-mesh objects are not obstacles
    import types
    ball = types.SimpleNamespace(type="MESH", location=(1.5, 1.0, 10.0), dimensions=(2.4, 2.4, 2.4))
    lamp = types.SimpleNamespace(type="LIGHT", location=(5.0, 5.0, 30.0))
    marker = types.SimpleNamespace(type="MESH", location=(0.0, 0.0, 0.0), dimensions=(1.0, 1.0, 1.0))
    depsgraph.scene.objects[:] = [ball, lamp, marker, types.SimpleNamespace(type="CAMERA", location=(0, 0, 30), dimensions=(1, 1, 1))]
    depsgraph.scene.blackhole_obj = marker
    # ... but only when asked to: by default scene meshes leave the image alone, as in the reference (hit = False, :304-305)
    eng1 = addon.RelativisticRenderEngine()
    eng1.render(depsgraph)
    assert np.array_equal(np.array(eng1.result.layers[0].passes["Combined"].rect), rect)
    depsgraph.scene.curved_space_objects = 1.0
    eng2 = addon.RelativisticRenderEngine()
    eng2.render(depsgraph)
    sph = eng2.scene_spheres(depsgraph)
    assert sph.shape == (1, 4) and np.allclose(sph[0], [1.5, 1.0, 10.0, 1.2])
    rect2 = np.array(eng2.result.layers[0].passes["Combined"].rect)
    changed = np.abs(rect2 - rect).max(1) > 0
    assert 0 < changed.sum() < len(rect) // 2
    many = [types.SimpleNamespace(type="MESH", location=(2.0 + j, 0.0, 9.0), dimensions=(1.0, 1.0, 1.0)) for j in range(10)]
    depsgraph.scene.objects[:] = many
    with pytest.warns(RuntimeWarning, match="only the 8 nearest"):
        assert eng2.scene_spheres(depsgraph).shape == (8, 4)
    depsgraph.scene.objects[:] = []
    depsgraph.scene.blackhole_obj = None
    depsgraph.scene.curved_space_objects = 0.0
    assert (eng.mark_y_min, eng.mark_y_max, eng.mark_x_min, eng.mark_x_max) == (0, 8, 0, 12)
    # shading goes through Blender's texture evaluate with the reference's (u, v)
    col = eng.background_hit(np.array([0.0, 1.0, 0.0]))
    assert np.allclose(col, [0.5 + 0.5 * np.sin(np.pi * -0.5), 0.5, 0.25 + 0.25 * np.cos(-np.pi)])

    addon.unregister()
    assert addon.RelativisticRenderEngine not in bpy._registered and not hasattr(bpy.types.Scene, "mass")
    assert "RelRenEn" not in addon.CUSTOM_RENDER_PT_blackhole.COMPAT_ENGINES


def test_camera_pixel_directions_and_pickle(tmp_path):
    from blackhole_geodesic_calculator_amd.camera import RelativisticCamera
    cam = RelativisticCamera(resolution=[6, 8], field_of_view=[0.6, 0.6], integrator=StubIntegrator())
    d = cam.pixel_directions()
    assert d.shape == (6, 8, 3) and np.allclose(np.linalg.norm(d, axis=-1), 1.0)
    assert np.allclose(d[3, 4], [0.0, 0.0, -1.0])  # x - int(W/2), y - int(H/2)
    cam.run()
    assert cam.ray_end.shape == (6, 8, 6) and cam.ray_blackhole_hit.shape == (6, 8) and cam.ray_blackhole_hit[3, 4] == 1
    p = tmp_path / "cam.pkl"
    cam.save(p)
    cam2 = RelativisticCamera().load(p)
    assert np.array_equal(cam2.ray_end, cam.ray_end) and np.array_equal(cam2.ray_blackhole_hit, cam.ray_blackhole_hit)
    with pytest.raises(ValueError):
        RelativisticCamera(a=1.2)
    assert RelativisticCamera(a=0.9, integrator=StubIntegrator()).a == 0.9


def test_frame_tracer_objects_fill_the_collision_stub():
    """spheres + object_hit: pixels whose ray enters a sphere are coloured by the callback with the entry
    point (world coordinates), normal and index; the rest of the frame is unchanged (:239-246)."""
    from blackhole_geodesic_calculator_amd.frame import FrameTracer, spacetime_ray_cast_batch
    W, H, S = 24, 20, 2
    bh = np.array([1.0, -2.0, 0.5])
    sph_world = [[1.0 + 1.5, -2.0 + 1.0, 0.5 + 10.0, 1.2], [1.0 - 2.0, -2.0 - 1.0, 0.5 + 5.0, 1.0]]
    seen = {}

    def object_hit(loc, normal, index):
        seen["loc"], seen["normal"], seen["index"] = loc, normal, index
        return np.stack([0.25 + 0.0 * index, 0.5 + 0.0 * index, index.astype(float)], -1)

    base = FrameTracer(StubIntegrator(), W, H, S, fov_x=0.6, fov_y=0.6, sampling_seed=42.0, origin=CAM + bh, bh_loc=bh)
    withobj = FrameTracer(StubIntegrator(), W, H, S, fov_x=0.6, fov_y=0.6, sampling_seed=42.0, origin=CAM + bh, bh_loc=bh,
                          spheres=sph_world, object_hit=object_hit)
    b0, b1 = np.ones((H, W, 4)), np.ones((H, W, 4))
    list(base.ray_trace(b0, sky))
    list(withobj.ray_trace(b1, sky))
    changed = np.abs(b0 - b1).max(-1) > 0
    assert 10 < changed.sum() < W * H // 2
    c = np.array(sph_world)[seen["index"]]
    assert np.allclose(np.linalg.norm(seen["loc"] - c[:, 0:3], axis=1), c[:, 3])           # entry points on the spheres
    assert np.allclose(seen["normal"], (seen["loc"] - c[:, 0:3]) / c[:, 3:4])
    d = base.directions()[0]
    hit, hit_bh, end_dir, end_loc = spacetime_ray_cast_batch(StubIntegrator(), CAM + bh, d, bh)
    assert not hit.any()                                                                     # the stub's `hit = False`
    hit, *_ = spacetime_ray_cast_batch(StubIntegrator(), CAM + bh, d, bh, spheres=sph_world)
    assert hit.any() and not (hit & hit_bh).any()


def test_frame_tracer_disk_colour():
    """disk=(R_in, R_out): pixels whose ray ends on the disk get the Limited engine's disk colour
    (checkHitDisk profile); everything else is unchanged."""
    from blackhole_geodesic_calculator_amd.frame import FrameTracer, disk_colour
    W, H, S = 24, 20, 1
    base = FrameTracer(StubIntegrator(), W, H, S, fov_x=0.6, fov_y=0.6, sampling_seed=42.0, origin=CAM)
    withdisk = FrameTracer(StubIntegrator(), W, H, S, fov_x=0.6, fov_y=0.6, sampling_seed=42.0, origin=CAM, disk=(3.0, 7.0))
    b0, b1 = np.ones((H, W, 4)), np.ones((H, W, 4))
    list(base.ray_trace(b0, sky))
    list(withdisk.ray_trace(b1, sky))
    d = base.directions()[0]
    t = -CAM[2] / d[..., 2]
    loc = CAM + t[..., None] * d
    R = np.hypot(loc[..., 0], loc[..., 1])
    on = (R >= 3.0) & (R <= 7.0)
    assert on.sum() > 20
    assert np.allclose(b1[on][:, 0:3], disk_colour(loc[on], 3.0, 7.0))
    assert np.array_equal(b1[~on], b0[~on])
    # the profile: white texture, peak where scale == mean
    c = disk_colour(np.array([[3.0 + 0.2 * 4.0, 0.0, 0.0]]), 3.0, 7.0)
    assert np.allclose(c, 1.0 / np.sqrt(2 * np.pi * 0.3))


def test_addon_object_lighting_is_the_contracts_lambert_model_with_shadow_rays():
    """ONE lighting contract for objects in the curved region: the add-on's host routine (spacetime_hit_many) is, term
    for term, what the device kernel implements and oracle/shade_reference.object_colour restates -- Lambert lamps,
    intensity^2 n.l / d^2, n.l clamped at 0, straight light paths, shadow rays against the other traced spheres
    (raytracer/RelativisticRenderEngine.py:341-363)."""
    import types
    from oracle import shade_reference as sh
    bpy, depsgraph = fake_bpy.install(width=8, height=8, samples=1)
    addon = importlib.import_module("blackhole_geodesic_calculator_amd.blender_addon")
    eng = addon.RelativisticRenderEngine()
    spheres = np.array([[2.0, 1.0, 8.0, 1.5], [2.6, 1.6, 11.0, 0.8], [-3.0, 0.5, -1.0, 1.2]])
    lamps = [types.SimpleNamespace(type="LIGHT", location=(4.0, 3.0, 20.0)), types.SimpleNamespace(type="LIGHT", location=(-10.0, 0.0, 2.0))]
    eng.lamps, eng._lit_spheres = lamps, spheres
    rng = np.random.default_rng(4)
    idx = rng.integers(0, 3, 400)
    n = rng.normal(size=(400, 3))
    n /= np.linalg.norm(n, axis=1)[:, None]
    loc = spheres[idx, 0:3] + spheres[idx, 3:4] * n
    got = eng.spacetime_hit_many(loc, n, idx)
    end = np.concatenate([loc, np.zeros((400, 3))], 1)
    want = sh.object_colour(end, idx, spheres, np.ones((3, 3)), [[*l.location, 10.0] for l in lamps])
    assert np.abs(got - want).max() < 1e-12
    lit = want.sum(1) > 0
    assert 0.2 < lit.mean() < 0.9            # some points lit, some on the far side
    # the second sphere sits between the first one and lamp 0: with it removed more of sphere 0 is lit
    eng._lit_spheres = spheres[[0, 2]]
    alone = eng.spacetime_hit_many(loc[idx == 0], n[idx == 0], np.zeros((idx == 0).sum(), dtype=int))
    assert (alone.sum(1) > got[idx == 0].sum(1) + 1e-12).any()
    names = [p[0] for p in addon.PROPS + addon.EXTRA_PROPS]
    assert "device_shading" in names and "curved_space_objects" in names and len(addon.PROPS) == 12


def test_addon_sky_identity_follows_the_content(tmp_path):
    """The device path keeps the sky image on the GPU while its identity stands still (blender_addon._sky_identity).  The
    identity must move whenever the CONTENT may have: a file rewritten outside Blender (same name, size, path; is_dirty
    False), other pixels under the same name, another colour space, another frame of an image sequence, unsaved edits,
    an explicit invalidate_device_sky()."""
    import types
    bpy, depsgraph = fake_bpy.install(width=8, height=8, samples=1)
    addon = importlib.import_module("blackhole_geodesic_calculator_amd.blender_addon")
    f = tmp_path / "sky.png"
    f.write_bytes(b"a" * 100)
    img = fake_bpy.FakeImage(str(f))
    bpy.data.images["sky.png"] = img
    eng = addon.RelativisticRenderEngine()
    eng.sky_image_path = str(f)
    id0 = eng._sky_identity()
    assert id0 is not None and id0 == eng._sky_identity()             # stands still while nothing changes
    # the file rewritten in place (another length; mtime too)
    f.write_bytes(b"b" * 120)
    id1 = eng._sky_identity()
    assert id1 is not None and id1 != id0
    # other pixels, everything else equal (an image reloaded from an identical-looking file): the strided checksum
    img.array = np.ascontiguousarray(img.array[:, ::-1])
    img.pixels = img.array.reshape(-1).tolist()
    id2 = eng._sky_identity()
    assert id2 is not None and id2 != id1
    # colour space
    img.colorspace_settings = types.SimpleNamespace(name="Linear")
    id3 = eng._sky_identity()
    assert id3 != id2
    img.colorspace_settings.name = "sRGB"
    assert eng._sky_identity() != id3
    # an image sequence: the scene's frame is part of the identity
    img.source = "SEQUENCE"
    bpy.context.scene.frame_current = 3
    id4 = eng._sky_identity()
    bpy.context.scene.frame_current = 4
    assert eng._sky_identity() != id4
    # unsaved edits and the explicit hook: cannot be proven unchanged -> None (read and upload), the hook once
    addon.invalidate_device_sky()
    assert eng._sky_identity() is None and eng._sky_identity() is not None
    img.is_dirty = True
    assert eng._sky_identity() is None
    # no image at all is an identity of its own
    eng.sky_image_path = ""
    assert eng._sky_identity() == ("<none>",)
