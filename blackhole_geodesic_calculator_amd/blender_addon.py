"""Blender add-on: the reference's render-engine plugin surface over the MI355X integrator.

The plugin surface is unchanged with respect to raytracer/RelativisticRenderEngine.py -- the
engine id ("RelRenEn", :42), the `render(self, depsgraph)` entry (:50), the result protocol
(begin_result / layer.rect / update_result / update_progress / end_result, :158-168), the
"Blackhole Settings" panel (:466-495), the twelve scene properties with their names and defaults
(:504-517) and register()/unregister() (:537-580).  What changes is the per-ray hand-off: instead
of one `curvedpy ... calc_trajectory` Python call per ray (:293-294) the frame driver issues one
batched GPU trace per sample (frame.FrameTracer), shading stays Blender's `Texture.evaluate`.

This module imports `bpy`; it is loaded by Blender (or by the tests' fake-bpy harness), never by
the package __init__.

Known deviations from the reference, all deliberate:
  * progress: the reference divides the already-fractional progress by res_y again (:166 vs
    :261, so its bar stays near 0); here update_progress receives the fraction itself;
  * the RenderResult is refreshed once per sample instead of once per row (`buf.tolist()` per row
    is O(H^2 W) Python work, :163); the final image is identical;
  * one scene property more than the reference's twelve: `curved_space_objects` (default 0 = off).  Off, the
    image is the reference's: its spacetime_ray_cast always reports hit = False (:304-305), so scene meshes
    never affect a render.  Set to 1, the MESH objects of the scene (except the black-hole marker) are traced as
    lamp-lit bounding spheres inside the curved region -- the collision test the reference leaves as a stub;
    at most 8 (the solver's limit), the nearest to the hole, with a warning when more are dropped.  ONE lighting
    contract for an object hit, whichever path shades it: `spacetime_hit`'s Lambert lamps (:341-363) -- colour +=
    intensity^2 n.l / d^2 per lamp, light paths straight, n.l clamped at 0 -- WITH its shadow rays (:352-356), cast
    against the other traced spheres (the only geometry the curved region knows); the host routine below and the
    device kernel (`object_colour`, csrc/frame_kernels.hip; restated in oracle/shade_reference.py) agree to rounding;
  * a second extra property, `device_shading` (default 0 = off).  Off: every escaping ray is shaded by Blender's own
    `Texture.evaluate`, one Python call per ray as in the reference (:366-378) -- at 1024 x 1024 that is a million
    calls per sample next to a quarter-millisecond trace.  Set to 1: the sky image's pixels are read ONCE
    (`bpy.data.images[...].pixels`), uploaded, and ray generation, trace, sky lookup, object lighting and the
    multisample mean all run on the device (device_frame.DeviceFrame: bhg_raygen_device -> bhg_trace[_dir]_device ->
    bhg_shade[_dir/_scene]_device); one array comes back per frame.  The lookup is then the library's bilinear
    filter of the image, not Blender's texture filter (which cannot be reproduced outside Blender): the images agree
    to the filter difference.  Needs the whole frame (no mark_* window: the window's RNG alignment is a host matter);
    with a window set the host path is used.
"""
import warnings
import os

import bpy
import numpy as np
from bl_ui.properties_render import RenderButtonsPanel
from bpy.types import Panel

from .frame import FrameTracer, equirect_uv
from .integrator import GeodesicIntegratorSchwarzschild

bl_info = {
    "name": "Relativistic Render Engine (MI355X)",
    "bl_label": "Relativistic Render Engine",
    "blender": (4, 1, 0),
    "category": "Render",
}


def _unset(value, default):
    """-1 is the reference's "unset" sentinel for scene properties (:59-60, :111-118)."""
    return default if value == -1 else value


class RelativisticRenderEngine(bpy.types.RenderEngine):
    bl_idname = "RelRenEn"
    bl_label = "Relativistic"
    bl_use_preview = True

    # ---- Blender entry point (:50) -----------------------------------------------------------
    def render(self, depsgraph):
        scene = depsgraph.scene
        self.max_integration_step = _unset(scene.max_integration_step, np.inf)
        self.sampling_seed = scene.sampling_seed
        self.int_depth_curve_end = scene.integration_depth

        self.samples = bpy.data.scenes["Scene"].eevee.taa_render_samples
        self.scale = scene.render.resolution_percentage / 100.0
        self.res_x = int(scene.render.resolution_x * self.scale)
        self.res_y = int(scene.render.resolution_y * self.scale)
        self.field_of_view_x = scene.field_of_view_x
        self.field_of_view_y = scene.field_of_view_y

        self._load_sky(scene.sky_image)

        self.mass = scene.mass
        bh = scene.blackhole_obj
        self.bh_loc = np.zeros(3) if bh is None else np.array(list(bh.location), dtype=np.float64)

        self.mark_y_min = _unset(scene.mark_y_min, 0)
        self.mark_y_max = _unset(scene.mark_y_max, self.res_y)
        self.mark_x_min = _unset(scene.mark_x_min, 0)
        self.mark_x_max = _unset(scene.mark_x_max, self.res_x)
        self.device_shading = float(getattr(scene, "device_shading", 0) or 0) != 0.0
        self.render_devices = int(float(getattr(scene, "render_devices", 1) or 1))

        # one solver object per frame, as the reference builds it (:134)
        self.GeoInt = GeodesicIntegratorSchwarzschild(mass=self.mass, time_like=False, verbose=False)

        # the reference refreshes the depsgraph so that "-f <frame>" renders work (:140-141)
        depsgraph = bpy.context.evaluated_depsgraph_get()
        depsgraph.update()

        if not self.is_preview:
            self.render_scene(depsgraph)

    def _load_sky(self, path):
        self.sky_image_path = path
        name = os.path.basename(path)
        self.sky_tex_name = name + "_tex"
        if name and name not in bpy.data.images:
            bpy.data.images.load(path)
        if name and self.sky_tex_name not in bpy.data.textures:
            tex = bpy.data.textures.new(self.sky_tex_name, "IMAGE")
            tex.image = bpy.data.images[name]

    # ---- result hand-back (:152-168) ---------------------------------------------------------
    def render_scene(self, depsgraph):
        buf = np.ones((self.res_y, self.res_x, 4))
        result = self.begin_result(0, 0, self.res_x, self.res_y)
        layer = result.layers[0].passes["Combined"]
        rows_per_refresh = max(1, self.res_y)
        for i, frac in enumerate(self.ray_trace(depsgraph, self.res_x, self.res_y, 1, buf, self.samples)):
            self.update_progress(frac)
            if (i + 1) % rows_per_refresh == 0:
                self._set_rect(layer, buf)
                self.update_result(result)
        self._set_rect(layer, buf)
        self.update_result(result)
        self.end_result(result)

    @staticmethod
    def _set_rect(layer, buf):
        """layer.rect <- buf [H, W, 4]: ONE flat float32 array through foreach_set where the pass offers it (Blender's
        bpy_prop_array does), else the reference's list of rows (:163; a million-pixel .tolist() is a second of Python)."""
        rect = getattr(layer, "rect", None)
        if rect is not None and hasattr(rect, "foreach_set"):
            rect.foreach_set(np.ascontiguousarray(buf, dtype=np.float32).reshape(-1))
        else:
            layer.rect = buf.reshape(-1, 4).tolist()

    # ---- frame driver (:172-267): same generator protocol, batched hot path -------------------
    def ray_trace(self, depsgraph, width, height, depths, buf, samples, approx=False):
        cam = depsgraph.scene.camera.matrix_world
        origin = np.array(list(cam.translation), dtype=np.float64)
        rotation = tuple(cam.to_euler())
        mark = None
        if (self.mark_y_min, self.mark_y_max, self.mark_x_min, self.mark_x_max) != (0, height, 0, width):
            mark = (self.mark_y_min, self.mark_y_max, self.mark_x_min, self.mark_x_max)
        self.lamps = [ob for ob in depsgraph.scene.objects if ob.type == "LIGHT"]  # :175
        # objects in the curved region are opt-in (scene.curved_space_objects): off reproduces the reference image
        want_objects = float(getattr(depsgraph.scene, "curved_space_objects", 0) or 0) != 0.0
        spheres = self.scene_spheres(depsgraph) if want_objects else np.zeros((0, 4))
        self._lit_spheres = spheres
        if getattr(self, "device_shading", False) and mark is None:
            yield from self.ray_trace_device(width, height, samples, buf, origin, rotation, spheres)
            return
        tracer = FrameTracer(
            self.GeoInt, width, height, samples, fov_x=self.field_of_view_x, fov_y=self.field_of_view_y,
            sampling_seed=self.sampling_seed, origin=origin, rotation_euler=rotation, bh_loc=self.bh_loc,
            max_step=self.max_integration_step, curve_end=self.int_depth_curve_end, mark=mark,
            spheres=spheres if len(spheres) else None, object_hit=self.spacetime_hit_many)
        yield from tracer.ray_trace(buf, self.background_hit_many)

    # ---- objects in the curved region: the collision test the reference leaves as a stub (:304-305) ----
    def scene_spheres(self, depsgraph):
        """Mesh objects of the scene as bounding spheres [[x, y, z, radius]] in world coordinates (the
        README's orbiting-sphere animation, README.md:9-13, uses a UV sphere); the black-hole marker object
        is skipped.  At most 8 (the solver's limit), the nearest to the hole first."""
        bh = getattr(depsgraph.scene, "blackhole_obj", None)
        out = []
        for ob in depsgraph.scene.objects:
            if ob.type != "MESH" or ob is bh:
                continue
            loc = np.array(list(ob.location), dtype=np.float64)
            radius = 0.5 * max(float(d) for d in ob.dimensions)
            if radius > 0.0:
                out.append([loc[0], loc[1], loc[2], radius])
        out.sort(key=lambda s: float(np.linalg.norm(np.array(s[:3]) - self.bh_loc)))
        if len(out) > 8:
            warnings.warn(f"{len(out)} mesh objects in the scene: only the 8 nearest to the hole are traced", RuntimeWarning)
        return np.array(out[:8], dtype=np.float64).reshape(-1, 4)

    LAMP_INTENSITY = 10.0   # spacetime_hit's default `intensity` (:317)

    def spacetime_hit_many(self, loc, normal, index, intensity=None):
        """Vectorised spacetime_hit (:317-363): white Lambert lamps, colour += intensity^2 n.l / d^2, n.l clamped at 0,
        light paths straight, shadow rays (:352-356) cast against the OTHER traced spheres (world coordinates) -- the
        same model, term for term, as the device kernel's object_colour.  loc, normal [n, 3]; index [n] = sphere hit."""
        intensity = self.LAMP_INTENSITY if intensity is None else intensity
        loc = np.asarray(loc, dtype=np.float64).reshape(-1, 3)
        normal = np.asarray(normal, dtype=np.float64).reshape(-1, 3)
        index = np.asarray(index).reshape(-1)
        spheres = np.asarray(getattr(self, "_lit_spheres", np.zeros((0, 4))), dtype=np.float64).reshape(-1, 4)
        color = np.zeros(loc.shape)
        for lamp in getattr(self, "lamps", []):
            light_vec = np.array(list(lamp.location), dtype=np.float64) - loc
            d2 = (light_vec * light_vec).sum(-1)
            dist = np.sqrt(d2)
            light_dir = light_vec / dist[:, None]
            ndl = (normal * light_dir).sum(-1)
            lit = ndl > 0.0
            for q in range(len(spheres)):
                oc = loc - spheres[q, 0:3]
                b = (oc * light_dir).sum(-1)
                disc = b * b - ((oc * oc).sum(-1) - spheres[q, 3] ** 2)
                sd = np.sqrt(np.maximum(disc, 0.0))
                t0, t1 = -b - sd, -b + sd
                blocked = (disc > 0.0) & (((t0 > 1e-5) & (t0 < dist)) | ((t0 <= 1e-5) & (t1 > 1e-5))) & (index != q)
                lit &= ~blocked
            color += np.where(lit, intensity * intensity * ndl / d2, 0.0)[:, None]
        return color

    # ---- the whole frame on the device (opt-in, scene.device_shading) ------------------------------------------------
    SKY_CHECKSUM_SAMPLES = 1024      # pixel values read for the identity's checksum (of w * h * 4)

    def _sky_identity(self):
        """What tells one sky image's CONTENT from another's without reading all of its pixels (33 MB for a 2k x 1k image):
        name, size, file path -- and what moves when the content does: the file's mtime and length (an image edited
        outside Blender and reloaded keeps name, size and path, and `is_dirty` stays False), the image source with the
        scene's current frame for sequences and movies, the colour space and alpha mode, and a checksum over
        SKY_CHECKSUM_SAMPLES pixel values at a fixed stride.  None -- "cannot be proven unchanged: read and upload" -- while
        the image has unsaved edits (`is_dirty`), when the pixels cannot be sampled, or after invalidate_device_sky()."""
        global _SKY_FORCE_REFRESH
        if _SKY_FORCE_REFRESH:
            _SKY_FORCE_REFRESH = False
            return None
        name = os.path.basename(getattr(self, "sky_image_path", "") or "")
        if not name or name not in bpy.data.images:
            return ("<none>",)
        img = bpy.data.images[name]
        if getattr(img, "is_dirty", False):
            return None
        size = getattr(img, "size", None) or (0, 0)
        filepath = str(getattr(img, "filepath", "") or "")
        source = str(getattr(img, "source", "FILE"))
        ident = [name, int(size[0]), int(size[1]), filepath, source,
                 str(getattr(getattr(img, "colorspace_settings", None), "name", "")), str(getattr(img, "alpha_mode", ""))]
        if source in ("SEQUENCE", "MOVIE"):
            scene = getattr(getattr(bpy, "context", None), "scene", None)
            ident.append(int(getattr(scene, "frame_current", 0) or 0))
        if source in ("FILE", "SEQUENCE", "MOVIE") and filepath and getattr(img, "packed_file", None) is None:
            path = filepath
            absp = getattr(getattr(bpy, "path", None), "abspath", None)
            if callable(absp):
                try:
                    path = absp(filepath)
                except Exception:   # noqa: BLE001
                    path = filepath
            try:
                st = os.stat(path)
                ident += [int(st.st_mtime_ns), int(st.st_size)]
            except OSError:
                ident += [None, None]      # (no such file -- a generated image, a relative path: the checksum below carries it)
        try:
            px = img.pixels
            n = int(size[0]) * int(size[1]) * 4
            if n <= 0 or len(px) < n:
                return None
            step = max(1, n // self.SKY_CHECKSUM_SAMPLES)
            # (+ 1: a stride that is a multiple of 4 would only ever look at one channel)
            vals = np.asarray([px[i] for i in range(0, n, step + (1 if step % 4 == 0 else 0))], dtype=np.float32)
            ident.append(hash(vals.tobytes()))
        except Exception:   # noqa: BLE001
            return None
        return tuple(ident)

    def _sky_pixels(self):
        """The sky image as float32 [h, w, 4], rows bottom-up as Blender stores them -- which is the library's
        convention too (texture coordinate v = -1 is row 0).  None when there is no readable image."""
        name = os.path.basename(getattr(self, "sky_image_path", "") or "")
        if not name or name not in bpy.data.images:
            return None
        img = bpy.data.images[name]
        size = getattr(img, "size", None)
        if size is None or not hasattr(img, "pixels"):
            return None
        w, h = int(size[0]), int(size[1])
        if w <= 0 or h <= 0:
            return None
        px = np.empty(w * h * 4, dtype=np.float32)
        if hasattr(img.pixels, "foreach_get"):
            img.pixels.foreach_get(px)
        else:
            px[:] = np.asarray(img.pixels[:], dtype=np.float32)
        return px.reshape(h, w, 4)

    def ray_trace_device(self, width, height, samples, buf, origin, rotation, spheres):
        """The generator protocol of ray_trace with everything between the jitter stream and the averaged pixels on
        the GPU(s): a library-owned frame (bhg_frame_*, include/bhgeo.h; _ffi.Frame -- ctypes only, no PyTorch: Blender's
        bundled Python has none).  Per listed device ONE ray-generation launch, ONE trace launch for all samples, ONE
        shade + mean launch; one gather onto the first device; ONE float RGBA array back.  scene.render_devices = N
        shards the image tiles over the first N GPUs of the machine (the reference's render thread is one thread of one
        process, :152-168; its commented-out mp.Pool, :210-216, is where the author wanted the parallelism)."""
        from . import _ffi
        from .raygen import euler_xyz_matrix, python_random_stream
        n_dev = max(1, int(getattr(self, "render_devices", 1) or 1))
        avail = _ffi.device_count()
        if n_dev > avail:
            warnings.warn(f"render_devices = {n_dev} but {avail} GPU(s) are visible: using {max(avail, 1)}", RuntimeWarning)
            n_dev = max(avail, 1)
        first = int(getattr(self.GeoInt.context, "device", 0))
        devices = [(first + i) % max(avail, 1) for i in range(n_dev)]
        if os.environ.get("BHGEO_DEVICES"):
            # an explicit device list, e.g. "2,3" or -- several contexts of ONE GPU, the N > 1 code path on a one-GPU
            # machine (tests) -- "0,0"
            devices = [int(v) for v in os.environ["BHGEO_DEVICES"].split(",")]
        # The frame object outlives the render: Blender makes a new engine instance for every frame of an animation, and
        # creating the frame (the MT19937 stream, the buffers, the first launch of every kernel) costs 100 ms where a render
        # costs 2.4 -- so it is kept at module level and only its camera and scene move (bhg_frame_set_camera / _set_scene).
        # Anything that changes the rays' layout (resolution, samples, seed, device list) makes a new one.
        key = (tuple(devices), int(width), int(height), int(samples), float(self.sampling_seed))
        cam_origin = np.asarray(origin, dtype=np.float64) - self.bh_loc                          # :278
        fr = _DEVICE_FRAMES.get(key)
        if fr is None:
            release_device_frames()
            jitter = python_random_stream(self.sampling_seed, 2 * samples * width * height)      # :189
            fr = _ffi.Frame(devices, width, height, samples, fov_x=self.field_of_view_x, fov_y=self.field_of_view_y,
                            origin=cam_origin, rot=euler_xyz_matrix(rotation), jitter=jitter)
            _DEVICE_FRAMES[key] = fr
            self.device_frame_reused = False
        else:
            fr.set_camera(fov_x=self.field_of_view_x, fov_y=self.field_of_view_y, origin=cam_origin, rot=euler_xyz_matrix(rotation))
            self.device_frame_reused = True
        try:
            sp, lamps = None, None
            if len(spheres) > 0:
                sp = np.array(spheres, dtype=np.float64).reshape(-1, 4)
                sp[:, 0:3] -= self.bh_loc                    # the solver and the shade kernel work BH-centred
                all_lamps = getattr(self, "lamps", [])
                if len(all_lamps) > 4:
                    # (the host path sums over every lamp; the device scene holds four -- say so instead of quietly
                    # rendering a darker image)
                    warnings.warn(f"{len(all_lamps)} lamps in the scene: the device path lights objects with the first 4 "
                                  "(scene.device_shading = 0 sums over all of them)", RuntimeWarning)
                lamps = [[*(np.array(list(l.location), dtype=np.float64) - self.bh_loc), self.LAMP_INTENSITY]
                         for l in all_lamps][:4]
            # The sky image is read (33 MB of float RGBA for a 2k x 1k image) and uploaded to every device only when it is
            # another image than the frame already holds: an animation renders hundreds of frames against one sky.
            sky_id = self._sky_identity()
            sky = None
            if sky_id is None or getattr(fr, "sky_identity", None) != sky_id:
                sky = self._sky_pixels()
                if sky is None:
                    sky = np.zeros((2, 2, 4), dtype=np.float32)     # no sky image: black, as background_hit returns (:369-370)
            self.device_sky_uploaded = sky is not None
            fr.set_scene(sky, spheres=sp, sphere_rgb=None if sp is None else np.ones((len(sp), 3)), lamps=lamps)
            fr.sky_identity = sky_id
            rgba = fr.render(self.GeoInt.params(self.max_integration_step, self.int_depth_curve_end))
            buf[:, :, :] = rgba.reshape(height, width, 4)
            self.last_device_frame = fr.info()
        except BaseException:
            release_device_frames()      # (a frame that failed is not kept)
            raise
        n = samples * height
        for i in range(n):       # progress: the frame is one launch per device, reported after the fact at ray_trace's cadence
            yield (i + 1) / n

    # ---- shading (:366-378), Blender's own texture filter ------------------------------------
    def background_hit(self, direction):
        if self.sky_tex_name not in bpy.data.textures:
            return np.array([0.0, 0.0, 0.0])
        u, v = equirect_uv(np.asarray(direction, dtype=np.float64))
        return np.array(bpy.data.textures[self.sky_tex_name].evaluate((float(u), float(v), 0)).xyz)

    def background_hit_many(self, directions):
        directions = np.asarray(directions, dtype=np.float64).reshape(-1, 3)
        if self.sky_tex_name not in bpy.data.textures:
            return np.zeros(directions.shape)
        tex = bpy.data.textures[self.sky_tex_name]
        u, v = equirect_uv(directions)
        return np.array([tex.evaluate((float(a), float(b), 0)).xyz for a, b in zip(u, v)], dtype=np.float64)


class CUSTOM_RENDER_PT_blackhole(RenderButtonsPanel, Panel):
    bl_label = "Blackhole Settings"
    COMPAT_ENGINES = {RelativisticRenderEngine.bl_idname}

    _ROWS = (("blackhole_obj", "Blackhole"), ("mass", "Mass"), ("max_integration_step", "Max integration step"),
             ("integration_depth", "Integration depth"), ("field_of_view_x", "field_of_view_x"),
             ("field_of_view_y", "field_of_view_y"), ("sampling_seed", "Sampling seed"), ("sky_image", "Sky image"),
             ("mark_x_min", "mark_x_min"), ("mark_x_max", "mark_x_max"), ("mark_y_min", "mark_y_min"),
             ("mark_y_max", "mark_y_max"), ("curved_space_objects", "Objects in curved space (0/1)"),
             ("device_shading", "Shade on the GPU (0/1)"), ("render_devices", "GPUs to render on"))

    def draw(self, context):
        col = self.layout.split().column()
        for prop, text in self._ROWS:
            col.row().prop(bpy.context.scene, prop, text=text)


# scene properties: names and defaults of the reference (:504-517)
PROPS = [
    ("blackhole_obj", bpy.props.PointerProperty(name="blackhole_obj", type=bpy.types.Object)),
    ("mass", bpy.props.FloatProperty(name="Mass", default=0.5)),
    ("max_integration_step", bpy.props.FloatProperty(name="max_integration_step", default=10000)),
    ("integration_depth", bpy.props.FloatProperty(name="integration_depth", default=50)),
    ("sampling_seed", bpy.props.FloatProperty(name="sampling_seed", default=42)),
    ("field_of_view_x", bpy.props.FloatProperty(name="field_of_view_x", default=1)),
    ("field_of_view_y", bpy.props.FloatProperty(name="field_of_view_y", default=1)),
    ("sky_image", bpy.props.StringProperty(name="sky_image", default="", subtype="FILE_PATH")),
    ("mark_y_min", bpy.props.FloatProperty(name="mark_y_min", default=-1.0)),
    ("mark_y_max", bpy.props.FloatProperty(name="mark_y_max", default=-1.0)),
    ("mark_x_min", bpy.props.FloatProperty(name="mark_x_min", default=-1.0)),
    ("mark_x_max", bpy.props.FloatProperty(name="mark_x_max", default=-1.0)),
]

# beyond the reference, all opt-in (module docstring): objects inside the curved region; the frame shaded on the GPU;
# the number of GPUs the device path shards the image over
EXTRA_PROPS = [
    ("curved_space_objects", bpy.props.FloatProperty(name="curved_space_objects", default=0)),
    ("device_shading", bpy.props.FloatProperty(name="device_shading", default=0)),
    ("render_devices", bpy.props.FloatProperty(name="render_devices", default=1)),
]

# the library-owned frame of the device path, kept across renders (see ray_trace_device): at most one
_DEVICE_FRAMES = {}
_SKY_FORCE_REFRESH = False


def invalidate_device_sky():
    """Force the next device render to read the sky image and upload it again, whatever its identity says (a hook for
    scripts that change the image in ways the identity cannot see)."""
    global _SKY_FORCE_REFRESH
    _SKY_FORCE_REFRESH = True


def release_device_frames():
    """Free the GPU buffers the device path keeps between renders (called by unregister(), and when the frame's shape changes)."""
    for fr in list(_DEVICE_FRAMES.values()):
        try:
            fr.close()
        except Exception:
            pass
    _DEVICE_FRAMES.clear()


_EXCLUDED_PANELS = {"VIEWLAYER_PT_filter", "VIEWLAYER_PT_layer_passes"}


def get_panels():
    """Stock panels that declare themselves compatible with BLENDER_RENDER (:520-534)."""
    return [p for p in bpy.types.Panel.__subclasses__()
            if "BLENDER_RENDER" in getattr(p, "COMPAT_ENGINES", ()) and p.__name__ not in _EXCLUDED_PANELS]


def _extra_panels():
    """Eevee sampling / world / material panels the reference also enables (:551-564)."""
    try:
        from bl_ui import properties_material, properties_render, properties_world
    except ImportError:
        return []
    names = ((properties_render, "RENDER_PT_eevee_sampling"), (properties_world, "WORLD_PT_context_world"),
             (properties_material, "EEVEE_MATERIAL_PT_context_material"), (properties_material, "EEVEE_MATERIAL_PT_surface"))
    return [getattr(mod, n) for mod, n in names if hasattr(mod, n)]


def register():
    bpy.utils.register_class(RelativisticRenderEngine)
    bpy.utils.register_class(CUSTOM_RENDER_PT_blackhole)
    for name, prop in PROPS + EXTRA_PROPS:
        setattr(bpy.types.Scene, name, prop)
    for panel in get_panels() + [CUSTOM_RENDER_PT_blackhole] + _extra_panels():
        panel.COMPAT_ENGINES.add(RelativisticRenderEngine.bl_idname)


def unregister():
    release_device_frames()
    bpy.utils.unregister_class(RelativisticRenderEngine)
    bpy.utils.unregister_class(CUSTOM_RENDER_PT_blackhole)
    for name, _ in PROPS + EXTRA_PROPS:
        delattr(bpy.types.Scene, name)
    for panel in get_panels() + [CUSTOM_RENDER_PT_blackhole] + _extra_panels():
        panel.COMPAT_ENGINES.discard(RelativisticRenderEngine.bl_idname)


if __name__ == "__main__":
    register()
