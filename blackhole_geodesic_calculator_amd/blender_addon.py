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
    at most 8 (the solver's limit), the nearest to the hole, with a warning when more are dropped.
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
                layer.rect = buf.reshape(-1, 4).tolist()
                self.update_result(result)
        layer.rect = buf.reshape(-1, 4).tolist()
        self.update_result(result)
        self.end_result(result)

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

    def spacetime_hit_many(self, loc, normal, index, intensity=10):
        """Vectorised spacetime_hit (:317-363): white Lambert lamps with 1/d^2 falloff; no shadow rays (there
        is no flat-space scene to cast them into here) and n.l clamped at 0."""
        color = np.zeros(loc.shape)
        for lamp in getattr(self, "lamps", []):
            light_vec = np.array(list(lamp.location), dtype=np.float64) - loc
            light_dist = (light_vec * light_vec).sum(-1)
            light_dir = light_vec / np.sqrt(light_dist)[:, None]
            ndl = np.maximum((normal * light_dir).sum(-1), 0.0)
            color += (intensity * intensity * ndl / light_dist)[:, None]
        return color

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
             ("mark_y_max", "mark_y_max"), ("curved_space_objects", "Objects in curved space (0/1)"))

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

# beyond the reference: opt-in switch for objects inside the curved region (module docstring)
EXTRA_PROPS = [
    ("curved_space_objects", bpy.props.FloatProperty(name="curved_space_objects", default=0)),
]

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
    bpy.utils.unregister_class(RelativisticRenderEngine)
    bpy.utils.unregister_class(CUSTOM_RENDER_PT_blackhole)
    for name, _ in PROPS + EXTRA_PROPS:
        delattr(bpy.types.Scene, name)
    for panel in get_panels() + [CUSTOM_RENDER_PT_blackhole] + _extra_panels():
        panel.COMPAT_ENGINES.discard(RelativisticRenderEngine.bl_idname)


if __name__ == "__main__":
    register()
