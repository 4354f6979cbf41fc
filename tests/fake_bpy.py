"""A minimal stand-in for Blender's bpy / bl_ui modules: exactly the surface the render-engine
add-on touches.  Test infrastructure; Blender itself is not available."""
import sys
import types

import numpy as np


class _Vec(list):
    @property
    def xyz(self):
        return self[:3]


class FakeTexture:
    def __init__(self, name, kind):
        self.name, self.kind, self.image = name, kind, None

    def evaluate(self, uvw):
        """A smooth synthetic sky so that tests can predict colours: rgb = f(u, v)."""
        u, v = float(uvw[0]), float(uvw[1])
        return _Vec([0.5 + 0.5 * np.sin(np.pi * u), 0.5 + 0.5 * v, 0.25 + 0.25 * np.cos(2 * np.pi * u), 1.0])


class FakeImage:
    """bpy.types.Image as far as the add-on reads it: size (w, h) and pixels, a flat float RGBA sequence, rows
    bottom-up.  The content is a deterministic synthetic sky (tests predict colours from `array`)."""

    def __init__(self, path, width=64, height=32, seed=11):
        self.filepath = path
        self.size = (width, height)
        rng = np.random.default_rng(seed)
        v, u = np.meshgrid(np.linspace(0, 1, height), np.linspace(0, 1, width), indexing="ij")
        img = np.stack([0.2 + 0.6 * u, 0.1 + 0.8 * v, 0.5 + 0.4 * np.sin(6.0 * u + 3.0 * v), np.ones_like(u)], -1)
        img[..., :3] += 0.05 * rng.random((height, width, 3))
        self.array = img.astype(np.float32)          # [h, w, 4], row 0 = bottom
        self.pixels = self.array.reshape(-1).tolist()


class _Store(dict):
    def load(self, path):
        import os
        self[os.path.basename(path)] = FakeImage(path)

    def new(self, name, kind):
        self[name] = FakeTexture(name, kind)
        return self[name]


class _Matrix:
    def __init__(self, translation, euler):
        self.translation, self._euler = list(translation), tuple(euler)

    def to_euler(self):
        return self._euler


class FakeResult:
    def __init__(self):
        self.layers = [types.SimpleNamespace(passes={"Combined": types.SimpleNamespace(rect=None)})]


class RenderEngine:
    is_preview = False

    def __init__(self):
        self.progress, self.updates, self.ended = [], 0, 0
        self.result = None

    def begin_result(self, x, y, w, h):
        self.result = FakeResult()
        self.result.size = (w, h)
        return self.result

    def update_result(self, result):
        self.updates += 1

    def update_progress(self, frac):
        self.progress.append(frac)

    def end_result(self, result):
        self.ended += 1


class Panel:
    COMPAT_ENGINES = set()


class _StockPanel(Panel):
    COMPAT_ENGINES = {"BLENDER_RENDER"}


def install(width=16, height=12, samples=2, camera=(1e-4, 0.0, 30.0), euler=(0.0, 0.0, 0.0), **scene_props):
    """Create fake bpy / bl_ui modules in sys.modules and return (bpy, depsgraph)."""
    bpy = types.ModuleType("bpy")
    bpy.types = types.SimpleNamespace(RenderEngine=RenderEngine, Panel=Panel, Object=object,
                                      Scene=type("Scene", (), {}))
    bpy.props = types.SimpleNamespace(
        PointerProperty=lambda **kw: ("pointer", kw), FloatProperty=lambda **kw: ("float", kw),
        StringProperty=lambda **kw: ("string", kw))
    registered = []
    bpy.utils = types.SimpleNamespace(register_class=registered.append, unregister_class=registered.remove)
    bpy._registered = registered
    bpy.data = types.SimpleNamespace(images=_Store(), textures=_Store(),
                                     scenes={"Scene": types.SimpleNamespace(eevee=types.SimpleNamespace(taa_render_samples=samples))})
    props = dict(max_integration_step=-1, sampling_seed=42.0, integration_depth=50, field_of_view_x=0.6,
                 field_of_view_y=0.6, sky_image="/tmp/sky.png", mass=0.5, blackhole_obj=None, mark_y_min=-1,
                 mark_y_max=-1, mark_x_min=-1, mark_x_max=-1)
    props.update(scene_props)
    scene = types.SimpleNamespace(
        render=types.SimpleNamespace(resolution_x=width, resolution_y=height, resolution_percentage=100),
        camera=types.SimpleNamespace(matrix_world=_Matrix(camera, euler)), objects=[], **props)
    depsgraph = types.SimpleNamespace(scene=scene, update=lambda: None)
    bpy.context = types.SimpleNamespace(evaluated_depsgraph_get=lambda: depsgraph, scene=scene)
    bl_ui = types.ModuleType("bl_ui")
    pr = types.ModuleType("bl_ui.properties_render")
    pr.RenderButtonsPanel = type("RenderButtonsPanel", (), {})
    pr.RENDER_PT_eevee_sampling = type("RENDER_PT_eevee_sampling", (Panel,), {"COMPAT_ENGINES": set()})
    pw = types.ModuleType("bl_ui.properties_world")
    pw.WORLD_PT_context_world = type("WORLD_PT_context_world", (Panel,), {"COMPAT_ENGINES": set()})
    pm = types.ModuleType("bl_ui.properties_material")
    pm.EEVEE_MATERIAL_PT_context_material = type("EEVEE_MATERIAL_PT_context_material", (Panel,), {"COMPAT_ENGINES": set()})
    pm.EEVEE_MATERIAL_PT_surface = type("EEVEE_MATERIAL_PT_surface", (Panel,), {"COMPAT_ENGINES": set()})
    bl_ui.properties_render, bl_ui.properties_world, bl_ui.properties_material = pr, pw, pm
    for name, mod in (("bpy", bpy), ("bpy.types", bpy.types), ("bl_ui", bl_ui), ("bl_ui.properties_render", pr),
                      ("bl_ui.properties_world", pw), ("bl_ui.properties_material", pm)):
        sys.modules[name] = mod
    sys.modules.pop("blackhole_geodesic_calculator_amd.blender_addon", None)
    return bpy, depsgraph
