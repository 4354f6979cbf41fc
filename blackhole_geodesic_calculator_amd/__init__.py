"""MI355X-native null-geodesic ray integrator -- drop-in for the curvedpy per-ray solve that
bldevries/blackhole_geodesic_calculator's Blender render engine calls
(raytracer/RelativisticRenderEngine.py:134, :293-294).

Host side is Python over a ctypes C ABI (include/bhgeo.h -> libbhgeo.so, hand-written HIP for
gfx950).  Nothing here computes geodesics on the CPU: without the built library and a GPU the
compute calls raise.
"""
from . import _ffi
from ._ffi import (FLAG_EXITED_SPHERE, FLAG_HIT_DISK, FLAG_HIT_HORIZON, FLAG_MAX_STEPS, FLAG_NAN, FLAG_REACHED_END,
                   FLAG_START_INSIDE, FLAG_STEP_TOO_SMALL, METHOD_DP54, METHOD_RK4, RHS_CHRISTOFFEL,
                   RHS_KERR_BL, RHS_REDUCED, BhgError)
from .integrator import GeodesicIntegratorKerr, GeodesicIntegratorSchwarzschild
from .raygen import camera_directions, python_random_stream

__all__ = [
    "GeodesicIntegratorSchwarzschild", "GeodesicIntegratorKerr", "camera_directions", "python_random_stream", "BhgError",
    "FLAG_HIT_HORIZON", "FLAG_HIT_DISK", "FLAG_START_INSIDE", "FLAG_REACHED_END", "FLAG_EXITED_SPHERE",
    "FLAG_MAX_STEPS", "FLAG_STEP_TOO_SMALL", "FLAG_NAN", "METHOD_DP54", "METHOD_RK4",
    "RHS_CHRISTOFFEL", "RHS_REDUCED", "RHS_KERR_BL",
]
