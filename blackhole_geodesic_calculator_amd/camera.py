"""Pre-traced camera: the batched per-frame result arrays the Cam edition consumes.

Mirrors the surface of curvedpy.cameras.camera.RelativisticCamera as the reference uses it
(raytracer/RelativisticRenderEngineCamEdition.py:11, :206-215, :225, :228):
    cam = RelativisticCamera(resolution=[H, W], field_of_view=[fx, fy], a=..., camera_location=...,
                             camera_rotation_euler=...)          (commented kwargs, :207-213)
    cam.run(verbose=True, verbose_lvl=100)                       (:214)
    cam.load(path)                                               (:215)
    cam.ray_blackhole_hit[iy, ix] == 1                           (:225)
    cam.ray_end[iy, ix, 3:6]                                     (:228)   (0:3 position, 3:6 direction)
One GPU launch traces the whole frame; results are cacheable per (camera, hole, resolution)
because the engine's rays do not change between frames of a static camera.
"""
from __future__ import annotations

import pickle

import numpy as np

from .integrator import GeodesicIntegratorKerr, GeodesicIntegratorSchwarzschild
from .raygen import euler_xyz_matrix


class RelativisticCamera:
    def __init__(self, resolution=(64, 64), field_of_view=(0.6, 0.6), a=0.0, M=0.5,
                 camera_location=(1e-4, 0.0, 30.0), camera_rotation_euler=(0.0, 0.0, 0.0),
                 max_step=np.inf, curve_end=50.0, verbose=False, integrator=None, device=0, **integrator_kw):
        if not abs(a) < 1.0:
            raise ValueError("|a| (dimensionless spin a/M, CamEdition.py:210) must be < 1")
        self.resolution = [int(resolution[0]), int(resolution[1])]  # [H, W]
        self.field_of_view = [float(field_of_view[0]), float(field_of_view[1])]  # [x, y]
        self.a, self.M = float(a), float(M)
        self.camera_location = np.asarray(camera_location, dtype=np.float64)
        self.camera_rotation_euler = tuple(float(e) for e in camera_rotation_euler)
        self.max_step, self.curve_end = max_step, curve_end
        self.verbose = verbose
        self._integrator = integrator
        self._device, self._integrator_kw = device, integrator_kw
        self.ray_end = None
        self.ray_blackhole_hit = None
        self.results = None

    def pixel_directions(self):
        """Pixel-centre pinhole directions [H, W, 3], same formula as the engine's ray generator
        (raytracer/RelativisticRenderEngine.py:224-230) without the multisample jitter."""
        H, W = self.resolution
        aspect = H / W
        xs = self.field_of_view[0] * (np.arange(W) - int(W / 2)) / W
        ys = self.field_of_view[1] * (np.arange(H) - int(H / 2)) / H * aspect
        d = np.empty((H, W, 3))
        d[..., 0] = xs[None, :]
        d[..., 1] = ys[:, None]
        d[..., 2] = -1.0
        rot = euler_xyz_matrix(self.camera_rotation_euler)
        if not np.array_equal(rot, np.eye(3)):
            d = d @ rot.T
        return d / np.sqrt((d * d).sum(-1))[..., None]

    def run(self, verbose=False, verbose_lvl=0):
        if self._integrator is not None:
            gi = self._integrator
        elif self.a != 0.0:
            gi = GeodesicIntegratorKerr(mass=self.M, a=self.a, device=self._device, **self._integrator_kw)
        else:
            gi = GeodesicIntegratorSchwarzschild(mass=self.M, time_like=False, verbose=False, device=self._device,
                                                 **self._integrator_kw)
        H, W = self.resolution
        if hasattr(gi, "ray_set"):
            # pixel-centre rays generated on the device (jitter None = u 1/2): the directions never cross PCIe
            rays = gi.ray_set(W, H, 1, self.field_of_view[0], self.field_of_view[1], self.camera_location,
                              self.camera_rotation_euler, jitter=None)
            o = gi.trace_rays(rays, max_step=self.max_step, curve_end=self.curve_end,
                              want=("end", "flags", "n_steps", "n_accepted"))
            rays.close()
            out = {"ray_end": o["end"].reshape(H, W, 6), "flags": o["flags"].reshape(H, W),
                   "ray_blackhole_hit": ((o["flags"] & 1) != 0).astype(np.uint8).reshape(H, W),
                   "n_steps": o["n_steps"].reshape(H, W), "n_accepted": o["n_accepted"].reshape(H, W)}
        else:
            out = gi.trace(self.pixel_directions(), self.camera_location, max_step=self.max_step,
                           curve_end=self.curve_end)
        self.ray_end = out["ray_end"]
        self.ray_blackhole_hit = out["ray_blackhole_hit"]
        self.results = {"flags": out["flags"], "n_steps": out["n_steps"], "n_accepted": out["n_accepted"]}
        if verbose or self.verbose:
            print(f"RelativisticCamera: traced {self.ray_blackhole_hit.size} rays, "
                  f"{int(self.ray_blackhole_hit.sum())} end on the horizon")
        return self

    # pickle persistence, as the Cam edition expects (pkl_file property, CamEdition.py:79, :215)
    _STATE = ("resolution", "field_of_view", "a", "M", "camera_location", "camera_rotation_euler", "max_step",
              "curve_end", "ray_end", "ray_blackhole_hit", "results")

    def save(self, path):
        with open(path, "wb") as f:
            pickle.dump({k: getattr(self, k) for k in self._STATE}, f)

    def load(self, path):
        with open(path, "rb") as f:
            st = pickle.load(f)
        for k in self._STATE:
            setattr(self, k, st[k])
        return self
