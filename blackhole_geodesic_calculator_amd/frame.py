"""Frame driver: the reference's per-sample / per-row / per-pixel loops, batched.

Restates the data flow of raytracer/RelativisticRenderEngine.py:172-267 (`ray_trace`):
    random.seed(sampling_seed)                                        :189
    for s in samples: for y in rows: for x in columns:                :195-218
        direction = pinhole + jitter, rotate, normalise               :224-230
        hit, hit_bh, end_dir, end_loc = spacetime_ray_cast(...)       :237   <- the hot path
        sbuf[y, x, 0:3] += black if hit_bh else background_hit(end_dir)   :242-246
      buf[y, :, 0:3] = sbuf[y, :, 0:3] / (s + 1)                      :250
      if y < H-1: buf[y+1, :, 0:3] = 1 - buf[y+1, :, 0:3]             :251-252 (progress cue)
      yield (s*W*H + W*y) / N                                         :261
with ONE batched GPU trace per sample instead of W*H Python calls.  The generator protocol, the
in-place mutation of `buf`, the yielded values and the final image are the same.
"""
from __future__ import annotations

import numpy as np

from .raygen import camera_directions, python_random_stream


def disk_colour(loc, r_in, r_out, texture=None, disk_phase=0.0, disk_mean=0.2, disk_stddev=0.3, disk_intensity=1.0):
    """Colour of thin-disk hits, the Limited engine's checkHitDisk + :300
    (LimitedRelativisticRenderEngine.py:427-436): rgb = texture(texture_x, scale) * intensity with
    scale = (R - R_in)/(R_out - R_in), a Gaussian radial profile and texture_x = (phase + acos(x/R) sign(y))/pi.
    loc [n, 3] are crossing points (BH-centred); texture(u[n], v[n]) -> rgb[n, 3] or None for white.
    (sign(0) is taken as +1 where the reference divides 0/0.)"""
    loc = np.asarray(loc, dtype=np.float64)
    x, y = loc[..., 0], loc[..., 1]
    R = np.sqrt(x * x + y * y)
    scale = (R - r_in) / (r_out - r_in)
    intensity = disk_intensity * np.exp(-((scale - disk_mean) ** 2) / (2 * disk_stddev ** 2)) / np.sqrt(2 * np.pi * disk_stddev)
    texture_x = (disk_phase + np.arccos(np.clip(x / R, -1.0, 1.0)) * np.where(y < 0.0, -1.0, 1.0)) / np.pi
    rgb = np.ones(loc.shape[:-1] + (3,)) if texture is None else np.asarray(texture(texture_x, scale), dtype=np.float64)
    return rgb * intensity[..., None]


def spacetime_ray_cast_batch(integrator, origin, directions, bh_loc=(0.0, 0.0, 0.0), max_step=np.inf,
                             curve_end=50.0, spheres=None, return_objects=False, disk=None):
    """Batched form of spacetime_ray_cast (RelativisticRenderEngine.py:271-313).

    origin: camera world position [3]; directions [..., 3] unit vectors.
    Returns (hit, hit_bh, end_dir, end_loc): hit_bh a bool array, end_dir/end_loc [..., 3].  Without
    `spheres`, hit == False everywhere, as the reference's stub has it (:304-305).  With
    spheres=[[cx, cy, cz, radius], ...] in WORLD coordinates (bh_loc is subtracted here like the
    origin, :278), hit is True where the curve entered a sphere; end_loc is then the entry point (world
    coordinates minus bh_loc, like every end_loc) and end_dir the direction there.  return_objects=True
    appends (normal [..., 3], index [...]) -- the `normal`, `index` a scene.ray_cast hit carries (:441).
    disk=(R_in, R_out): thin disk in the hole's z = 0 plane (LimitedRelativisticRenderEngine.py:283-286); rays
    that end on it have index == -2, hit == False, and end_loc = the crossing point.
    A camera inside the hole gives hit_bh == True for every ray (:311-313); end_dir/end_loc then hold
    the start values.
    """
    bh = np.asarray(bh_loc, dtype=np.float64)
    origin = np.asarray(origin, dtype=np.float64) - bh  # :278
    kw = {} if disk is None else {"disk": (float(disk[0]), float(disk[1]))}
    if spheres is not None and len(spheres):
        sp = np.array(spheres, dtype=np.float64).reshape(-1, 4)
        sp[:, 0:3] -= bh
        out = integrator.trace(directions, origin, max_step=max_step, curve_end=curve_end, spheres=sp, **kw)
    else:
        sp = None
        out = integrator.trace(directions, origin, max_step=max_step, curve_end=curve_end, **kw)
    end = out["ray_end"]
    hit_bh = out["ray_blackhole_hit"].astype(bool)
    if sp is None:
        hit = np.zeros(hit_bh.shape, dtype=bool)
        index = np.full(hit_bh.shape, -1, dtype=np.int8)
        normal = np.zeros(end.shape[:-1] + (3,))
    else:
        index = out["object_id"]
        hit = index >= 0
        c = sp[np.maximum(index, 0)]
        normal = np.where(hit[..., None], (end[..., 0:3] - c[..., 0:3]) / c[..., 3:4], 0.0)
    if disk is not None:
        index = np.where(out["flags"] == 128, np.int8(-2), index).astype(np.int8)   # BHG_FLAG_HIT_DISK
    if return_objects:
        return hit, hit_bh, end[..., 3:6], end[..., 0:3], normal, index
    return hit, hit_bh, end[..., 3:6], end[..., 0:3]


class FrameTracer:
    """One frame = S batched traces of W*H rays each, accumulated exactly like the reference."""

    def __init__(self, integrator, width, height, samples, *, fov_x=1.0, fov_y=1.0, sampling_seed=42.0,
                 origin=(0.0, 0.0, 0.0), rotation_euler=(0.0, 0.0, 0.0), bh_loc=(0.0, 0.0, 0.0),
                 max_step=np.inf, curve_end=50.0, mark=None, spheres=None, object_hit=None, disk=None,
                 disk_hit=None, device_rays=True):
        self.integrator = integrator
        self.width, self.height, self.samples = int(width), int(height), int(samples)
        self.fov_x, self.fov_y = float(fov_x), float(fov_y)
        self.sampling_seed = sampling_seed
        self.origin = np.asarray(origin, dtype=np.float64)
        self.rotation_euler = tuple(float(e) for e in rotation_euler)
        self.bh_loc = np.asarray(bh_loc, dtype=np.float64)
        self.max_step, self.curve_end = max_step, curve_end
        self.mark = mark  # (y_min, y_max, x_min, x_max), inclusive, or None
        # objects in the curved region (the stub at :304-305 filled in): spheres in world coordinates and
        # object_hit(loc[n,3], normal[n,3], index[n]) -> rgb[n,3], the vectorised spacetime_hit (:240, :317)
        self.spheres = spheres
        self.object_hit = object_hit
        # thin disk (Limited engine, :283-300): disk=(R_in, R_out), disk_hit(loc[n,3]) -> rgb[n,3]
        # (default: disk_colour with a white texture)
        self.disk = disk
        self.disk_hit = disk_hit
        self.last_counters = None
        # device_rays: generate the rays on the device and keep them there (integrator.ray_set) when the integrator can,
        # instead of generating them on the host and uploading them for every sample; only end_dir + flags (and
        # end_loc / object ids when a disk or objects are set) come back.  Bit-identical rays for an unrotated
        # camera (tests), within an ulp before normalisation otherwise.
        self.device_rays = bool(device_rays)
        self._rays = None
        self._rays_key = None

    # the jitter stream depends only on (seed, window, S): cache it across frames (:189 re-seeds
    # identically on every render())
    def directions(self):
        return camera_directions(self.width, self.height, self.samples, self.fov_x, self.fov_y,
                                 self.sampling_seed, self.rotation_euler, self.mark)

    def _window(self):
        W, H = self.width, self.height
        y_min, y_max, x_min, x_max = self.mark if self.mark is not None else (0, H, 0, W)
        rows = [y for y in range(H) if y_min <= y <= y_max]
        cols = np.array([x for x in range(W) if x_min <= x <= x_max], dtype=np.int64)
        return rows, cols

    def _resident_rays(self, rows, cols):
        """The frame's ray set on the device, rebuilt only when the camera, the window or the seed change (the
        engine re-seeds identically on every render(), :189: a static camera traces the same rays every frame)."""
        key = (self.width, self.height, self.samples, self.fov_x, self.fov_y, self.sampling_seed, self.rotation_euler,
               tuple(self.origin - self.bh_loc), self.mark)
        if self._rays is None or self._rays_key != key:
            P = len(rows) * len(cols)
            stream = python_random_stream(self.sampling_seed, 2 * self.samples * P)   # draws inside the window only (:219)
            full = self.mark is None
            pix = None if full else (np.asarray(rows, dtype=np.int64)[:, None] * self.width + cols[None, :]).reshape(-1)
            if self._rays is not None:
                self._rays.close()
            self._rays = self.integrator.ray_set(self.width, self.height, self.samples, self.fov_x, self.fov_y,
                                                 self.origin - self.bh_loc, self.rotation_euler, jitter=stream,
                                                 jitter_is_compact=not full, pixels=pix)
            self._rays_key = key
        return self._rays

    def _cast_sample_resident(self, rays, s, shape):
        """Sample s of the resident ray set through the solver: the same six values spacetime_ray_cast_batch returns."""
        P = shape[0] * shape[1]
        want = ["end_dir", "flags"]
        sp = None
        if self.spheres is not None and len(self.spheres):
            sp = np.array(self.spheres, dtype=np.float64).reshape(-1, 4)
            sp[:, 0:3] -= self.bh_loc
            want += ["end_loc", "object_id"]
        elif self.disk is not None:
            want += ["end_loc"]
        kw = {} if self.disk is None else {"disk": (float(self.disk[0]), float(self.disk[1]))}
        out = self.integrator.trace_rays(rays, first=s * P, n=P, max_step=self.max_step, curve_end=self.curve_end,
                                         spheres=sp, want=tuple(want), **kw)
        flags = out["flags"].reshape(shape)
        end_dir = out["end_dir"].reshape(shape + (3,))
        end_loc = out["end_loc"].reshape(shape + (3,)) if "end_loc" in out else np.zeros(shape + (3,))
        hit_bh = (flags & 1) != 0
        if sp is None:
            hit = np.zeros(shape, dtype=bool)
            index = np.full(shape, -1, dtype=np.int8)
            normal = np.zeros(shape + (3,))
        else:
            index = out["object_id"].reshape(shape)
            hit = index >= 0
            c = sp[np.maximum(index, 0)]
            normal = np.where(hit[..., None], (end_loc - c[..., 0:3]) / c[..., 3:4], 0.0)
        if self.disk is not None:
            index = np.where(flags == 128, np.int8(-2), index).astype(np.int8)   # BHG_FLAG_HIT_DISK
        return hit, hit_bh, end_dir, end_loc, normal, index

    def ray_trace(self, buf, background_hit):
        """Generator with the reference's protocol: mutates buf[H, W, 4] in place, yields the
        progress fraction once per rendered row.  `background_hit(directions[n,3]) -> rgb[n,3]`
        shades the escaping rays (vectorised counterpart of :366-378)."""
        W, H, S = self.width, self.height, self.samples
        N = S * W * H
        y_min, y_max, x_min, x_max = self.mark if self.mark is not None else (0, H, 0, W)
        rows = [y for y in range(H) if y_min <= y <= y_max]
        cols = np.array([x for x in range(W) if x_min <= x <= x_max], dtype=np.int64)
        sbuf = np.zeros((H, W, 4))
        resident = self.device_rays and hasattr(self.integrator, "ray_set") and bool(rows) and len(cols) > 0
        if resident:
            rset = self._resident_rays(rows, cols)
        else:
            dirs = self.directions()  # [S, H, W, 3], NaN outside the window
        steps = 0
        rays = 0
        colour = np.zeros((len(rows), len(cols), 3))
        for s in range(S):
            if rows and len(cols):
                if resident:
                    hit, hit_bh, end_dir, end_loc, normal, index = self._cast_sample_resident(rset, s, (len(rows), len(cols)))
                else:
                    d = dirs[s][np.ix_(rows, cols)]  # [R, C, 3]
                    hit, hit_bh, end_dir, end_loc, normal, index = spacetime_ray_cast_batch(
                        self.integrator, self.origin, d, self.bh_loc, self.max_step, self.curve_end,
                        spheres=self.spheres, return_objects=True, disk=self.disk)
                colour = np.zeros(hit_bh.shape + (3,))
                on_disk = index == -2
                esc = ~hit_bh & ~hit & ~on_disk   # :239-246: hit -> spacetime_hit, hit_bh -> black, else background
                if on_disk.any():
                    shade = self.disk_hit or (lambda loc: disk_colour(loc, self.disk[0], self.disk[1]))
                    colour[on_disk] = np.asarray(shade(end_loc[on_disk]), dtype=np.float64)
                if esc.any():
                    colour[esc] = np.asarray(background_hit(end_dir[esc]), dtype=np.float64)
                if hit.any() and self.object_hit is not None:
                    colour[hit] = np.asarray(self.object_hit(end_loc[hit] + self.bh_loc, normal[hit], index[hit]),
                                             dtype=np.float64)
                rays += hit_bh.size
            for ri, y in enumerate(rows):
                sbuf[y, cols, 0:3] += colour[ri]
                buf[y, :, 0:3] = sbuf[y, :, 0:3] / (s + 1)
                if y < H - 1:
                    buf[y + 1, :, 0:3] = 1 - buf[y + 1, :, 0:3]
                yield (s * W * H + W * y) / N
        self.last_counters = {"rays": rays}


def equirect_uv(direction, normalise=True):
    """(u, v) texture coordinates of background_hit (RelativisticRenderEngine.py:373-375):
    theta = 1 - acos(d_z)/pi, phi = atan2(d_y, d_x)/pi, evaluate((-phi, 2*theta - 1, 0)).

    The solver's exit directions are not unit vectors in general; the Cam edition renormalises
    before the lookup (RelativisticRenderEngineCamEdition.py:433-437) and so does this function by
    default (the main engine does not and gets NaN for |d_z| > 1)."""
    d = np.asarray(direction, dtype=np.float64)
    if normalise:
        d = d / np.sqrt((d * d).sum(-1))[..., None]
    theta = 1 - np.arccos(d[..., 2]) / np.pi
    phi = np.arctan2(d[..., 1], d[..., 0]) / np.pi
    return -phi, 2 * theta - 1
