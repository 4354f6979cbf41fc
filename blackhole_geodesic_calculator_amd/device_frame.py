"""Device-resident frame pipeline: jitter stream -> rays -> geodesics -> shaded, sample-averaged
pixels, without a host round trip in between.

Covers, on the GPU, the reference's ray generation (raytracer/RelativisticRenderEngine.py:185-230),
the per-ray solve (:293-294), the sky lookup (:366-378, with the build's own bilinear filter in
place of Blender's) and the multisample mean (:242-250).  PyTorch is used only as plumbing for
device memory and the stream; all kernels are libbhgeo's.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _ffi
from .raygen import euler_xyz_matrix, python_random_stream
from .sky import synthetic_sky  # noqa: F401  (re-exported: tests and bench import it from here)


def _copy_params(p: _ffi.Params) -> _ffi.Params:
    q = _ffi.Params()
    import ctypes
    ctypes.memmove(ctypes.byref(q), ctypes.byref(p), ctypes.sizeof(_ffi.Params))
    return q


class DeviceFrame:
    """All buffers of one frame shard on one GPU.

    pixels: flat pixel ids (y*W + x) this GPU owns (e.g. dist.rank_pixels); None = whole frame.
    Rays are laid out [S][P]: ray s*P + p is sample s of pixel pixels[p].
    """

    def __init__(self, ctx: _ffi.Context, width, height, samples, *, fov_x=1.0, fov_y=1.0, sampling_seed=42.0,
                 origin=(1e-4, 0.0, 30.0), rotation_euler=(0.0, 0.0, 0.0), bh_loc=(0.0, 0.0, 0.0), pixels=None,
                 jitter=None, device=None, buffers=None, directions_only=False):
        """directions_only: a frame without disk or objects reads only the exit DIRECTIONS of its rays (the sky
        lookup, :366-378) -- trace() then has the kernel write those alone (d_dir [n, 3], bhg_trace_dir_device: half
        the bytes written per ray and read by the shade kernel; d_end is not filled) and shade() / shade_f32() read
        them.  With a disk or objects set the frame falls back to whole end records by itself."""
        self.ctx = ctx
        self.directions_only = bool(directions_only)
        self.d_dir = None
        self._dir_traced = False
        self._traced = None      # what the last trace wrote: None (nothing usable), "dir" or "end"
        self.W, self.H, self.S = int(width), int(height), int(samples)
        self.fov_x, self.fov_y = float(fov_x), float(fov_y)
        self.origin = np.asarray(origin, dtype=np.float64) - np.asarray(bh_loc, dtype=np.float64)  # :278
        self.rot = euler_xyz_matrix(rotation_euler)
        self.dev = torch.device("cuda", ctx.device) if device is None else device
        if jitter is None:
            jitter = python_random_stream(sampling_seed, 2 * self.S * self.W * self.H)  # :189
        self.d_jitter = torch.as_tensor(np.asarray(jitter, dtype=np.float64)).to(self.dev)
        if pixels is None:
            self.d_pixels = None
            self.P = self.W * self.H
        else:
            self.d_pixels = torch.as_tensor(np.asarray(pixels, dtype=np.int64)).to(self.dev)
            self.P = int(self.d_pixels.numel())
        n = self.S * self.P
        self.n = n
        if buffers is not None:
            # views into a caller-owned block shared by several frames (FrameBatch): one trace call for all
            self.d_k0, self.d_end, self.d_flags, self.d_steps, self.d_acc = buffers
            assert self.d_k0.shape == (n, 3) and (self.d_end is None or self.d_end.shape == (n, 6)) and self.d_flags.numel() == n
        else:
            self.d_k0 = torch.empty((n, 3), dtype=torch.float64, device=self.dev)
            # whole end records: allocated when a full-record trace is first issued (a direction-only frame never needs them)
            self.d_end = None if self.directions_only else torch.empty((n, 6), dtype=torch.float64, device=self.dev)
            self.d_flags = torch.empty(n, dtype=torch.uint8, device=self.dev)
            self.d_steps = torch.empty(n, dtype=torch.int32, device=self.dev)
            self.d_acc = torch.empty(n, dtype=torch.int32, device=self.dev)
        self.d_rgba = torch.empty((self.P, 4), dtype=torch.float64, device=self.dev)
        self.d_sky = None
        self.sky_wh = (0, 0)
        # scene beyond the sky (set_disk / set_objects): thin disk, object spheres with lamps
        self.disk = None
        self.disk_profile = dict(disk_phase=0.0, disk_mean=0.2, disk_stddev=0.3, disk_intensity=1.0)
        self.d_disk_tex = None
        self.spheres = None
        self.sphere_rgb = None
        self.lamps = None
        self.d_obj = None

    def _stream(self):
        return torch.cuda.current_stream(self.dev).cuda_stream

    def set_sky(self, sky_rgba_f32):
        """Equirectangular sky [TH, TW, 4] float32."""
        sky = np.ascontiguousarray(sky_rgba_f32, dtype=np.float32)
        assert sky.ndim == 3 and sky.shape[2] == 4
        self.d_sky = torch.as_tensor(sky).to(self.dev)
        self.sky_wh = (sky.shape[1], sky.shape[0])

    def set_disk(self, r_in, r_out, texture_rgba_f32=None, **profile):
        """Thin disk in z = 0 between r_in and r_out (LimitedRelativisticRenderEngine.py:283-300, :413-438);
        profile: disk_phase, disk_mean, disk_stddev, disk_intensity (:55-58).  The trace must be run with the
        same radii in its params (make_params(disk_r_in=..., disk_r_out=...))."""
        self.disk = (float(r_in), float(r_out))
        self._traced = None      # the scene changed: what the last trace wrote no longer matches it
        self.disk_profile.update(profile)
        if texture_rgba_f32 is not None:
            tex = np.ascontiguousarray(texture_rgba_f32, dtype=np.float32)
            assert tex.ndim == 3 and tex.shape[2] == 4
            self.d_disk_tex = torch.as_tensor(tex).to(self.dev)

    def set_objects(self, spheres, sphere_rgb=None, lamps=None):
        """Object spheres [[cx, cy, cz, radius]] (BH-centred), their colours and the point lamps
        [[x, y, z, intensity]] that light them.  Cheap to call per frame (an animation moves them): the
        values travel as kernel arguments."""
        spheres = np.asarray(spheres, dtype=np.float64).reshape(-1, 4)
        if self.spheres is None or not np.array_equal(spheres, self.spheres):
            # the geometry the rays were traced against is gone (spheres moved, resized, appeared or went away): what
            # the last trace wrote -- end states, flags, object ids -- no longer fits, and shade() says so
            self._traced = None
        self.spheres = spheres
        self.sphere_rgb = sphere_rgb
        self.lamps = lamps
        if self.d_obj is None:
            self.d_obj = torch.empty(self.n, dtype=torch.int8, device=self.dev)

    def pixel_cost(self):
        """Attempted steps of the last trace summed over the samples of each pixel ([P], order of `pixels`): the
        measured cost dist.measured_tile_cost() orders and deals the tiles by."""
        return self.d_steps.view(self.S, self.P).to(torch.int64).sum(0)

    def generate_rays(self):
        self._traced = None      # new rays: results of an earlier trace belong to the old ones
        self.ctx.raygen_device(self.W, self.H, self.S, self.fov_x, self.fov_y, self.d_jitter.data_ptr(),
                               self.d_k0.data_ptr(), self.P,
                               d_pixels=0 if self.d_pixels is None else self.d_pixels.data_ptr(),
                               rot=self.rot, stream=self._stream())

    def trace(self, params: _ffi.Params):
        # work-order hint: the rays are S blocks of P (sample-major); lets the library start all samples of a
        # region together (the pixels of a shard are usually sorted longest-first, dist.rank_pixels(tile_cost=))
        if params.order_blocks == 0 and self.S > 1:
            params = _copy_params(params)
            params.order_blocks = self.S
        has_obj = self.spheres is not None and len(self.spheres) > 0
        self._dir_traced = self.directions_only and not has_obj and self.disk is None and not (params.disk_r_out > 0.0)
        self._traced = "dir" if self._dir_traced else "end"
        if self._dir_traced:
            if self.d_dir is None:
                self.d_dir = torch.empty((self.n, 3), dtype=torch.float64, device=self.dev)
            self.ctx.trace_dir_device(params, self.n, self.d_k0.data_ptr(), self.d_dir.data_ptr(), x0_shared=self.origin,
                                      d_flags=self.d_flags.data_ptr(), d_n_steps=self.d_steps.data_ptr(),
                                      d_n_accepted=self.d_acc.data_ptr(), stream=self._stream())
            return
        if self.d_end is None:
            self.d_end = torch.empty((self.n, 6), dtype=torch.float64, device=self.dev)
        self.ctx.trace_device(params, self.n, self.d_k0.data_ptr(), self.d_end.data_ptr(), x0_shared=self.origin,
                              d_flags=self.d_flags.data_ptr(), d_n_steps=self.d_steps.data_ptr(),
                              d_n_accepted=self.d_acc.data_ptr(), stream=self._stream(),
                              spheres=self.spheres if has_obj else None,
                              d_object_id=self.d_obj.data_ptr() if has_obj else 0)

    def scene(self):
        tex = self.d_disk_tex
        return _ffi.make_scene(self.d_sky.data_ptr(), self.sky_wh[0], self.sky_wh[1],
                               d_disk_tex=0 if tex is None else tex.data_ptr(),
                               disk_w=0 if tex is None else tex.shape[1], disk_h=0 if tex is None else tex.shape[0],
                               disk=self.disk, spheres=self.spheres, sphere_rgb=self.sphere_rgb, lamps=self.lamps,
                               **self.disk_profile)

    def _shade_form(self):
        """ONE predicate for both shade paths: "dir" (exit directions, sky only) or "end" (whole records, any scene);
        raises when the scene was changed after the last trace so that its output no longer fits."""
        if self.d_sky is None:
            raise RuntimeError("set_sky() first")
        traced = self._traced
        if traced is None:
            raise RuntimeError("trace() first (the scene changed since the last trace, or nothing was traced yet)")
        has_scene = self.disk is not None or (self.spheres is not None and len(self.spheres) > 0)
        if traced == "dir" and has_scene:
            raise RuntimeError("the last trace wrote exit directions only, but the frame now has a disk / objects: trace() again")
        return traced

    def shade(self):
        form = self._shade_form()
        if self.disk is not None or (self.spheres is not None and len(self.spheres) > 0):
            self.ctx.shade_scene_device(self.d_end.data_ptr(), self.d_flags.data_ptr(), self.P, self.S, self.scene(),
                                        self.d_rgba.data_ptr(),
                                        d_object_id=0 if self.d_obj is None else self.d_obj.data_ptr(),
                                        stream=self._stream())
            return self.d_rgba
        if form == "dir":
            self.ctx.shade_dir_device(self.d_dir.data_ptr(), self.d_flags.data_ptr(), self.P, self.S, self.d_sky.data_ptr(),
                                      self.sky_wh[0], self.sky_wh[1], d_rgba=self.d_rgba.data_ptr(), stream=self._stream())
            return self.d_rgba
        self.ctx.shade_device(self.d_end.data_ptr(), self.d_flags.data_ptr(), self.P, self.S, self.d_sky.data_ptr(),
                              self.sky_wh[0], self.sky_wh[1], self.d_rgba.data_ptr(), stream=self._stream())
        return self.d_rgba

    def shade_f32(self, out, scatter=None):
        """Shade + sample mean written as float32 RGBA into `out` ([P, 4], or [H*W, 4] with scatter = this shard's
        flat pixel ids): what layer.rect takes, without the fp64 intermediate."""
        form = self._shade_form()
        assert out.dtype == torch.float32 and out.is_contiguous()
        if form == "dir":
            self.ctx.shade_dir_device(self.d_dir.data_ptr(), self.d_flags.data_ptr(), self.P, self.S, self.d_sky.data_ptr(),
                                      self.sky_wh[0], self.sky_wh[1], d_rgba_f32=out.data_ptr(),
                                      d_scatter=0 if scatter is None else scatter.data_ptr(), stream=self._stream())
            return out
        self.ctx.shade_scene_f32_device(self.d_end.data_ptr(), self.d_flags.data_ptr(), self.P, self.S, self.scene(),
                                        out.data_ptr(), d_object_id=0 if self.d_obj is None else self.d_obj.data_ptr(),
                                        d_scatter=0 if scatter is None else scatter.data_ptr(), stream=self._stream())
        return out

    def render(self, params: _ffi.Params, regenerate_rays=False):
        """rays (cached: the engine re-seeds identically every frame) -> trace -> shade."""
        if regenerate_rays or not getattr(self, "_rays_ready", False):
            self.generate_rays()
            self._rays_ready = True
        self.trace(params)
        return self.shade()


class FrameBatch:
    """Several cameras' frames of the same size traced by ONE library call (per-ray origins, bhg_trace_device's
    d_x0): e.g. the five inclinations of a disk study.  Each member is a DeviceFrame whose ray buffers are views
    into one block; shading stays per frame."""

    def __init__(self, ctx: _ffi.Context, cameras, width, height, samples, *, pixels=None, jitter=None, device=None,
                 **frame_kw):
        """cameras: list of dicts with origin=, rotation_euler= (and optionally bh_loc=)."""
        self.ctx = ctx
        dev = torch.device("cuda", ctx.device) if device is None else device
        W, H, S = int(width), int(height), int(samples)
        P = W * H if pixels is None else len(pixels)
        n1 = S * P
        m = len(cameras)
        if jitter is None:
            jitter = python_random_stream(frame_kw.get("sampling_seed", 42.0), 2 * S * W * H)
        self.d_k0 = torch.empty((m * n1, 3), dtype=torch.float64, device=dev)
        self.d_x0 = torch.empty((m * n1, 3), dtype=torch.float64, device=dev)
        self.d_end = torch.empty((m * n1, 6), dtype=torch.float64, device=dev)
        self.d_flags = torch.empty(m * n1, dtype=torch.uint8, device=dev)
        self.d_steps = torch.empty(m * n1, dtype=torch.int32, device=dev)
        self.d_acc = torch.empty(m * n1, dtype=torch.int32, device=dev)
        self.frames = []
        for j, cam in enumerate(cameras):
            sl = slice(j * n1, (j + 1) * n1)
            f = DeviceFrame(ctx, W, H, S, pixels=pixels, jitter=jitter, device=dev,
                            buffers=(self.d_k0[sl], self.d_end[sl], self.d_flags[sl], self.d_steps[sl], self.d_acc[sl]),
                            **cam, **frame_kw)
            self.d_x0[sl] = torch.as_tensor(f.origin, device=dev)
            self.frames.append(f)
        self.n = m * n1
        self.dev = dev

    def generate_rays(self):
        for f in self.frames:
            f.generate_rays()

    def trace(self, params: _ffi.Params):
        # work-order hint: the rays are (cameras x samples) equal blocks of P rays, each block's pixels in the same
        # (usually longest-first) order: lets the library start the expensive regions of ALL frames first
        nb = len(self.frames) * self.frames[0].S
        if params.order_blocks == 0 and nb > 1:
            params = _copy_params(params)
            params.order_blocks = nb
        self.ctx.trace_device(params, self.n, self.d_k0.data_ptr(), self.d_end.data_ptr(), d_x0=self.d_x0.data_ptr(),
                              d_flags=self.d_flags.data_ptr(), d_n_steps=self.d_steps.data_ptr(),
                              d_n_accepted=self.d_acc.data_ptr(),
                              stream=torch.cuda.current_stream(self.dev).cuda_stream)
        for f in self.frames:    # (whole records, whatever the members were constructed with)
            f._dir_traced, f._traced = False, "end"

    def shade(self):
        return [f.shade() for f in self.frames]
