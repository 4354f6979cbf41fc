"""Host-side mirror of the solver object the reference's render engine drives.

Reference interface (the package it comes from, curvedpy, is not in the reference tree):
    curvedpy.GeodesicIntegratorSchwarzschild(mass=, time_like=False, verbose=False)
                                                    raytracer/RelativisticRenderEngine.py:134
    .calc_trajectory(k0_xyz, x0_xyz, max_step=, curve_end=, nr_points_curve=10000, verbose=False)
        -> (k_xyz, x_xyz, result)                   raytracer/RelativisticRenderEngine.py:293-294
    consumed: x_xyz[axis][-1], k_xyz[axis][-1]      :299-308
              result['start_inside_hole'], result['hit_blackhole']   :296-297

Same names, argument meaning and return shapes; the arithmetic runs on the GPU through
libbhgeo.so (include/bhgeo.h).  `trace()` is the batched form (one launch for N rays) that the
frame driver and the pre-traced camera use (Cam edition contract, CamEdition.py:225-228).
"""
from __future__ import annotations

import numpy as np

from . import _ffi

_METHODS = {"RK45": _ffi.METHOD_DP54, "DP54": _ffi.METHOD_DP54, "RK4": _ffi.METHOD_RK4}
_RHS = {"christoffel": _ffi.RHS_CHRISTOFFEL, "reduced": _ffi.RHS_REDUCED, "kerr_bl": _ffi.RHS_KERR_BL}


class GeodesicIntegratorSchwarzschild:
    """Geodesic integrator around a Schwarzschild black hole of horizon radius 2*mass: null rays (time_like=False, what
    the engine asks for) or massive particles (time_like=True)."""

    def __init__(self, mass=1.0, time_like=False, verbose=False, *, device=0, rtol=1e-3, atol=1e-6,
                 method="RK45", rhs_form="christoffel", h_fixed=0.1, max_steps=0, context=None):
        if method not in _METHODS:
            raise ValueError(f"method must be one of {sorted(_METHODS)}")
        if rhs_form not in _RHS:
            raise ValueError(f"rhs_form must be one of {sorted(_RHS)}")
        self.mass = float(mass)
        self.r_s = 2.0 * self.mass  # R_horizon = 2*M in geometrized units (:95)
        # time_like=True: massive particles, g(k, k) = -1, the curve parameter is the proper time -- the initial
        # "direction" k0 is then dx/dtau (the engine only ever passes False, RelativisticRenderEngine.py:134).  The norm
        # enters the Christoffel form through (k^t)^2 and the Boyer-Lindquist form through E and L at the start; the
        # reduced form is the closed form for NULL rays and cannot be asked for it.
        self.time_like = bool(time_like)
        if self.time_like and rhs_form == "reduced":
            raise ValueError("rhs_form='reduced' is the null closed form: time_like=True needs 'christoffel' or 'kerr_bl'")
        self.verbose = bool(verbose)
        self.rtol, self.atol = float(rtol), float(atol)
        self.method, self.rhs_form = method, rhs_form
        self.h_fixed, self.max_steps = float(h_fixed), int(max_steps)
        self.spin = 0.0  # Kerr a in length units (GeodesicIntegratorKerr)
        self._ctx = context if context is not None else _ffi.Context(device)

    # ------------------------------------------------------------------------------------
    @property
    def context(self) -> _ffi.Context:
        return self._ctx

    def params(self, max_step=np.inf, curve_end=50.0, r_exit=0.0, disk=None) -> _ffi.Params:
        if max_step is None or max_step == -1:  # the engine's "unset" sentinel (:59-60)
            max_step = np.inf
        return _ffi.make_params(r_s=self.r_s, lambda_end=curve_end, max_step=max_step, rtol=self.rtol,
                                atol=self.atol, h_fixed=self.h_fixed, r_exit=r_exit,
                                method=_METHODS[self.method], rhs_form=_RHS[self.rhs_form],
                                max_steps=self.max_steps, disk_r_in=disk[0] if disk else 0.0,
                                disk_r_out=disk[1] if disk else 0.0, spin=self.spin, time_like=self.time_like)

    # ------------------------------------------------------------------------------------
    def trace(self, k0, x0, max_step=np.inf, curve_end=50.0, r_exit=0.0, disk=None, spheres=None):
        """Batched solve.  k0[N,3] (or [...,3]); x0[3] shared origin or same leading shape as k0.
        r_exit: outward sphere-exit radius (Limited engine's ray_trace, Limited...py:273-278);
        disk=(R_in, R_out): thin disk in z = 0, first crossing inside the annulus ends the ray with
        FLAG_HIT_DISK and the crossing point in ray_end (checkHitDisk, Limited...py:413-438).
        spheres=[[cx, cy, cz, radius], ...] (at most 8, BH-centred): objects inside the curved region; a ray
        entering one ends there with FLAG_HIT_OBJECT and object_id = its index -- the collision test the
        reference leaves as a stub ("hit = False", RelativisticRenderEngine.py:304-305).

        Returns dict with
            ray_end[..., 6]            position (0:3) and direction (3:6) at the end of each curve
            ray_blackhole_hit[...]     uint8, 1 where the ray ended on the horizon
            flags[...], n_steps[...], n_accepted[...]
            object_id[...]             int8, only with spheres: index of the sphere hit, else -1
        """
        k0 = np.asarray(k0, dtype=np.float64)
        lead = k0.shape[:-1]
        k0f = k0.reshape(-1, 3)
        x0 = np.asarray(x0, dtype=np.float64)
        x0f = x0 if x0.ndim == 1 else x0.reshape(-1, 3)
        obj = None
        if spheres is not None:
            end, flags, steps, acc, obj = self._ctx.trace(k0f, x0f, self.params(max_step, curve_end, r_exit, disk),
                                                          spheres=spheres)
        else:
            end, flags, steps, acc = self._ctx.trace(k0f, x0f, self.params(max_step, curve_end, r_exit, disk))
        out = {
            "ray_end": end.reshape(lead + (6,)),
            "ray_blackhole_hit": ((flags & _ffi.FLAG_HIT_HORIZON) != 0).astype(np.uint8).reshape(lead),
            "flags": flags.reshape(lead),
            "n_steps": steps.reshape(lead),
            "n_accepted": acc.reshape(lead),
        }
        if obj is not None:
            out["object_id"] = obj.reshape(lead)
        return out

    # ------------------------------------------------------------------------------------
    def ray_set(self, width, height, samples, fov_x, fov_y, origin, rotation_euler=(0.0, 0.0, 0.0), jitter=None,
                jitter_is_compact=False, pixels=None) -> _ffi.RaySet:
        """The camera rays of a frame generated on the device and kept there (bhg_rays_create): the engine's
        pinhole + jitter formula (RelativisticRenderEngine.py:224-230) evaluated by libbhgeo's ray-generation
        kernel from the MT19937 stream; jitter=None gives pixel centres.  origin is BH-centred."""
        from .raygen import euler_xyz_matrix
        return _ffi.RaySet(self._ctx, width, height, samples, fov_x, fov_y, origin, euler_xyz_matrix(rotation_euler), jitter,
                           jitter_is_compact, pixels)

    def trace_rays(self, rays: _ffi.RaySet, first=0, n=None, max_step=np.inf, curve_end=50.0, r_exit=0.0, disk=None,
                   spheres=None, want=("end_dir", "flags")):
        """Trace rays [first, first + n) of a resident ray set; only the arrays named in `want` come back
        (end, end_loc, end_dir, flags, n_steps, n_accepted, object_id)."""
        return rays.trace(self.params(max_step, curve_end, r_exit, disk), first, n, want, spheres)

    # ------------------------------------------------------------------------------------
    def calc_trajectory(self, k0_xyz, x0_xyz, max_step=np.inf, curve_end=50, nr_points_curve=50,
                        verbose=False, r_exit=0.0, disk=None, spheres=None, **_ignored):
        """Per-ray drop-in for the call at RelativisticRenderEngine.py:293-294.

        Returns (k_xyz, x_xyz, result): k_xyz and x_xyz have shape (3, T') with the curve sampled at
        t_eval = linspace(0, curve_end, nr_points_curve) up to where the ray ends (T' <= nr_points_curve,
        as solve_ivp's t_eval gives), so that `x, y, z = x_xyz` (:299) and `x[-1]` (:307-308) work as
        in the reference.  result['end_loc'] / ['end_dir'] carry the exact end state (the event root for
        horizon rays), the same numbers trace() returns.  r_exit / disk=(R_in, R_out) (not in the reference's signature:
        the Limited engine's exit sphere and thin disk, Limited...py:273-278, :413-438): the curve then ends where the
        ray leaves the sphere or meets the disk, result['hit_disk'] says which.
        spheres=[[cx, cy, cz, radius], ...] (BH-centred, at most 8): objects in the curved region -- this call is exactly
        where the reference left its collision test as a stub ("NOW YOU DO COLLISION DETECTION ... hit = False", :304-305).
        A ray that enters a sphere ends there: result['hit_object'] = True, result['object_id'] = its index,
        result['end_loc'] the entry point -- the same flag, sphere and point trace(spheres=) gives for that ray -- and the
        sampled curve stops in front of it.
        """
        k0 = np.asarray(k0_xyz, dtype=np.float64).reshape(3)
        x0 = np.asarray(x0_xyz, dtype=np.float64).reshape(3)
        n_pts = max(2, int(nr_points_curve))
        obj = None
        if spheres is not None:
            traj, nv, end, flags, obj = self._ctx.trajectory(k0[None, :], x0, self.params(max_step, curve_end, r_exit, disk), n_pts,
                                                             spheres=spheres)
        else:
            traj, nv, end, flags = self._ctx.trajectory(k0[None, :], x0, self.params(max_step, curve_end, r_exit, disk), n_pts)
        m = int(nv[0])
        fl = int(flags[0])
        # (views of this call's own result block -- nothing else refers to it: two 240-kB copies less per call at the
        # engine's nr_points_curve = 10000)
        x_xyz = traj[0, 0:3, :m]
        k_xyz = traj[0, 3:6, :m]
        result = {
            "start_inside_hole": bool(fl & _ffi.FLAG_START_INSIDE),
            "hit_blackhole": bool(fl & _ffi.FLAG_HIT_HORIZON),
            "hit_disk": fl == _ffi.FLAG_HIT_DISK,
            "flags": fl,
            "end_loc": end[0, 0:3].copy(),
            "end_dir": end[0, 3:6].copy(),
        }
        if obj is not None:
            result["hit_object"] = fl == _ffi.FLAG_HIT_OBJECT
            result["object_id"] = int(obj[0])
        if verbose or self.verbose:
            print("calc_trajectory:", {k: result[k] for k in ("start_inside_hole", "hit_blackhole", "flags")})
        return k_xyz, x_xyz, result

class GeodesicIntegratorKerr(GeodesicIntegratorSchwarzschild):
    """Null geodesics around a Kerr black hole (BASELINE.json config 5).

    The reference lists Kerr as a goal (README.md:218) and its pre-traced camera takes `a = 0.9`
    (raytracer/RelativisticRenderEngineCamEdition.py:210, :217); with mass 0.5 that can only be the
    dimensionless spin a/M, which is what `a` means here.  Same boundary as the Schwarzschild
    integrator: Cartesian k0 / x0 in, Cartesian end state out; integrated in Boyer-Lindquist
    coordinates with sympy-derived Christoffel symbols (tools/gen_kerr_rhs.py)."""

    def __init__(self, mass=1.0, a=0.0, time_like=False, verbose=False, **kw):
        kw.pop("rhs_form", None)
        super().__init__(mass=mass, time_like=time_like, verbose=verbose, rhs_form="kerr_bl", **kw)
        if not abs(a) < 1.0:
            raise ValueError("|a| = |J|/M^2 must be < 1")
        self.a = float(a)
        self.spin = self.a * self.mass
        self.r_plus = self.mass + (self.mass**2 - self.spin**2) ** 0.5
