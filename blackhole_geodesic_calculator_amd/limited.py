"""Host-side mirror of the solver object the reference's LIMITED render engine drives -- the only place in the reference
where the exit sphere and the thin disk are specified.

Reference interface (curvedpy's older API; the package is not in the reference tree):
    self.SW = curvedpy.SchwarzschildGeodesic(metric=self.metric)          raytracer/LimitedRelativisticRenderEngine.py:90, :204
    x, y, z, end_loc, end_dir, mes = self.SW.ray_trace(direction, loc_hit=loc, exit_tolerance=, ratio_obj_to_blackhole=,
                                                       curve_end=self.SW.approximateCurveEnd(ratio), max_step=)   :273-278
    consumed: the sampled path x, y, z (checkHitDisk, :284, :413-438), end_loc, end_dir (:316-319),
              mes['hit_blackhole'] (:308), mes['error'] == 'Outside' (:311-314)
    the older inline formula for curve_end: 50 + 2*50*(ratio/20 - 1)      :279 (commented)
    metric: 'schwarzschild' (default, :487) or a flat one (README.md:233)

The engine's geometry (:259-266, :316-319): the hole sits inside a Blender object -- a sphere whose radius is
`ratio_obj_to_blackhole` horizon radii; a camera ray that hits that object at `loc` is handed over at loc - ob.location
(BH-centred), integrated through the curved region until it LEAVES the sphere again (or falls into the hole), and
continues as a straight Blender ray from end_loc + ob.location along end_dir.

What is known and what is inferred.  Known from the call sites: names, argument names and defaults (ratio 30, exit
tolerance 0.2, :488-489), the six return values and how each is consumed, the two `mes` keys.  INFERRED (the package's
source is not available): lengths are in units of the horizon radius (r_s = 1: `ratio_obj_to_blackhole` is the object's
radius over the hole's); the exit sphere is r_exit = ratio * r_s, crossed outward; `exit_tolerance` is the slack the
START point is allowed against that sphere -- a mesh hit lies on a facet, slightly off the ideal radius -- so that a
start further out than r_exit (1 + exit_tolerance) is reported as mes['error'] = 'Outside' and not integrated (the
engine paints such pixels red, :311-314), while a start within the slack is integrated as it stands (outside r_exit but
moving inward it crosses the sphere inward first -- no event -- and ends on its way out).  The arithmetic runs on the GPU
through libbhgeo.so: bhg_trace (exit event r_exit, disk annulus) and bhg_trajectory (the sampled path).
"""
from __future__ import annotations

import numpy as np

from . import _ffi

_METRICS = {"schwarzschild": 1.0, "flat": 0.0, "minkowski": 0.0}


class SchwarzschildGeodesic:
    """`curvedpy.SchwarzschildGeodesic(metric=)` as the Limited engine uses it, over the MI355X integrator."""

    def __init__(self, metric="schwarzschild", *, r_s=None, device=0, context=None, rtol=1e-3, atol=1e-6, rhs_form="christoffel",
                 nr_points_curve=200):
        if metric not in _METRICS:
            raise ValueError(f"metric must be one of {sorted(_METRICS)}")
        self.metric = metric
        self.r_s = float(_METRICS[metric] if r_s is None else r_s)   # horizon radius; 0 = flat space (README.md:233)
        self.rtol, self.atol = float(rtol), float(atol)
        self.rhs_form = {"christoffel": _ffi.RHS_CHRISTOFFEL, "reduced": _ffi.RHS_REDUCED}[rhs_form]
        self.nr_points_curve = int(nr_points_curve)
        self._ctx = context if context is not None else _ffi.Context(device)

    @property
    def context(self) -> _ffi.Context:
        return self._ctx

    @staticmethod
    def approximateCurveEnd(ratio_obj_to_blackhole):
        """curve_end heuristic, growing with the sphere (the formula the engine carried inline before, :279)."""
        return 50.0 + 2.0 * 50.0 * (float(ratio_obj_to_blackhole) / 20.0 - 1.0)

    def _unit(self):
        return self.r_s if self.r_s > 0.0 else 1.0     # (flat space: the sphere's radius is `ratio` length units)

    def params(self, ratio_obj_to_blackhole=30.0, curve_end=None, max_step=np.inf, disk=None) -> _ffi.Params:
        if max_step is None or max_step == -1:
            max_step = np.inf
        if curve_end is None:
            curve_end = self.approximateCurveEnd(ratio_obj_to_blackhole)
        r_exit = float(ratio_obj_to_blackhole) * self._unit()
        return _ffi.make_params(r_s=self.r_s, lambda_end=float(curve_end), max_step=max_step, rtol=self.rtol, atol=self.atol,
                                r_exit=r_exit, rhs_form=self.rhs_form, disk_r_in=disk[0] if disk else 0.0,
                                disk_r_out=disk[1] if disk else 0.0)

    # ------------------------------------------------------------------------------------------------------------
    def ray_trace_many(self, directions, loc_hits, exit_tolerance=0.2, ratio_obj_to_blackhole=30.0, curve_end=None,
                       max_step=np.inf, disk=None):
        """The batched form (one launch for N rays): directions [N, 3], loc_hits [N, 3] (BH-centred start points on the
        object's surface).  disk = (R_in, R_out) in length units (the engine passes disk_R_in * ratio, :285): the first
        plane crossing inside the annulus ends the ray there (located on the step's dense output, where checkHitDisk
        interpolates linearly between samples, :419-421).
        Returns (end_loc [N, 3], end_dir [N, 3], mes) with mes a dict of arrays: 'hit_blackhole' [N] bool, 'outside' [N]
        bool (the rays the per-ray form reports as mes['error'] == 'Outside': not integrated, end = start), 'hit_disk'
        [N] bool (end_loc is then the crossing point), 'flags', 'n_steps'."""
        d = np.ascontiguousarray(np.asarray(directions, dtype=np.float64).reshape(-1, 3))
        x0 = np.ascontiguousarray(np.asarray(loc_hits, dtype=np.float64).reshape(-1, 3))
        if x0.shape != d.shape:
            raise ValueError("directions and loc_hits must both be [N, 3]")
        r_exit = float(ratio_obj_to_blackhole) * self._unit()
        outside = np.linalg.norm(x0, axis=1) > r_exit * (1.0 + float(exit_tolerance))
        end, flags, steps, _ = self._ctx.trace(d, x0, self.params(ratio_obj_to_blackhole, curve_end, max_step, disk))
        end = np.array(end)
        flags = np.array(flags)
        if outside.any():        # not integrated: handed back where they started, flagged
            end[outside, 0:3] = x0[outside]
            end[outside, 3:6] = d[outside]
        mes = {"hit_blackhole": ((flags & _ffi.FLAG_HIT_HORIZON) != 0) & ~outside, "outside": outside,
               "hit_disk": (flags == _ffi.FLAG_HIT_DISK) & ~outside, "flags": flags, "n_steps": np.array(steps)}
        return end[:, 0:3], end[:, 3:6], mes

    def ray_trace(self, direction, loc_hit=None, exit_tolerance=0.2, ratio_obj_to_blackhole=30.0, curve_end=None, max_step=np.inf,
                  nr_points_curve=None, verbose=False, **_ignored):
        """Per-ray drop-in for the call at LimitedRelativisticRenderEngine.py:273-278.
        Returns (x, y, z, end_loc, end_dir, mes): x, y, z the path sampled at t = linspace(0, curve_end, nr_points_curve) up
        to where the ray ends (what checkHitDisk walks, :413-438); end_loc / end_dir the exact end state (the exit-sphere
        or horizon root); mes['hit_blackhole'], and mes['error'] = 'Outside' for a start beyond the tolerated radius
        (x, y, z then hold the start point alone)."""
        d = np.asarray(direction, dtype=np.float64).reshape(3)
        x0 = np.asarray(loc_hit if loc_hit is not None else (0.0, 0.0, 0.0), dtype=np.float64).reshape(3)
        r_exit = float(ratio_obj_to_blackhole) * self._unit()
        if np.linalg.norm(x0) > r_exit * (1.0 + float(exit_tolerance)):
            mes = {"hit_blackhole": False, "error": "Outside"}
            return x0[0:1].copy(), x0[1:2].copy(), x0[2:3].copy(), x0.copy(), d.copy(), mes
        n_pts = max(2, int(self.nr_points_curve if nr_points_curve is None else nr_points_curve))
        traj, nv, end, flags = self._ctx.trajectory(d[None, :], x0, self.params(ratio_obj_to_blackhole, curve_end, max_step), n_pts)
        m = int(nv[0])
        fl = int(flags[0])
        mes = {"hit_blackhole": bool(fl & _ffi.FLAG_HIT_HORIZON), "start_inside_hole": bool(fl & _ffi.FLAG_START_INSIDE), "flags": fl}
        if verbose:
            print("ray_trace:", mes)
        return traj[0, 0, :m], traj[0, 1, :m], traj[0, 2, :m], end[0, 0:3].copy(), end[0, 3:6].copy(), mes   # (views of this call's own block)


class ApproxSchwarzschildGeodesic:
    """`curvedpy.ApproxSchwarzschildGeodesic(ratio_obj_to_blackhole=, exit_tolerance=)` with `.generatedRayTracer(loc,
    direction) -> (end_loc, end_dir, mes)` and the attributes `.ratio_obj_to_blackhole`, `.exit_tolerance` the engine
    compares its settings with (raytracer/LimitedRelativisticRenderEngine.py:39, :97-101, :269).  In the reference this is
    a pre-tabulated approximation of the exact solve, keyed on the two settings, there because the exact solve is slow on
    the CPU; here it IS the exact solve (one GPU call), so nothing is tabulated and nothing depends on a data file --
    `generatedRayTracer` returns what `SchwarzschildGeodesic.ray_trace` returns, minus the sampled path."""

    def __init__(self, ratio_obj_to_blackhole=30.0, exit_tolerance=0.2, *, metric="schwarzschild", device=0, context=None, **solver_kw):
        self.ratio_obj_to_blackhole = float(ratio_obj_to_blackhole)
        self.exit_tolerance = float(exit_tolerance)
        self._sw = SchwarzschildGeodesic(metric=metric, device=device, context=context, **solver_kw)

    @property
    def context(self) -> _ffi.Context:
        return self._sw.context

    def generatedRayTracer(self, loc, direction):
        end_loc, end_dir, mes = self._sw.ray_trace_many(np.asarray(direction, dtype=np.float64).reshape(1, 3),
                                                        np.asarray(loc, dtype=np.float64).reshape(1, 3),
                                                        exit_tolerance=self.exit_tolerance,
                                                        ratio_obj_to_blackhole=self.ratio_obj_to_blackhole)
        out = {"hit_blackhole": bool(mes["hit_blackhole"][0])}
        if mes["outside"][0]:
            out["error"] = "Outside"
        return end_loc[0], end_dir[0], out

    def generatedRayTracer_many(self, locs, directions):
        """The batched form: (end_loc [N, 3], end_dir [N, 3], mes) as SchwarzschildGeodesic.ray_trace_many."""
        return self._sw.ray_trace_many(directions, locs, exit_tolerance=self.exit_tolerance,
                                       ratio_obj_to_blackhole=self.ratio_obj_to_blackhole)
