"""GPU parity tests (run with -m gpu on an MI355X).  Everything goes through the C ABI
(libbhgeo.so via ctypes) and is compared with the CPU oracle on identical inputs.

Stated tolerances (fp64):
  * flags, attempted and accepted step counts: identical (same controller decisions).  One
    documented exception: horizon-hitting rays integrated with the CHRISTOFFEL form at tight
    tolerances (rtol <= 1e-6) -- that form sums 1/(r-r_s)^2 terms that cancel, so close to the
    horizon its error estimate is rounding noise and an accept/reject can flip; at most 0.5 % of
    rays, horizon rays only, step counts within 3 (allow_flips=True below).
  * horizon rays additionally get 1e-6 (their end state sits on the coordinate singularity, where
    k^i diverges; the engine only reads their flag, RelativisticRenderEngine.py:242-244);
  * disk hits additionally get 1e-11 |k| / |k_z| (a plane crossing is located only as sharply as
    the ray is steep);
  * end state: |gpu - oracle| <= TOL_END + COND * S_i, where S_i is the oracle's OWN sensitivity
    of ray i to a 1-ulp perturbation of k0 (measured per ray, in the test).  The device kernel
    re-associates the RK sums (Nystrom form, FMA); rays that orbit near the photon sphere amplify
    any 1e-16 noise exponentially, and S_i is exactly that amplification.  Median <= 1e-11.
"""
import math

import numpy as np
import pytest

from conftest import CAM, GOLDEN_TRACE_SETS, frame_rays, golden_kwargs, load_golden

pytestmark = pytest.mark.gpu

TOL_END = 1e-9   # absolute floor, values are O(1..50)
# Largest fraction of a Kerr fuzz draw's rays whose step sequence may differ from the oracle's (all of them "touchy": horizon rays
# or small L_z).  Measured over 375 draws (round 5, BHG_FUZZ=1500, profiles/r05_fuzz1500.log + r05_kerr_fuzz.log): > 90 % of the
# draws have no such ray at all; the worst are 2.53 % (31 of 1226 rays, half of them horizon rays) and 2 of 76 rays -- asserted
# at 4 % or three rays, whichever is more (8 % up to round 4).
KERR_FUZZ_DIFFER = 0.04
# ... and per class since the 1 187-draw run (round 5, BHG_FUZZ=6000, profiles/r05_fuzz6000.log, scripts/dev/dev_r05_fuzz_fail.py):
# two draws whose rays nearly all END ON THE HORIZON (87 % and 64 % of them, camera near the equatorial plane, rtol 5e-3) had
# 5.1 % and 6.4 % of their rays differ -- every one a horizon ray, 5.9 % and 10 % of the draw's horizon rays, by one or two
# steps in the median (the right-hand side is singular there, 1/Delta).  So: horizon rays (either side says so) against
# the draw's horizon rays, the others against the draw.
# The run repeated with these bounds: 1 499 Kerr draws, 1.85 M rays, 483 k of them horizon rays -- 102 draws have a ray that
# differs, 618 rays in all: 614 horizon rays (worst draws 10.4 %, 9.9 %, 5.9 % of their horizon rays), 4 near-axis rays (worst
# draw 0.075 % of its rays), none that is neither.
# (asserted at 15 % in round 5; ADVICE r05: near the measured worst, 10.4 %, plus a margin)
KERR_FUZZ_DIFFER_HORIZON = 0.125
KERR_FUZZ_DIFFER_OTHER = 0.005
LAST_COMPARE = {}   # filled by _compare: rays compared, rays further than TOL_END from the oracle, rays beyond the scaled bound
COND = 500.0     # multiples of the oracle's own 1-ulp input sensitivity S_i (an estimate from three perturbations, not a
                 # bound).  Measured over 240 fuzz draws (round 4, LAST_COMPARE["worst_multiple_of_sensitivity"]): the worst ray
                 # of the worst draw sits at 187 S_i, the 99th percentile of the draws' worst rays at 6.3 S_i, the median draw
                 # has no ray beyond the absolute floor at all (round 3 asserted 1e3)


def _params(**kw):
    from blackhole_geodesic_calculator_amd import _ffi
    return _ffi.make_params(**kw)


def _sensitivity(oracle, k0, x0, ref_end, **kw):
    """Per-ray conditioning: how far the ORACLE's end state moves when k0 moves by an ulp or two
    (the largest of three perturbation patterns; one pattern alone can sit in a ray's blind spot)."""
    k0 = np.asarray(k0, float)
    eps = np.finfo(float).eps
    pats = (np.nextafter(k0, np.inf), np.nextafter(k0, -np.inf), k0 * (1.0 + np.array([2.0, -2.0, 2.0]) * eps))
    with np.errstate(invalid="ignore"):      # (inf - inf on rays whose fixed-step state overflowed: NaN, treated as "no bound" by the callers)
        return np.max([np.abs(oracle.trace(kp, x0, **kw)["end"] - ref_end).max(1) for kp in pats], axis=0)


ROUNDING_FLIPS = {"draws": 0, "rays": 0, "flips": 0}    # over the whole run, fuzz draws only (see _compare)
ROUNDING_FLIPS_MAX = 2     # measured: ONE such ray in 9 500 draws / 13 M rays (round 5); a suite run has ~100 draws


def _compare(ctx, oracle, k0, x0, allow_flips=False, outliers=0.0, step_flips=0, rounding_flips=0, **kw):
    o = oracle.trace(k0, x0, **kw)
    spheres = kw.get("spheres")
    if spheres is not None:
        end, flags, steps, acc, obj = ctx.trace(k0, x0, _params(**{a: b for a, b in kw.items() if a != "spheres"}),
                                                spheres=spheres)
        assert np.array_equal(obj, o["object_id"])
    else:
        end, flags, steps, acc = ctx.trace(k0, x0, _params(**kw))
    if allow_flips:
        # (fixed steps through the coordinate singularity: whether such a ray is seen crossing the horizon radius
        # before its state turns NaN is rounding noise too -- both sides must still call it horizon and/or NaN)
        fbad = flags != o["flags"]
        assert np.all((flags[fbad] & ~np.uint8(1 | 64)) == 0) and np.all((o["flags"][fbad] & ~np.uint8(1 | 64)) == 0)
        bad = fbad | (steps != o["n_attempted"]) | (acc != o["n_accepted"])
        assert bad.mean() <= (0.005 if allow_flips is True else allow_flips) and np.all((flags[bad] & (1 | 64)) != 0)
        assert np.abs(steps.astype(int) - o["n_attempted"].astype(int)).max(initial=0) <= 3
    elif step_flips:
        # Kerr at scale: flags identical, and at most `step_flips` rays may take a different number of steps, each by at
        # most two (an accept / reject decision within rounding of err_norm = 1: the start state is converted to
        # Boyer-Lindquist with different -- equally accurate -- sqrt / atan2 / sincos on the two sides, a few ulp apart)
        assert np.array_equal(flags, o["flags"])
        sdiff = (steps != o["n_attempted"]) | (acc != o["n_accepted"])
        assert sdiff.sum() <= step_flips, int(sdiff.sum())
        assert np.abs(steps.astype(int) - o["n_attempted"].astype(int)).max(initial=0) <= 2
    else:
        assert np.array_equal(flags, o["flags"])
        sdiff = np.nonzero((steps != o["n_attempted"]) | (acc != o["n_accepted"]))[0]
        # Identical accept / reject sequences on EVERY ray: the default, and what every deterministic test asks.  Only the
        # randomised draws pass rounding_flips = 2: at most that many rays of a draw may differ, each with a
        # sequence the CHECKER ITSELF produces when the ray's direction moves by an ulp or two -- an error norm within rounding of
        # 1.  (9 500 draws, 13 M rays, round 5, profiles/r05_fuzz6000.log: one such ray -- a horizon ray at rtol 1.5e-6, 29 / 16
        # steps on the GPU, 31 / 17 in the checker, 29 / 16 in the checker with k0 scaled by 1 - 1e-16.)  The run's total is
        # kept in ROUNDING_FLIPS and may not pass ROUNDING_FLIPS_MAX.
        assert len(sdiff) <= rounding_flips, f"{len(sdiff)} rays differ in step count (allowed: {rounding_flips})"
        for i in sdiff:
            xi = x0 if np.ndim(x0) == 1 else np.asarray(x0)[i]
            seen = set()
            for eps in (1e-16, -1e-16, 2e-16, -2e-16, 3e-16, -3e-16, 4e-16, -4e-16):
                oo = oracle.trace(np.asarray(k0)[i:i + 1] * (1.0 + eps), xi, **kw)
                seen.add((int(oo["n_attempted"][0]), int(oo["n_accepted"][0])))
            assert (int(steps[i]), int(acc[i])) in seen, f"ray {i}: GPU {steps[i]}/{acc[i]}, checker {o['n_attempted'][i]}/{o['n_accepted'][i]}, checker near by {seen}"
        if rounding_flips:
            ROUNDING_FLIPS["draws"] += 1
            ROUNDING_FLIPS["rays"] += int(len(steps))
            ROUNDING_FLIPS["flips"] += int(len(sdiff))
            if len(sdiff):
                print(f"rounding flips: {len(sdiff)} in this draw, {ROUNDING_FLIPS}")
            assert ROUNDING_FLIPS["flips"] <= max(ROUNDING_FLIPS_MAX, 2e-6 * ROUNDING_FLIPS["rays"]), ROUNDING_FLIPS
        LAST_COMPARE["rounding_flips"] = int(len(sdiff))
    d = np.abs(end - o["end"]).max(1) if len(end) else np.zeros(0)
    if len(end):
        fin = np.isfinite(o["end"]).all(1)
        cond = COND * (10.0 if kw.get("rhs_form") == 2 else 1.0)  # Kerr: hundreds of libm sin/cos calls differ by an ulp
        tol = TOL_END + cond * np.nan_to_num(_sensitivity(oracle, k0, x0, o["end"], **kw), nan=np.inf, posinf=np.inf)
        # horizon rays end AT the coordinate singularity: k^i and (Christoffel form) the rounding noise of
        # the 1/(r-r_s) terms grow without bound there; the engine never reads this state (:242-244)
        tol = tol + np.where((o["flags"] & 1) != 0, 1e-6, 0.0)
        # a disk-plane crossing is only as well located as the ray is steep: dt = dz / |k_z|
        dsk = o["flags"] == 128
        if dsk.any():
            # (over the disk rays only: other rows may hold the +-inf / NaN of an overflowed fixed-step state, and a reduction
            # over inf - inf is an "invalid value" warning in the parity path)
            kd = o["end"][dsk, 3:6]
            steep = np.abs(kd[:, 2]) / np.linalg.norm(kd, axis=1)
            tol[dsk] += 1e-11 / np.maximum(steep, 1e-12)
        # ... and so is the entry point into an object sphere: dt = dg / |n.k|, n the surface normal there
        hit = o["flags"] == 0x88
        if hit.any():
            c = np.asarray(spheres, float).reshape(-1, 4)[o["object_id"][hit]]
            nrm = (o["end"][hit, 0:3] - c[:, 0:3]) / c[:, 3:4]
            kk = o["end"][hit, 3:6]
            steep = np.abs((nrm * kk).sum(1)) / np.linalg.norm(kk, axis=1)
            tol[hit] += 1e-11 / np.maximum(steep, 1e-12)
        over = d[fin] > tol[fin]
        # how much work the sensitivity-scaled part of the bound does: rays further than the absolute floor from the oracle
        # (rays cut off by the step budget end at a lambda that is the SUM of their step sizes: with tolerances near
        # rounding level the error estimate -- a difference of nearly equal stage values -- is noise at 1e-8 relative,
        # the step sizes follow it, and the state at the cut-off moves by k * d(lambda); rays that run to lambda_end or to
        # an event end at a pinned place.  The record keeps the two apart.)
        cut = (o["flags"] & np.uint8(16 | 32)) != 0
        pinned = fin & ~cut
        # the largest multiple of a ray's own sensitivity its difference amounts to (beyond the absolute floor): what COND bounds
        sens = (tol - TOL_END) / cond
        with np.errstate(divide="ignore", invalid="ignore"):
            ratio = np.where((d > TOL_END) & fin & (sens > 0), (d - TOL_END) / sens, 0.0)
        # (the record's floor follows the size of the numbers: TOL_END is 1e-9 on values of O(1..50); a draw with r_s = 2.5,
        # the camera at 108 and paths of 380 ends rays at |x| ~ 270, where 1e-9 is 4e-12 of the value)
        floor = TOL_END * np.maximum(1.0, np.abs(np.nan_to_num(o["end"])).max(1) / 50.0)
        LAST_COMPARE.update(rays=int(fin.sum()), beyond_floor=int((d[pinned] > floor[pinned]).sum()), beyond_bound=int(over.sum()),
                            worst_multiple_of_sensitivity=float(np.nan_to_num(ratio, nan=0.0, posinf=0.0).max(initial=0.0)),
                            worst=float(d[pinned].max(initial=0.0)), cut_off=int((fin & cut).sum()),
                            worst_cut_off=float(d[fin & cut].max(initial=0.0)))
        # `outliers`: S_i is an estimate from three perturbations, not a bound; the fuzz test lets a
        # fraction of rays exceed it, but never by more than a factor 1e3
        assert over.mean() <= outliers and np.all(d[fin] <= 1e3 * tol[fin]), \
            f"worst end-state excess {np.max(d[fin] - tol[fin])}, {over.sum()} rays over"
    return end, flags, steps, d


@pytest.mark.parametrize("rhs_form", [0, 1])
@pytest.mark.parametrize("name", GOLDEN_TRACE_SETS)
def test_golden_vectors(ctx, oracle, name, rhs_form):
    """GPU against the committed scipy vectors AND against the oracle on the same inputs."""
    g = load_golden(name)
    kw = golden_kwargs(g, rhs_form)
    end, flags, steps, d = _compare(ctx, oracle, g["k0"], g["x0"], **kw)
    assert np.array_equal(flags, g["flags"])
    assert np.array_equal(steps, g["n_attempted"])
    assert np.abs(end - g["end"]).max() <= 1e-8
    assert np.median(d) <= 1e-11


# The stated fp64 tolerance as plain numbers: |gpu - oracle| and |gpu - scipy golden| per class of
# ray on every committed golden set -- asserted bound; the measured worst case is in the comment.  `over` = rays that
# differ from the oracle by more than 1e-9, i.e. the ones for which _compare's sensitivity-scaled bound does any work.
STATED = {  # class: (Schwarzschild bound, Kerr bound)
    "escaped": (1e-8, 5e-8),    # measured 4.1e-9 (one ray of the disk set that winds around the photon sphere) / 1.2e-8
    "horizon": (5e-9, 1e-6),    # measured 1.5e-9 / 2.6e-7 (end state on the Boyer-Lindquist coordinate singularity)
    "disk": (1e-10, 1e-9),      # measured 1.1e-11 / 4.9e-11
    "object": (1e-12, None),    # measured 1.2e-14
}
CLASS_OF = {"escaped": lambda f: (f == 4) | (f == 8), "horizon": lambda f: (f & 1) != 0, "disk": lambda f: f == 128,
            "object": lambda f: f == 0x88}


def _golden_cases():
    for rhs in (0, 1):
        for name in GOLDEN_TRACE_SETS:
            yield name, rhs, None, slice(None)
        yield "disk", rhs, dict(r_s=1.0, lambda_end=80.0, disk_r_in=4.5, disk_r_out=10.5, rhs_form=rhs), slice(None)
    yield "objects", 0, dict(r_s=1.0, lambda_end=70.0, max_step=0.25, r_exit=35.0, disk_r_in=3.0, disk_r_out=7.0), slice(None)
    yield "kerr_a09", 2, dict(r_s=1.0, lambda_end=60.0, rhs_form=2), slice(48, None)    # off-axis camera (see test_kerr_golden_and_oracle)
    yield "kerr_disk", 2, dict(r_s=1.0, lambda_end=80.0, rhs_form=2, disk_r_in=3.0, disk_r_out=10.0), slice(None)


@pytest.mark.parametrize("name,rhs,kw,sl", list(_golden_cases()), ids=lambda v: str(v) if isinstance(v, (str, int)) else "")
def test_stated_tolerance_per_class_on_every_golden_set(ctx, oracle, name, rhs, kw, sl):
    g = load_golden(name)
    kw = dict(kw) if kw is not None else golden_kwargs(g, rhs)
    if "spin" in g and rhs == 2:
        kw["spin"] = float(g["spin"])
    sp = g["spheres"] if name == "objects" else None
    k0, x0 = g["k0"][sl], (g["x0"][sl] if g["x0"].ndim == 2 else g["x0"])
    r = ctx.trace(k0, x0, _params(**kw), spheres=sp) if sp is not None else ctx.trace(k0, x0, _params(**kw))
    end, flags = r[0], r[1]
    o = oracle.trace(k0, x0, spheres=sp, **kw) if sp is not None else oracle.trace(k0, x0, **kw)
    assert np.array_equal(flags, o["flags"]) and np.array_equal(flags, g["flags"][sl])
    d_o = np.abs(end - o["end"]).max(1)
    d_g = np.abs(end - g["end"][sl]).max(1)
    seen = 0
    for cls, sel in CLASS_OF.items():
        m = sel(flags)
        if not m.any():
            continue
        seen += int(m.sum())
        bound = STATED[cls][1 if rhs == 2 else 0]
        assert d_o[m].max() <= bound and d_g[m].max() <= bound, (name, rhs, cls, d_o[m].max(), d_g[m].max())
    assert seen == len(flags)                                     # every ray falls in exactly one class
    over = int((d_o > 1e-9).sum())
    assert over <= (1 if rhs != 2 else 40), over                  # measured: 0 or 1 per Schwarzschild set, 34 / 7 for the Kerr sets


@pytest.mark.parametrize("rhs_form", [0, 1])
def test_acceleration_matches_oracle(ctx, oracle, rhs_form):
    rng = np.random.default_rng(3)
    x = rng.normal(size=(4000, 3)) * 6
    x = x[np.linalg.norm(x, axis=1) > 1.2]
    k = rng.normal(size=x.shape)
    a = ctx.acceleration(x, k, _params(r_s=1.0, rhs_form=rhs_form))
    ref = oracle.acceleration(x, k, r_s=1.0, rhs_form=rhs_form)
    rel = np.abs(a - ref).max(1) / np.abs(ref).max(1)
    # The Christoffel form sums O(1) terms that cancel to O(L^2) for nearly radial rays, so its
    # rounding noise relative to the RESULT is large there (the oracle's own evaluation has the
    # same property); the reduced form cancels the same way in L^2 = r^2 k^2 - (x.k)^2.
    assert rel.max() < 5e-11
    assert np.median(rel) < 1e-15


@pytest.mark.parametrize("M,a", [(0.5, 0.45), (1.3, -0.9)])
def test_kerr_acceleration_on_the_device_matches_oracle_and_hamiltonian_form(ctx, oracle, M, a):
    """The DEVICE Kerr right-hand side (accel_kerr_bl: Newton reciprocals, the shared sincos, product inversion),
    probed point by point through bhg_acceleration with rhs_form = BHG_RHS_KERR_BL -- Boyer-Lindquist triples in and
    out, E and L from the null condition at the point -- against the oracle's evaluation of the same generated
    snippet (libm sin / cos, IEEE divisions) and against Hamilton's equations with the inverse metric
    (tests/kerr_hamiltonian.py: no Christoffel symbols, nothing generated)."""
    import kerr_hamiltonian as kh
    q, u = kh.sample_points(3000, M, a, seed=11)
    got = ctx.acceleration(q, u, _params(r_s=2 * M, rhs_form=2, spin=a))
    ref = oracle.acceleration(q, u, r_s=2 * M, rhs_form=2, spin=a)
    scale = np.abs(ref).max(1) + 1e-300
    r_plus = M + np.sqrt(M * M - a * a)
    near = 1.0 / ((q[:, 0] - r_plus) / r_plus) ** 2      # both lose digits to the cancellation in Delta near r_plus
    rel = np.abs(got - ref).max(1) / scale
    assert np.all(rel < 1e-12 + 1e-15 * near), float((rel / (1e-12 + 1e-15 * near)).max())
    assert np.median(rel) < 5e-15
    ham = np.array([kh.acceleration(q[i], u[i], M, a) for i in range(400)])
    rel_h = np.abs(got[:400] - ham).max(1) / (np.abs(ham).max(1) + 1e-300)
    assert np.all(rel_h < 1e-9 + 1e-12 * near[:400]) and np.median(rel_h) < 1e-12


@pytest.mark.parametrize("rhs_form", [0, 1])
def test_seeded_rays_default_controller(ctx, oracle, rhs_form):
    k = frame_rays(50000, seed=21)
    _compare(ctx, oracle, k, CAM, r_s=1.0, lambda_end=50.0, rhs_form=rhs_form)


def test_per_ray_origins_and_ragged_sizes(ctx, oracle):
    rng = np.random.default_rng(22)
    for n in (1, 2, 63, 64, 65, 127, 129, 1000):
        k = frame_rays(n, seed=n)
        x0 = CAM + rng.normal(size=(n, 3)) * np.array([2.0, 2.0, 4.0])
        _compare(ctx, oracle, k, x0, r_s=1.0, lambda_end=60.0)


def test_empty_input(ctx):
    end, flags, steps, acc = ctx.trace(np.zeros((0, 3)), CAM, _params())
    assert end.shape == (0, 6) and flags.shape == (0,) and steps.shape == (0,)


def test_consecutive_calls_of_every_shape_on_one_context(oracle):
    """The work counters come in two sets used by alternate launches, each launch zeroing the set of the next: a context
    must give the same answers whatever it ran before -- sizes from one ray to many batches, empty calls in between,
    event variants, both steppers, Kerr, and the host-buffer chunk pipeline (several launches per call)."""
    from blackhole_geodesic_calculator_amd import _ffi
    c = _ffi.Context(0)
    try:
        cases = []
        for i, n in enumerate((1, 64, 65, 4097, 20000)):
            k = frame_rays(n, seed=100 + i)
            cases.append((k, _params(r_s=1.0, lambda_end=50.0)))
            cases.append((k, _params(r_s=1.0, lambda_end=80.0, r_exit=40.0, disk_r_in=4.5, disk_r_out=10.5)))
            cases.append((k, _params(r_s=1.0, lambda_end=50.0, method=1, h_fixed=0.1)))
            cases.append((k, _params(r_s=1.0, lambda_end=50.0, rhs_form=2, spin=0.3)))
        first = [c.trace(k, CAM, p) for k, p in cases]
        rng = np.random.default_rng(5)
        for it in range(3):
            for j in rng.permutation(len(cases)):
                if j % 5 == 0:
                    c.trace(np.zeros((0, 3)), CAM, cases[j][1])       # an empty call launches nothing, flips nothing
                got = c.trace(cases[j][0], CAM, cases[j][1])
                for a, b in zip(first[j], got):
                    assert np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True), (it, j)
        k, p = cases[1]
        end, flags, steps, _ = first[1]
        o = oracle.trace(k, CAM, r_s=1.0, lambda_end=80.0, r_exit=40.0, disk_r_in=4.5, disk_r_out=10.5)
        assert np.array_equal(flags, o["flags"]) and np.array_equal(steps, o["n_attempted"])
    finally:
        c.close()


def test_start_inside_mixed_with_normal_rays(ctx, oracle):
    n = 500
    k = frame_rays(n, seed=23)
    x0 = np.tile(CAM, (n, 1))
    x0[::7] = np.array([0.3, -0.2, 0.4])  # inside the horizon
    x0[5::64] = np.array([0.0, 0.0, 1.0])  # exactly on it: r0 <= r_s counts as inside
    end, flags, steps, _ = _compare(ctx, oracle, k, x0, r_s=1.0, lambda_end=50.0)
    inside = (flags & 2) != 0
    assert inside.sum() == len(range(0, n, 7)) + len([i for i in range(5, n, 64) if i % 7])
    assert np.all(steps[inside] == 0)
    assert np.array_equal(end[inside, 0:3], x0[inside]) and np.array_equal(end[inside, 3:6], k[inside])


def test_flat_space_and_other_masses(ctx, oracle):
    k = frame_rays(300, seed=24)
    end, flags, _, _ = _compare(ctx, oracle, k, CAM, r_s=0.0, lambda_end=50.0, rhs_form=1)
    assert np.all(flags == 4)
    assert np.abs(end[:, 0:3] - (CAM + 50.0 * k)).max() < 1e-12
    _compare(ctx, oracle, k, CAM, r_s=0.0, lambda_end=50.0, rhs_form=0)
    _compare(ctx, oracle, k, np.array([2.0, -1.0, 60.0]), r_s=3.0, lambda_end=120.0)


def test_max_step_regimes(ctx, oracle):
    k = frame_rays(600, seed=25)
    _compare(ctx, oracle, k, CAM, r_s=1.0, lambda_end=50.0, max_step=0.1)           # R-fine, ~490 steps/ray
    _compare(ctx, oracle, k, CAM, r_s=1.0, lambda_end=50.0, max_step=1e4)           # the engine's property default
    _compare(ctx, oracle, k, CAM, r_s=1.0, lambda_end=50.0, rtol=1e-8, atol=1e-10, rhs_form=1)  # tight tolerances
    _compare(ctx, oracle, k, CAM, r_s=1.0, lambda_end=50.0, rtol=1e-8, atol=1e-10, rhs_form=0, allow_flips=True)


def test_step_cap_and_tiny_lambda(ctx, oracle):
    k = frame_rays(200, seed=26)
    _, flags, steps, _ = _compare(ctx, oracle, k, CAM, r_s=1.0, lambda_end=50.0, max_steps=5)
    assert np.all((flags == 16) == (steps == 5)) and (flags == 16).any()
    _compare(ctx, oracle, k, CAM, r_s=1.0, lambda_end=1e-3)
    _compare(ctx, oracle, k, CAM, r_s=1.0, lambda_end=0.0)


def test_sphere_exit_event(ctx, oracle):
    rng = np.random.default_rng(27)
    n = 2000
    pos = rng.normal(size=(n, 3))
    pos = 30.0 * pos / np.linalg.norm(pos, axis=1)[:, None]
    aim = rng.normal(size=(n, 3)) * 5.0
    d = aim - pos
    d /= np.linalg.norm(d, axis=1)[:, None]
    end, flags, _, _ = _compare(ctx, oracle, d, pos, r_s=1.0, lambda_end=100.0, r_exit=30.0)
    ex = flags == 8
    assert ex.sum() > 0.8 * n
    assert np.abs(np.linalg.norm(end[ex, 0:3], axis=1) - 30.0).max() < 1e-9


def _disk_rays(n, seed, inc_deg):
    inc = math.radians(inc_deg)
    cam = np.array([30 * math.sin(inc), 0.0, 30 * math.cos(inc)])
    aim = np.random.default_rng(seed).normal(size=(n, 3)) * np.array([9.0, 9.0, 1.0])
    d = aim - cam
    return d / np.linalg.norm(d, axis=1)[:, None], cam


@pytest.mark.parametrize("rhs_form", [0, 1])
def test_disk_golden_and_oracle(ctx, oracle, rhs_form):
    """Thin-disk crossing (config 3 geometry): scipy golden vectors and the oracle."""
    g = load_golden("disk")
    kw = dict(r_s=1.0, lambda_end=80.0, disk_r_in=4.5, disk_r_out=10.5, rhs_form=rhs_form)
    end, flags, steps, d = _compare(ctx, oracle, g["k0"], g["x0"], **kw)
    assert np.array_equal(flags, g["flags"])
    assert np.abs(end - g["end"]).max() < 1e-8
    assert ctx.last_launch()["passes"] == 1  # plane crossings outside the annulus are resumed inside the one launch


@pytest.mark.parametrize("inc_deg", [85.0, 80.0, 60.0, 30.0, 5.0])
def test_disk_five_inclinations(ctx, oracle, inc_deg):
    k, cam = _disk_rays(20000, int(inc_deg), inc_deg)
    end, flags, steps, d = _compare(ctx, oracle, k, cam, r_s=1.0, lambda_end=80.0, disk_r_in=4.5, disk_r_out=10.5)
    disk = flags == 128
    assert disk.sum() > 1000
    R = np.hypot(end[disk, 0], end[disk, 1])
    assert np.abs(end[disk, 2]).max() < 1e-12 and R.min() >= 4.5 and R.max() <= 10.5
    assert np.all((flags == 128) | (flags == 4) | (flags == 1))


def test_disk_with_exit_sphere_rk4_and_fine(ctx, oracle):
    k, cam = _disk_rays(3000, 7, 70.0)
    _compare(ctx, oracle, k, cam, r_s=1.0, lambda_end=100.0, disk_r_in=3.0, disk_r_out=12.0, r_exit=30.5)
    _compare(ctx, oracle, k, cam, r_s=1.0, lambda_end=80.0, disk_r_in=4.5, disk_r_out=10.5, method=1, h_fixed=0.2)
    _compare(ctx, oracle, k[:500], cam, r_s=1.0, lambda_end=80.0, disk_r_in=4.5, disk_r_out=10.5, max_step=0.25)
    # a disk nobody hits (tiny annulus far away) must not change anything
    a = ctx.trace(k, cam, _params(r_s=1.0, lambda_end=80.0))
    b = ctx.trace(k, cam, _params(r_s=1.0, lambda_end=80.0, disk_r_in=500.0, disk_r_out=501.0))
    assert all(np.array_equal(x, y) for x, y in zip(a[1:], b[1:]))            # flags, step counts
    hor = (a[1] & 1) != 0
    # (the disk variant of the kernel is another instruction stream than the plain one: a horizon root -- Brent's in
    # both -- agrees to rounding, amplified by the diverging k at the coordinate singularity)
    assert np.array_equal(a[0][~hor], b[0][~hor]) and np.abs(a[0][hor] - b[0][hor]).max() < 1e-8


def _grazing_rays(n, seed, sigma=(9.0, 9.0, 0.03), inc_lo=60.0):
    """Cameras on the circle r = 30 at inclinations inc_lo ... 89.999 degrees (half of them within a degree of the disk
    plane), aimed at points scattered (sigma) about the hole: plane crossings that graze."""
    rng = np.random.default_rng(seed)
    inc = np.deg2rad(np.where(rng.random(n) < 0.5, rng.uniform(89.0, 89.999, n), rng.uniform(inc_lo, 89.999, n)))
    cam = 30.0 * np.stack([np.sin(inc), np.zeros(n), np.cos(inc)], -1)
    k = rng.normal(size=(n, 3)) * np.asarray(sigma) - cam
    return k / np.linalg.norm(k, axis=1)[:, None], cam


@pytest.mark.parametrize("rtol", [3e-3, 1e-2, 1e-1])
@pytest.mark.parametrize("rhs_form", [0, 1])
def test_grazing_disk_crossings_from_near_equatorial_cameras(ctx, oracle, rtol, rhs_form):
    """The regime in which round 3's disk pre-filter dropped real hits: near-equatorial cameras, a wide disk, loose
    tolerances (long steps).  A step the filter rejects never reaches the event search, so any hit it loses shows as
    a flag / step-count difference against the oracle, which has no filter (tests/test_disk_filter_bound.py holds the
    bound itself).  Wide annulus + narrow ones whose edges sit where the crossings are."""
    k, cam = _grazing_rays(12000, int(rtol * 1e4) + rhs_form)
    hits = 0
    for r_in, r_out in ((4.5, 35.0), (9.0, 11.0), (16.5, 17.5), (24.0, 26.0)):
        end, flags, steps, d = _compare(ctx, oracle, k, cam, r_s=1.0, lambda_end=90.0, disk_r_in=r_in, disk_r_out=r_out,
                                        rtol=rtol, atol=rtol * 1e-3, rhs_form=rhs_form)
        disk = flags == 128
        hits += int(disk.sum())
        R = np.hypot(end[disk, 0], end[disk, 1])
        assert np.abs(end[disk, 2]).max(initial=0.0) < 1e-10 and R.min(initial=r_in) >= r_in and R.max(initial=r_in) <= r_out
    assert hits > 4000
    # the fixed-step kernels filter on the cubic Hermite interpolant they locate events on
    _compare(ctx, oracle, k[:3000], cam[:3000], r_s=1.0, lambda_end=90.0, disk_r_in=4.5, disk_r_out=35.0, method=1, h_fixed=2.0,
             rhs_form=rhs_form)


@pytest.mark.parametrize("rtol", [3e-2, 1e-1])
def test_disk_filter_keeps_every_grazing_hit_next_to_an_annulus_edge(ctx, oracle, rtol):
    """Where a too-small excursion figure actually LOSES a hit: the dense crossing and the chord's lie on different
    sides of an annulus edge and further apart than the figure.  About one grazing crossing in 2,000 exceeds round 3's
    figure, one in 10^4 with an edge in between -- so: 400,000 rays from cameras within 0.3 degrees of the plane, all
    aimed at a ring of width 0.6 around the edge R = 17, once as the outer and once as the inner edge of the disk
    (round 3's library loses ~30 hits here; measured with BHGEO_LIB pointing at it)."""
    n = 400000
    rng = np.random.default_rng(int(rtol * 1e3))
    inc = np.deg2rad(rng.uniform(89.7, 89.999, n))
    cam = 30.0 * np.stack([np.sin(inc), np.zeros(n), np.cos(inc)], -1)
    ph, R = rng.uniform(0.0, 2.0 * np.pi, n), 17.0 + rng.uniform(-0.3, 0.3, n)
    k = np.stack([R * np.cos(ph), R * np.sin(ph), rng.normal(0.0, 0.01, n)], -1) - cam
    k /= np.linalg.norm(k, axis=1)[:, None]
    for r_in, r_out in ((4.5, 17.0), (17.0, 35.0)):
        kw = dict(r_s=1.0, lambda_end=90.0, disk_r_in=r_in, disk_r_out=r_out, rtol=rtol, atol=rtol * 1e-3, rhs_form=1)
        o = oracle.trace(k, cam, **kw)
        end, flags, steps, acc = ctx.trace(k, cam, _params(**kw))
        lost = (o["flags"] == 128) & (flags != 128)
        assert not lost.any(), f"{int(lost.sum())} disk hits lost (annulus {r_in} .. {r_out}, rtol {rtol})"
        assert np.array_equal(flags, o["flags"])
        # step counts: identical, except that a ray that ends ON the horizon at these loose tolerances takes a dozen
        # rejected steps next to the coordinate singularity, where accept / reject is rounding (the module docstring's
        # exception; measured: 1 ray of 400,000, with or without a disk)
        diff = (steps != o["n_attempted"]) | (acc != o["n_accepted"])
        assert diff.sum() <= 5 and np.all((flags[diff] & 1) != 0), (int(diff.sum()), flags[diff])
        hit = flags == 128
        assert hit.sum() > 50000
        # end states: a plane crossing is located as sharply as the ray is steep (d * steep is the measure); the few rays
        # that dive to R < 6 and wind around the hole on the way amplify rounding like everywhere else (_compare's S_i):
        # measured 5e-17 median, 2e-13 at the 99th percentile, 1.6e-11 at the 99.99th, 6.8e-8 worst (a 23-step ray ending at R = 4.8)
        steep = np.abs(o["end"][hit, 5]) / np.linalg.norm(o["end"][hit, 3:6], axis=1)
        ds = np.abs(end[hit] - o["end"][hit]).max(1) * steep
        assert np.quantile(ds, 0.99) < 1e-11 and np.quantile(ds, 0.9999) < 1e-9 and ds.max() < 1e-6


def test_grazing_disk_crossings_kerr(ctx, oracle):
    """The Boyer-Lindquist form of the same filter (theta = pi/2 plane, annulus in sqrt(r^2 + a^2)): flags against the
    oracle on near-equatorial cameras, to the Kerr fuzz test's own tolerance for flag differences."""
    k, cam = _grazing_rays(6000, 77, sigma=(9.0, 9.0, 0.1), inc_lo=75.0)
    cam = cam + np.array([0.0, 3.0, 0.0])        # off the phi = 0 half-plane
    for rtol in (1e-3, 1e-2):
        kw = dict(r_s=1.0, spin=0.45, rhs_form=2, lambda_end=90.0, disk_r_in=3.0, disk_r_out=35.0, rtol=rtol, atol=rtol * 1e-3)
        o = oracle.trace(k, cam, **kw)
        end, flags, steps, acc = ctx.trace(k, cam, _params(**kw))
        diff = flags != o["flags"]
        assert diff.mean() <= 0.002, (rtol, int(diff.sum()))
        # no disk hit may be LOST other than through such a flip: count them apart
        lost = (o["flags"] == 128) & (flags != 128)
        gained = (o["flags"] != 128) & (flags == 128)
        assert lost.sum() <= max(2, 3 * gained.sum() + 2), (rtol, int(lost.sum()), int(gained.sum()))
        same = ~diff & (steps == o["n_attempted"]) & (o["flags"] == 128)
        # end states of the agreeing disk hits (measured at rtol 1e-2: median 1e-12, 99 % 1.7e-9, 99.9 % 1.4e-7, worst 2e-6 --
        # the tail is the rays that wind around the hole first; rtol 1e-3: 4e-13 / 1.8e-10 / 1.4e-9 / 3e-8)
        d = np.abs(end[same] - o["end"][same]).max(1)
        assert same.sum() > 1000 and np.median(d) < 1e-11 and np.quantile(d, 0.99) < 1e-8 and np.quantile(d, 0.999) < 1e-6 and d.max() < 1e-5


@pytest.mark.parametrize("rhs_form", [0, 1])
def test_rk4_fixed_step(ctx, oracle, rhs_form):
    k = frame_rays(3000, seed=28)
    _compare(ctx, oracle, k, CAM, r_s=1.0, lambda_end=50.0, method=1, h_fixed=0.1, rhs_form=rhs_form)
    _compare(ctx, oracle, k[:200], CAM, r_s=1.0, lambda_end=50.0, method=1, h_fixed=0.37, rhs_form=rhs_form, r_exit=40.0)


# ---- Kerr, Boyer-Lindquist (BASELINE.json config 5) ---------------------------------------------------
def test_kerr_golden_and_oracle(ctx, oracle):
    g = load_golden("kerr_a09")
    kw = dict(r_s=1.0, lambda_end=60.0, rhs_form=2, spin=float(g["spin"]))
    end, flags, steps, d = _compare(ctx, oracle, g["k0"][48:], g["x0"][48:], **kw)   # off-axis camera
    assert np.array_equal(flags, g["flags"][48:]) and np.array_equal(steps, g["n_attempted"][48:])
    assert np.abs(end - g["end"][48:]).max() < 1e-5
    # the reference's near-axis camera (x = 1e-4): flags and step counts still identical
    o = oracle.trace(g["k0"][:48], g["x0"][:48], **kw)
    e2, f2, s2, a2 = ctx.trace(g["k0"][:48], g["x0"][:48], _params(**kw))
    assert np.array_equal(f2, o["flags"]) and np.array_equal(f2, g["flags"][:48])
    assert (s2 == o["n_attempted"]).mean() > 0.9


def test_kerr_seeded_rays_and_rk4(ctx, oracle):
    cam = np.array([4.0, -24.0, 13.0])
    rng = np.random.default_rng(41)
    k = (-cam / np.linalg.norm(cam))[None, :] + rng.normal(size=(20000, 3)) * 0.12
    k /= np.linalg.norm(k, axis=1)[:, None]
    # (20,000 rays: measured 0 rays with a different step count while the start states came from a prepare pass with
    # the checker's own order of operations, 1 ray since the trace waves convert them in their queue fill -- and 23
    # against 29 of 20,000 from the reference's camera next to the rotation axis, where both forms flip alike)
    end, flags, steps, d = _compare(ctx, oracle, k, cam, r_s=1.0, lambda_end=60.0, rhs_form=2, spin=0.45, step_flips=4)
    assert 0.02 < ((flags & 1) != 0).mean() < 0.5 and np.median(d) < 1e-9
    _compare(ctx, oracle, k[:3000], cam, r_s=1.0, lambda_end=60.0, rhs_form=2, spin=-0.3, r_exit=35.0)
    # fixed steps h = 0.1 through the 1/Delta singularity of the Boyer-Lindquist Christoffels: horizon rays
    # are rounding-sensitive there (which step first lands below r_plus(1+margin), or first turns NaN, flips
    # for ~1/3 of them); escaping rays must still agree step for step
    _compare(ctx, oracle, k[:2000], cam, r_s=1.0, lambda_end=40.0, rhs_form=2, spin=0.45, method=1, h_fixed=0.1, allow_flips=0.15)
    # start inside the (margin) horizon, and argument checks
    e, f, s_, a_ = ctx.trace(k[:4], np.array([0.3, 0.2, 0.6]), _params(r_s=1.0, rhs_form=2, spin=0.45))
    assert np.all(f == 3)
    from blackhole_geodesic_calculator_amd import _ffi
    with pytest.raises(_ffi.BhgError):
        ctx.trace(k[:4], cam, _params(r_s=1.0, rhs_form=2, spin=0.5))          # |a| must stay below M


def test_kerr_integrator_and_camera_adaptors(ctx, oracle):
    from blackhole_geodesic_calculator_amd import GeodesicIntegratorKerr
    from blackhole_geodesic_calculator_amd.camera import RelativisticCamera
    gi = GeodesicIntegratorKerr(mass=0.5, a=0.9, context=ctx)
    assert abs(gi.spin - 0.45) < 1e-15 and abs(gi.r_plus - (0.5 + (0.25 - 0.45**2) ** 0.5)) < 1e-15
    cam = RelativisticCamera(resolution=[24, 32], field_of_view=[0.6, 0.6], a=0.9, M=0.5,
                             camera_location=[0.0, -25.0, 12.0], camera_rotation_euler=(1.1, 0.0, 0.0), integrator=gi)
    cam.run()
    o = oracle.trace(cam.pixel_directions().reshape(-1, 3), cam.camera_location, r_s=1.0, lambda_end=50.0, rhs_form=2, spin=0.45)
    assert np.array_equal(cam.ray_blackhole_hit.reshape(-1), (o["flags"] & 1).astype(np.uint8))


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("BHG_FUZZ", "16"))))
def test_randomised_configurations(ctx, oracle, seed, record_property):
    """Fuzz: random metric size, camera, tolerances, step caps and optional events, all on one code
    path per draw; every draw must agree with the oracle like the fixed cases do."""
    rng = np.random.default_rng(1000 + seed)
    r_s = float(rng.choice([0.0, 0.3, 1.0, 2.5]))
    dist_cam = float(rng.uniform(3.0, 60.0)) * max(r_s, 0.5)
    cam = rng.normal(size=3)
    cam = dist_cam * cam / np.linalg.norm(cam)
    n = int(rng.integers(1, 3000))
    aim = rng.normal(size=(n, 3)) * max(r_s, 0.5) * float(rng.uniform(1.0, 8.0))
    k = aim - cam
    k /= np.linalg.norm(k, axis=1)[:, None]
    if rng.random() < 0.3:  # per-ray origins
        x0 = cam + rng.normal(size=(n, 3)) * 0.1 * dist_cam
    else:
        x0 = cam
    kw = dict(r_s=r_s, lambda_end=float(rng.uniform(0.5, 4.0)) * dist_cam, rhs_form=int(rng.integers(0, 2)))
    mode = int(rng.integers(0, 4))
    if mode == 0:
        kw.update(rtol=float(10 ** rng.uniform(-7, -2)), atol=float(10 ** rng.uniform(-10, -4)))
    elif mode == 1:
        kw.update(max_step=float(rng.uniform(0.05, 2.0)) * max(r_s, 0.5))
    elif mode == 2:
        kw.update(method=1, h_fixed=float(rng.uniform(0.05, 0.5)) * max(r_s, 0.5))
    if rng.random() < 0.4:
        kw["r_exit"] = float(rng.uniform(0.5, 1.5)) * dist_cam
    if rng.random() < 0.4:
        a = float(rng.uniform(1.5, 6.0)) * max(r_s, 0.5)
        kw.update(disk_r_in=a, disk_r_out=a * float(rng.uniform(1.1, 3.0)))
    if rng.random() < 0.2:
        kw["max_steps"] = int(rng.integers(1, 40))
    if seed % 4 == 3 and r_s > 0.0 and kw.get("method", 0) == 0:
        # geometry mode "near-equatorial camera + wide disk", tolerances up to 1e-1: plane crossings that graze, long
        # steps -- the regime of the disk pre-filter's bound (drawn after everything else: the other draws are unchanged)
        inc = np.deg2rad(np.where(rng.random(n) < 0.5, rng.uniform(89.0, 89.999, n), rng.uniform(60.0, 89.999, n)))
        phi = float(rng.uniform(0.0, 2.0 * np.pi))
        x0 = dist_cam * np.stack([np.sin(inc) * np.cos(phi), np.sin(inc) * np.sin(phi), np.cos(inc)], -1)
        k = rng.normal(size=(n, 3)) * np.array([0.3, 0.3, 0.001]) * dist_cam - x0
        k /= np.linalg.norm(k, axis=1)[:, None]
        rtol = float(10 ** rng.uniform(-3, -1))
        a = float(rng.uniform(1.5, 4.0)) * r_s
        kw.update(rtol=rtol, atol=rtol * 1e-3, disk_r_in=a, disk_r_out=a * float(rng.uniform(2.0, 10.0)),
                  lambda_end=3.0 * dist_cam)
        kw.pop("max_step", None)
    if seed % 5 == 4 and kw["rhs_form"] == 0:
        # massive particles (time_like=True): the same draw with g(k, k) = -1 and speeds 0.05 ... 1.5 (drawn last again)
        kw["time_like"] = 1
        k = k * rng.uniform(0.05, 1.5, (n, 1))
    tight = kw.get("rtol", 1e-3) <= 1e-6 and kw["rhs_form"] == 0
    _compare(ctx, oracle, k, x0, allow_flips=(0.02 if tight else False), outliers=2e-3, rounding_flips=2, **kw)
    # on record per draw (junit property / -rA): how many rays needed the sensitivity-scaled term at all, how many it let through
    for key, val in LAST_COMPARE.items():
        record_property(key, val)
    print(f"fuzz draw {seed}: {LAST_COMPARE}")
    # (measured, round 3, 16 draws: 14 draws with no such ray, one with 1 of 564, one -- rtol 1e-7 next to the photon
    # sphere -- with 53 of 919, worst 1.0e-8; none beyond the scaled bound.  300 draws, BHG_FUZZ=300: the same picture
    # for rays that end at lambda_end or on an event; rays cut off by max_steps differ by up to 5.5e-6 -- their end
    # lambda is not pinned, see _compare -- and are recorded apart.)
    # (round 4, 240 draws with the grazing-geometry mode in: 42 draws have a ray beyond the floor, none beyond the scaled
    # bound; one draw -- 231 rays at rtol 0.05 winding around the photon sphere -- differs by 1.4e-3 on a ray whose own
    # sensitivity is 5e-4: large, and 2.8 S_i)
    assert LAST_COMPARE["beyond_floor"] <= max(3, 0.10 * LAST_COMPARE["rays"]), LAST_COMPARE
    assert LAST_COMPARE["worst"] < 1e-6 or LAST_COMPARE["worst_multiple_of_sensitivity"] < 50.0, LAST_COMPARE


def test_kerr_disk_golden_and_frames(ctx, oracle):
    """Kerr a/M = 0.9 with the thin disk in the equatorial plane (rays resumed from Boyer-Lindquist records inside the launch)."""
    g = load_golden("kerr_disk")
    kw = dict(r_s=1.0, lambda_end=80.0, rhs_form=2, spin=float(g["spin"]), disk_r_in=3.0, disk_r_out=10.0)
    end, flags, steps, acc = ctx.trace(g["k0"], g["x0"], _params(**kw))
    assert np.array_equal(flags, g["flags"]) and np.array_equal(acc, g["n_accepted"])
    assert np.all(np.abs(end - g["end"]).max(1) < np.where((g["flags"] & 1) != 0, 1e-5, 1e-8))
    # seeded frames from three inclinations against the oracle (ids of the rays that differ: horizon / axis only)
    for inc, seed in ((1.45, 81), (1.0, 82), (0.3, 83)):
        cam = np.array([30 * np.sin(inc), 0.3, 30 * np.cos(inc)])
        aim = np.random.default_rng(seed).normal(size=(6000, 3)) * np.array([9.0, 9.0, 1.5])
        k = aim - cam
        k /= np.linalg.norm(k, axis=1)[:, None]
        kw2 = dict(kw, r_exit=40.0)
        o = oracle.trace(k, cam, **kw2)
        end, flags, steps, acc = ctx.trace(k, cam, _params(**kw2))
        same = (flags == o["flags"]) & (steps == o["n_attempted"]) & (acc == o["n_accepted"])
        assert (flags != o["flags"]).mean() <= 0.002 and (~same).mean() <= 0.03
        assert np.all((o["flags"][~same] & 1) != 0) or (~same).sum() <= 12
        d = np.abs(end - o["end"]).max(1)
        on = same & (o["flags"] == 128)
        assert on.sum() > 1000 and np.median(d[on]) < 1e-10 and d[on].max() < 1e-6
        assert np.abs(end[on, 2]).max() < 1e-9


def test_objects_golden_vectors(ctx, oracle):
    g = load_golden("objects")
    kw = dict(r_s=1.0, lambda_end=70.0, max_step=0.25, r_exit=35.0, disk_r_in=3.0, disk_r_out=7.0, spheres=g["spheres"])
    end, flags, steps, d = _compare(ctx, oracle, g["k0"], g["x0"], **kw)
    assert np.array_equal(flags, g["flags"]) and np.abs(end - g["end"]).max() <= 1e-8


def _analytic_sphere_hits(cam, k, spheres, lam_end):
    best = np.full(len(k), np.inf)
    bid = np.full(len(k), -1)
    for j, (cx, cy, cz, rho) in enumerate(spheres):
        oc = cam - np.array([cx, cy, cz])
        b = k @ oc
        disc = b * b - (oc @ oc - rho * rho)
        t = np.where(disc > 0, -b - np.sqrt(np.maximum(disc, 0)), np.inf)
        t = np.where(t > 0, t, np.inf)
        upd = t < best
        best = np.where(upd, t, best)
        bid = np.where(upd, j, bid)
    hit = np.isfinite(best) & (best < lam_end)
    return hit, np.where(hit, bid, -1), cam + np.where(hit, best, 0.0)[:, None] * k


@pytest.mark.parametrize("method", [0, 1])
def test_objects_flat_space_is_exact_ray_sphere_intersection(ctx, oracle, method):
    """r_s = 0: the curves are straight lines, every DP5(4) step is exact and as long as the controller
    allows, so most spheres are passed THROUGH inside one step (the chord rule), and the entry points
    are the textbook ray-sphere intersections."""
    rng = np.random.default_rng(70)
    n = 20000
    k = np.stack([rng.uniform(-0.3, 0.3, n), rng.uniform(-0.3, 0.3, n), -np.ones(n)], 1)
    k /= np.linalg.norm(k, axis=1)[:, None]
    sph = [(1.0, 0.5, 10.0, 1.5), (-2.0, 1.0, 0.0, 2.0), (0.5, -3.0, -12.0, 1.0), (3.0, 3.0, 5.0, 0.7)]
    kw = dict(r_s=0.0, lambda_end=60.0, rhs_form=1, method=method, h_fixed=0.37)
    end, flags, steps, acc, obj = ctx.trace(k, CAM, _params(**kw), spheres=sph)
    hit, bid, pos = _analytic_sphere_hits(CAM, k, sph, 60.0)
    assert hit.sum() > 1000
    assert np.array_equal((flags == 0x88), hit) and np.array_equal(obj, bid)
    assert np.all(flags[~hit] == 4)
    assert np.abs(end[hit, 0:3] - pos[hit]).max() < 1e-11
    assert np.abs(end[hit, 3:6] - k[hit]).max() < 1e-14
    _compare(ctx, oracle, k[:3000], CAM, spheres=sph, **kw)


@pytest.mark.parametrize("rhs_form", [0, 1])
@pytest.mark.parametrize("regime", ["default", "fine", "rk4"])
def test_objects_in_curved_space(ctx, oracle, rhs_form, regime):
    """Spheres beside, in front of and behind the hole (one of them straddling the disk plane), together with
    the exit sphere and the disk: flags, ids, step counts and entry points against the oracle."""
    k = frame_rays(12000, seed=71, fov=0.6)
    sph = [(2.0, 1.0, 8.0, 1.5), (-3.0, 0.5, -1.0, 1.2), (1.0, -4.0, -10.0, 2.0), (0.0, 3.5, 2.0, 0.8), (4.5, 0.0, 0.0, 1.0)]
    kw = dict(r_s=1.0, lambda_end=70.0, rhs_form=rhs_form, r_exit=35.0, disk_r_in=3.0, disk_r_out=7.0, spheres=sph)
    if regime == "fine":
        kw["max_step"] = 0.2
    elif regime == "rk4":
        kw.update(method=1, h_fixed=0.1)
    end, flags, steps, d = _compare(ctx, oracle, k, CAM, **kw)
    kinds = {int(f): int((flags == f).sum()) for f in np.unique(flags)}
    assert kinds.get(0x88, 0) > 200 and kinds.get(128, 0) > 50 and kinds.get(8, 0) > 1000 and kinds.get(1, 0) > 10


def test_objects_host_api_and_validation(ctx):
    from blackhole_geodesic_calculator_amd import GeodesicIntegratorKerr, GeodesicIntegratorSchwarzschild, _ffi
    gi = GeodesicIntegratorSchwarzschild(mass=0.5)
    k = frame_rays(64 * 64, seed=72, fov=0.3).reshape(64, 64, 3)
    out = gi.trace(k, CAM, curve_end=60.0, spheres=[[0.0, 0.0, 12.0, 1.0]])
    assert out["object_id"].shape == (64, 64) and out["object_id"].dtype == np.int8
    hit = out["flags"] == _ffi.FLAG_HIT_OBJECT
    assert hit.sum() > 0 and np.all(out["object_id"][hit] == 0) and np.all(out["object_id"][~hit] == -1)
    assert np.all(np.abs(np.linalg.norm(out["ray_end"][hit][:, 0:3] - np.array([0.0, 0.0, 12.0]), axis=1) - 1.0) < 1e-9)
    # no spheres given: same as the plain call
    base = gi.trace(k, CAM, curve_end=60.0)
    none = gi.trace(k, CAM, curve_end=60.0, spheres=[])
    assert np.array_equal(base["ray_end"], none["ray_end"]) and np.all(none["object_id"] == -1)
    for bad in ([[0, 0, 5, 0.0]], [[0, 0, 5, -1.0]], [[np.nan, 0, 5, 1.0]], [[0, 0, 5, 1.0]] * 9):
        with pytest.raises(_ffi.BhgError):
            gi.trace(k, CAM, spheres=bad)
    # (Kerr takes object spheres since round 4: test_kerr_object_spheres_golden_and_oracle)
    kerr = GeodesicIntegratorKerr(mass=0.5, a=0.3, context=ctx).trace(k, CAM + np.array([0.0, 0.3, 0.0]), spheres=[[0.0, 0.0, 12.0, 1.0]])
    assert (kerr["flags"] == _ffi.FLAG_HIT_OBJECT).sum() > 0


@pytest.mark.parametrize("seed", range(max(4, int(__import__("os").environ.get("BHG_FUZZ", "48")) // 3)))
def test_randomised_objects(ctx, oracle, seed):
    rng = np.random.default_rng(9000 + seed)
    r_s = float(rng.choice([0.0, 0.5, 1.0, 2.0]))
    u = max(r_s, 0.5)
    dist_cam = float(rng.uniform(6.0, 40.0)) * u
    cam = rng.normal(size=3)
    cam = dist_cam * cam / np.linalg.norm(cam)
    n = int(rng.integers(1, 4000))
    aim = rng.normal(size=(n, 3)) * u * float(rng.uniform(2.0, 10.0))
    k = aim - cam
    k /= np.linalg.norm(k, axis=1)[:, None]
    ns = int(rng.integers(1, 9))
    sph = []
    for _ in range(ns):
        c = rng.normal(size=3)
        c = c / np.linalg.norm(c) * float(rng.uniform(1.5, 12.0)) * u
        sph.append([c[0], c[1], c[2], float(rng.uniform(0.1, 2.5)) * u])   # may overlap each other, the hole, the camera
    kw = dict(r_s=r_s, lambda_end=float(rng.uniform(1.0, 3.0)) * dist_cam, rhs_form=int(rng.integers(0, 2)), spheres=sph)
    mode = int(rng.integers(0, 4))
    if mode == 0:
        kw.update(rtol=float(10 ** rng.uniform(-6, -2)), atol=float(10 ** rng.uniform(-9, -4)))
    elif mode == 1:
        kw.update(max_step=float(rng.uniform(0.05, 2.0)) * u)
    elif mode == 2:
        kw.update(method=1, h_fixed=float(rng.uniform(0.05, 0.5)) * u)
    if rng.random() < 0.4:
        kw["r_exit"] = float(rng.uniform(0.5, 1.5)) * dist_cam
    if rng.random() < 0.4:
        a = float(rng.uniform(1.5, 6.0)) * u
        kw.update(disk_r_in=a, disk_r_out=a * float(rng.uniform(1.1, 3.0)))
    tight = kw.get("rtol", 1e-3) <= 1e-6 and kw["rhs_form"] == 0
    _compare(ctx, oracle, k, cam, allow_flips=(0.02 if tight else False), outliers=2e-3, rounding_flips=2, **kw)


@pytest.mark.parametrize("seed", range(max(4, int(__import__("os").environ.get("BHG_FUZZ", "48")) // 4)))
def test_randomised_kerr(ctx, oracle, seed, record_property):
    rng = np.random.default_rng(5000 + seed)
    r_s = float(rng.choice([0.6, 1.0, 2.0]))
    spin = float(rng.uniform(-0.98, 0.98)) * 0.5 * r_s
    dist_cam = float(rng.uniform(6.0, 50.0)) * r_s
    cam = rng.normal(size=3)
    cam[2] *= 0.7
    cam = dist_cam * cam / np.linalg.norm(cam)
    if abs(cam[0]) + abs(cam[1]) < 0.05 * dist_cam:  # keep off the polar axis (coordinate singularity)
        cam[0] += 0.2 * dist_cam
    n = int(rng.integers(1, 2500))
    aim = rng.normal(size=(n, 3)) * r_s * float(rng.uniform(1.0, 6.0))
    k = aim - cam
    k /= np.linalg.norm(k, axis=1)[:, None]
    kw = dict(r_s=r_s, spin=spin, rhs_form=2, lambda_end=float(rng.uniform(1.0, 3.0)) * dist_cam)
    mode = int(rng.integers(0, 3))
    if mode == 0:
        kw.update(rtol=float(10 ** rng.uniform(-6, -2)), atol=float(10 ** rng.uniform(-9, -4)))
    elif mode == 1:
        kw.update(max_step=float(rng.uniform(0.1, 2.0)) * r_s)
    if rng.random() < 0.4:
        kw["r_exit"] = float(rng.uniform(0.6, 1.4)) * dist_cam
    if rng.random() < 0.2:
        kw["max_steps"] = int(rng.integers(1, 60))
    if rng.random() < 0.35:
        rin = float(rng.uniform(1.5, 5.0)) * r_s
        kw.update(disk_r_in=rin, disk_r_out=rin * float(rng.uniform(1.2, 3.0)))
    if seed % 5 == 4:
        kw["time_like"] = 1       # massive particles, speeds 0.05 ... 1.5
        k = k * rng.uniform(0.05, 1.5, (len(k), 1))
    # Boyer-Lindquist coordinates are singular on the horizon (1/Delta) and on the polar axis (cot theta):
    # rays that end on the horizon or pass close to the axis (small L_z) have rounding-sensitive step
    # sequences in the oracle and on the GPU alike.  Everything else must agree step for step.
    o = oracle.trace(k, cam, **kw)
    end, flags, steps, acc = ctx.trace(k, cam, _params(**kw))
    assert (flags != o["flags"]).mean() <= 0.002
    same = (steps == o["n_attempted"]) & (acc == o["n_accepted"]) & (flags == o["flags"])
    from oracle import scipy_reference as sr
    Lz = np.array([sr.kerr_constants(*sr.cart_to_bl(cam, kk, spin), 0.5 * r_s, spin, float(kw.get("time_like", 0)))[1] for kk in k[~same]])
    # (a horizon ray is one EITHER side ends on the horizon or in NaN: with a step budget, max_steps, the two can stop one
    # accept / reject decision apart -- MAX_STEPS here, HIT_HORIZON there)
    hor_all = ((flags | o["flags"]) & (1 | 64)) != 0
    hor = hor_all[~same]
    touchy = hor | (np.abs(Lz) < 0.3 * r_s)
    rec = dict(rays=len(k), differ=int((~same).sum()), differ_fraction=float((~same).mean()), touchy_fraction=float(touchy.mean()) if len(Lz) else 1.0,
               horizon_fraction=float(hor_all.mean()), differ_horizon=int(hor.sum()), horizon_rays=int(hor_all.sum()),
               differ_other=int((~hor).sum()), differ_neither=int((~touchy).sum()))
    print("kerr fuzz", seed, rec)
    for k_, v_ in rec.items():
        record_property(k_, v_)
    # (bounds and what was measured: KERR_FUZZ_DIFFER* at the top of this file; a ray that is neither a horizon ray nor near
    # the axis may flip too -- one in 1.5 M of them did, 19 of 5.24 M on the off-axis frame of test_gpu_fullsize -- but not two)
    assert hor.sum() <= max(3, KERR_FUZZ_DIFFER_HORIZON * hor_all.sum()), rec
    assert (~hor).sum() <= max(3, KERR_FUZZ_DIFFER_OTHER * len(k)), rec
    assert (~same).sum() <= max(3, KERR_FUZZ_DIFFER * len(k), KERR_FUZZ_DIFFER_HORIZON * hor_all.sum()), rec
    assert (~touchy).sum() <= 1, rec
    d = np.abs(end - o["end"]).max(1)
    tol = 1e-9 + 1e4 * _sensitivity(oracle, k, cam, o["end"], **kw) + np.where((o["flags"] & 1) != 0, 1e-5, 0.0)
    ok = same & np.isfinite(o["end"]).all(1)
    assert (d[ok] > tol[ok]).mean() <= 0.01
    # the typical agreement, over rays that do not end on the horizon (there the Boyer-Lindquist end state is
    # singular: u^r, u^phi grow without bound and differences of 1e-6 are rounding, see the tolerance above)
    away = ok & ((o["flags"] & 1) == 0)
    assert np.median(d[away]) < 1e-7 if away.any() else True


def test_gpu_error_against_converged_solution(ctx, oracle):
    """T2: the GPU at tolerance rtol sits within C*rtol of a converged solution, C measured (escaping rays
    of the config-2 frame that keep clear of the photon sphere, b >= 3 r_s)."""
    k = frame_rays(20000, seed=61)
    b = 30.0 * np.hypot(k[:, 0], k[:, 1]) / np.abs(k[:, 2])
    k = k[b >= 3.0]
    conv = oracle.trace(k, CAM, r_s=1.0, lambda_end=50.0, rtol=1e-12, atol=1e-14, rhs_form=1)
    assert np.all(conv["flags"] == 4)
    worst = {}
    for rtol in (1e-3, 1e-5, 1e-7):
        end, flags, _, _ = ctx.trace(k, CAM, _params(r_s=1.0, lambda_end=50.0, rtol=rtol, atol=rtol * 1e-3))
        assert np.all(flags == 4)
        worst[rtol] = np.abs(end - conv["end"]).max() / rtol
    # measured C = max |gpu - converged| / rtol over 18 191 rays with b >= 3 r_s (lengths are O(30)):
    # 6.3e3 at the loose default (errors up to ~6 just outside the photon sphere -- what scipy's default
    # tolerances deliver), 80 at 1e-5, 34 at 1e-7
    assert worst[1e-3] < 2e4 and worst[1e-5] < 400 and worst[1e-7] < 200, worst


@pytest.mark.parametrize("case", ["config2", "exit+disk", "fine", "reduced tight", "per-ray origins"])
def test_gpu_against_live_scipy(ctx, case):
    """The GPU against scipy.integrate.solve_ivp ITSELF (oracle/scipy_reference.py: the driver the golden vectors were made
    with), on fresh seeded rays, no C restatement in between: flags, attempted and accepted step counts identical, end
    states within the stated tolerances of section 2 of DESIGN.md."""
    from oracle import scipy_reference as sr
    rng = np.random.default_rng({"config2": 1, "exit+disk": 2, "fine": 3, "reduced tight": 4, "per-ray origins": 5}[case])
    form, skw, gkw, x0 = "christoffel", {}, {}, CAM
    if case == "config2":
        k = frame_rays(250, seed=int(rng.integers(1 << 30)))
    elif case == "exit+disk":
        inc = np.radians(80.0)
        x0 = np.array([30 * np.sin(inc), 0.0, 30 * np.cos(inc)])
        aim = rng.normal(size=(150, 3)) * np.array([8.0, 8.0, 0.5])
        k = aim - x0
        k /= np.linalg.norm(k, axis=1)[:, None]
        skw, gkw = dict(r_exit=40.0, disk=(4.5, 10.5), lambda_end=80.0), dict(r_exit=40.0, disk_r_in=4.5, disk_r_out=10.5, lambda_end=80.0)
    elif case == "fine":
        k = frame_rays(16, seed=int(rng.integers(1 << 30)))
        skw = gkw = dict(max_step=0.1)
    elif case == "reduced tight":
        k = frame_rays(60, seed=int(rng.integers(1 << 30)), fov=0.3)
        form, skw, gkw = "reduced", dict(rtol=1e-8, atol=1e-11), dict(rtol=1e-8, atol=1e-11, rhs_form=1)
    else:
        k = frame_rays(120, seed=int(rng.integers(1 << 30)))
        x0 = CAM + rng.normal(size=(120, 3)) * 2.0
    skw = dict(dict(r_s=1.0, lambda_end=50.0), **skw)
    gkw = dict(dict(r_s=1.0, lambda_end=50.0), **gkw)
    ref = sr.trace_rays(k, x0, form=form, **skw)
    end, flags, steps, acc = ctx.trace(k, x0, _params(**gkw))
    assert np.array_equal(flags, ref["flags"]) and np.array_equal(acc, ref["n_accepted"])
    # (a ray that ends ON THE DISK: solve_ivp's disk event is not terminal -- the crossings are sorted out afterwards -- so
    # scipy's count of attempted steps runs past the hit and is not comparable; the accepted steps up to the hit are)
    cmp_att = flags != 128
    assert np.array_equal(steps[cmp_att], ref["n_attempted"][cmp_att])
    d = np.abs(end - ref["end"]).max(1)
    hor = (flags & 1) != 0
    assert len(np.unique(flags)) >= (1 if case == "fine" else (3 if case == "exit+disk" else 2))
    assert d[~hor].max(initial=0.0) < 1e-8 and d[hor].max(initial=0.0) < 1e-4, (d[~hor].max(initial=0.0), d[hor].max(initial=0.0))


def test_gpu_sampled_curves_against_live_scipy(ctx):
    """Row a2's literal call -- calc_trajectory(..., nr_points_curve=T), RelativisticRenderEngine.py:293-294 -- on the GPU against
    solve_ivp(..., t_eval=linspace(0, curve_end, T)) itself: the same number of samples per ray (a ray that ends on an event
    yields the grid points up to the root, ivp.py:706-723), the same values to rounding, NaN behind them."""
    from oracle import scipy_reference as sr
    g = load_golden("fig5")
    cases = [(g["k0"], g["x0"], dict(r_s=1.0, lambda_end=60.0), 121),
             (frame_rays(24, seed=193), CAM, dict(r_s=1.0, lambda_end=50.0), 50),
             (frame_rays(24, seed=194, fov=0.25), CAM, dict(r_s=1.0, lambda_end=70.0, r_exit=31.0), 77),
             (frame_rays(12, seed=195, fov=0.25), CAM, dict(r_s=1.0, lambda_end=50.0, max_step=0.5, rhs_form=1), 33),
             (frame_rays(1, seed=196), CAM, dict(r_s=1.0, lambda_end=50.0), 10000)]      # the engine's literal call: one ray, 10 000 samples
    seen = {}
    for k, x0, kw, T in cases:
        k = np.atleast_2d(k)
        tr, nv, end, fl = ctx.trajectory(k, x0, _params(**kw), T)
        for i in range(len(k)):
            xi = x0 if np.ndim(x0) == 1 else x0[i]
            r = sr.trace_ray(k[i], xi, r_s=kw["r_s"], lambda_end=kw["lambda_end"], max_step=kw.get("max_step", np.inf),
                             form="reduced" if kw.get("rhs_form") == 1 else "christoffel", r_exit=kw.get("r_exit", 0.0),
                             nr_points_curve=T)
            sol = r["sol"]
            assert int(fl[i]) == r["flags"], (i, int(fl[i]), r["flags"])
            seen[int(fl[i])] = seen.get(int(fl[i]), 0) + 1
            m = sol.y.shape[1]
            assert nv[i] == m, (i, int(nv[i]), m)
            want = np.stack([sol.y[1], sol.y[3], sol.y[5], sol.y[0], sol.y[2], sol.y[4]])
            d = np.abs(tr[i, :, :m] - want).max()
            assert d < (1e-8 if not (int(fl[i]) & 1) else 1e-5), (i, d)
            assert np.isnan(tr[i, :, m:]).all()
    assert seen.get(1, 0) >= 5 and seen.get(4, 0) >= 20 and seen.get(8, 0) >= 10, seen


def test_gpu_kerr_against_live_scipy(ctx):
    """Config 5's metric against solve_ivp on the sympy-generated Boyer-Lindquist right-hand side
    (oracle/scipy_reference.py trace_ray_kerr), off the polar axis: flags and accepted steps identical, attempted steps
    identical, end states of escaping rays within the stated 5e-8 (horizon rays end at the coordinate singularity)."""
    from oracle import scipy_reference as sr
    inc = np.radians(60.0)
    cam = np.array([30 * np.sin(inc), 0.3, 30 * np.cos(inc)])
    rng = np.random.default_rng(11)
    aim = rng.normal(size=(120, 3)) * 3.0
    k = aim - cam
    k /= np.linalg.norm(k, axis=1)[:, None]
    ref = [sr.trace_ray_kerr(kk, cam, M=0.5, a=0.45, lambda_end=60.0) for kk in k]
    end, flags, steps, acc = ctx.trace(k, cam, _params(r_s=1.0, lambda_end=60.0, rhs_form=2, spin=0.45))
    rf = np.array([r["flags"] for r in ref], dtype=np.uint8)
    assert np.array_equal(flags, rf) and ((flags & 1) != 0).sum() >= 5 and (flags == 4).sum() >= 50
    differ = (steps != np.array([r["n_attempted"] for r in ref])) | (acc != np.array([r["n_accepted"] for r in ref]))
    assert differ.sum() <= 1, int(differ.sum())     # (3.6e-6 of the off-axis full-size frame's rays differ from the checker)
    d = np.abs(end - np.array([r["end"] for r in ref])).max(1)
    esc = (flags == 4) & ~differ
    assert d[esc].max() < 5e-8, d[esc].max()


def test_rtol_below_100_eps_is_raised_like_scipy_does(ctx, oracle):
    """scipy's validate_tol (_ivp/common.py:44-51) through the C ABI: rtol = 1e-15 is the solve at rtol = 100 eps, bit for bit
    -- for the trace and for the sampled curves -- and lands where the checker lands (tests/test_oracle.py has scipy itself)."""
    cam = np.array([0.5, 0.0, 8.0])
    k = frame_rays(64, seed=78, fov=0.9)
    kw = dict(r_s=1.0, lambda_end=12.0, atol=1e-30)
    lo = 100 * np.finfo(float).eps
    a = ctx.trace(k, cam, _params(rtol=1e-15, **kw))
    b = ctx.trace(k, cam, _params(rtol=lo, **kw))
    for x, y in zip(a, b):
        assert np.array_equal(x, y, equal_nan=True)
    assert not np.array_equal(ctx.trace(k, cam, _params(rtol=1e-13, **kw))[2], b[2])
    ta = ctx.trajectory(k[:3], cam, _params(rtol=1e-15, **kw), 50)
    tb = ctx.trajectory(k[:3], cam, _params(rtol=lo, **kw), 50)
    for x, y in zip(ta, tb):
        assert np.array_equal(x, y, equal_nan=True)
    o = oracle.trace(k, cam, rtol=1e-15, **kw)
    esc = (a[1] == 4) & (o["flags"] == 4)
    assert esc.sum() >= 30
    assert np.abs(a[0][esc] - o["end"][esc]).max() < 1e-7
    assert np.median(np.abs(a[2][esc].astype(float) / o["n_attempted"][esc] - 1.0)) < 0.02


def test_nonfinite_input_is_flagged_not_hung(ctx):
    k = frame_rays(130, seed=29)
    k[3] = np.nan
    k[77, 1] = np.inf
    end, flags, steps, _ = ctx.trace(k, CAM, _params())
    assert (flags[3] & 64) and (flags[77] & 64)
    ok = np.ones(130, bool)
    ok[[3, 77]] = False
    assert np.all((flags[ok] & 64) == 0) and np.isfinite(end[ok]).all()


def test_invalid_arguments_are_rejected(ctx):
    import ctypes as C
    from blackhole_geodesic_calculator_amd import _ffi
    k = frame_rays(4, seed=1)
    # bhg_trajectory is ONE launch: more rays than a launch takes (2^26: the kernels form result offsets in 32 bits) are
    # refused before anything is read -- the arrays handed over here are four rays long
    nv = np.zeros(4, np.uint32)
    tr = np.zeros((4, 6, 2))
    dp = C.POINTER(C.c_double)
    rc = _ffi.load().bhg_trajectory(ctx._h, C.byref(_params()), CAM.ctypes.data_as(dp), 1, k.ctypes.data_as(dp), (1 << 26) + 1, 2,
                                    tr.ctypes.data_as(dp), nv.ctypes.data_as(C.POINTER(C.c_uint32)), None, None)
    assert rc == _ffi.E_INVALID and "2^26" in _ffi.load().bhg_last_error().decode()
    for bad in (dict(r_s=-1.0), dict(rtol=0.0), dict(lambda_end=float("nan")), dict(method=7), dict(rhs_form=5),
                dict(max_step=0.0), dict(method=1, h_fixed=0.0)):
        with pytest.raises(_ffi.BhgError) as ei:
            ctx.trace(k, CAM, _params(**bad))
        assert ei.value.code == _ffi.E_INVALID


# ---- size-independent properties at BASELINE.json's full size (config 2: 1024 x 1024 x 5) -------
@pytest.fixture(scope="module")
def full_frame(ctx):
    from blackhole_geodesic_calculator_amd import camera_directions
    k0 = camera_directions(1024, 1024, 5, 0.6, 0.6, 42.0).reshape(-1, 3)
    end, flags, steps, acc = ctx.trace(k0, CAM, _params(r_s=1.0, lambda_end=50.0))
    return k0, end, flags, steps, acc


def test_full_frame_every_ray_matches_oracle(full_frame, oracle):
    """EVERY ray of the headline frame (1024 x 1024 x 5 = 5,242,880) against the oracle -- not a strided subsample (the
    oracle takes about a second for the frame on the box's cores): flags, attempted and accepted step counts identical
    ray for ray; end states of the escaping rays within the stated bound (STATED: 1e-8), horizon rays by quantiles."""
    k0, end, flags, steps, acc = full_frame
    o = oracle.trace(k0, CAM, r_s=1.0, lambda_end=50.0)
    nf, ns, na = int((flags != o["flags"]).sum()), int((steps != o["n_attempted"]).sum()), int((acc != o["n_accepted"]).sum())
    assert nf == 0 and ns == 0 and na == 0, (nf, ns, na)
    assert np.all(flags != 0) and np.all((flags & ~np.uint8(5)) == 0)  # every ray ended: horizon or lambda_end
    d = np.abs(end - o["end"]).max(1)
    assert np.median(d) <= 1e-11
    # escaped rays: the stated bound or the ray's own sensitivity (measured, round 5: worst 3.7e-9 of 5,000,441 -- not one ray
    # beyond the stated 1e-8); horizon rays (242,439): quantiles, see _horizon_class_ok
    from test_gpu_fullsize import _within_stated_or_sensitivity
    _within_stated_or_sensitivity(oracle, k0, None, flags, d, o["end"], dict(r_s=1.0, lambda_end=50.0), "config 2, every ray")
    esc = CLASS_OF["escaped"](flags)
    assert d[esc].max() <= STATED["escaped"][0]


def test_full_frame_order_independence(ctx, full_frame):
    """A ray's result must not depend on which lane / wave / batch traced it: permuting the
    input permutes the output bit for bit (exercises the lane-refill queue at full size)."""
    k0, end, flags, steps, acc = full_frame
    perm = np.random.default_rng(31).permutation(len(k0))
    e2, f2, s2, a2 = ctx.trace(k0[perm], CAM, _params(r_s=1.0, lambda_end=50.0))
    assert np.array_equal(f2, flags[perm]) and np.array_equal(s2, steps[perm]) and np.array_equal(a2, acc[perm])
    assert np.array_equal(e2, end[perm])


def test_full_frame_mirror_symmetry(full_frame):
    """The camera sits (almost) on the z axis: the horizon-hit fraction must match the shadow's
    analytic size to within the loose default tolerance, and step counts must be bounded."""
    k0, end, flags, steps, acc = full_frame
    assert steps.max() < 200 and steps[flags == 4].min() >= 1
    assert 0.03 < ((flags & 1) != 0).mean() < 0.08


def test_full_frame_conservation_tight(ctx):
    """L = x cross k and E are conserved along every escaping ray (tight tolerances)."""
    k0 = frame_rays(1 << 20, seed=33)
    end, flags, _, _ = ctx.trace(k0, CAM, _params(r_s=1.0, lambda_end=50.0, rtol=1e-10, atol=1e-12, rhs_form=1))
    esc = flags == 4
    L0 = np.cross(np.broadcast_to(CAM, k0.shape), k0)
    L1 = np.cross(end[:, 0:3], end[:, 3:6])
    assert np.abs(L1 - L0)[esc].max() < 1e-5
    frac = ((flags & 1) != 0).mean()
    # shadow of b_c = 2.598 r_s seen from r = 30: tan(alpha) ~ b_c sqrt(1 - r_s/r)/r; uniform rays in a 0.6 x 0.6 window
    alpha = math.asin(1.5 * math.sqrt(3.0) * math.sqrt(1 - 1 / 30.0) / 30.0)
    assert abs(frac - math.pi * math.tan(alpha) ** 2 / 0.36) < 2e-3


def test_device_buffer_entry_point_matches_host_entry_point(ctx, full_frame):
    import torch
    k0, end, flags, steps, acc = full_frame
    n = 200000
    dk = torch.from_numpy(k0[:n]).cuda()
    dend = torch.empty((n, 6), dtype=torch.float64, device="cuda")
    dfl = torch.empty(n, dtype=torch.uint8, device="cuda")
    dst = torch.empty(n, dtype=torch.int32, device="cuda")
    ctx.trace_device(_params(r_s=1.0, lambda_end=50.0), n, dk.data_ptr(), dend.data_ptr(), x0_shared=CAM,
                     d_flags=dfl.data_ptr(), d_n_steps=dst.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(dend.cpu().numpy(), end[:n])
    assert np.array_equal(dfl.cpu().numpy(), flags[:n])
    assert np.array_equal(dst.cpu().numpy().astype(np.uint32), steps[:n])
    del dk, dend, dfl, dst
    # the host-buffer call is a pipeline over 2^20-ray chunks (five here, the last one ragged) with the results
    # arriving in page-locked arrays; the whole frame in ONE device-resident call must give the same bits ...
    n = len(k0)
    dk = torch.from_numpy(k0).cuda()
    dend = torch.empty((n, 6), dtype=torch.float64, device="cuda")
    dfl = torch.empty(n, dtype=torch.uint8, device="cuda")
    dst = torch.empty(n, dtype=torch.int32, device="cuda")
    dac = torch.empty(n, dtype=torch.int32, device="cuda")
    ctx.trace_device(_params(r_s=1.0, lambda_end=50.0), n, dk.data_ptr(), dend.data_ptr(), x0_shared=CAM,
                     d_flags=dfl.data_ptr(), d_n_steps=dst.data_ptr(), d_n_accepted=dac.data_ptr(),
                     stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(dend.cpu().numpy(), end) and np.array_equal(dfl.cpu().numpy(), flags)
    assert np.array_equal(dst.cpu().numpy().astype(np.uint32), steps) and np.array_equal(dac.cpu().numpy().astype(np.uint32), acc)
    # ... and so must the same call into ordinary (pageable) numpy arrays through the staging ring, with only the
    # arrays asked for coming back
    m = 2 * (1 << 20) + 12345
    e2, f2, s2, a2 = ctx.trace(k0[:m], CAM, _params(r_s=1.0, lambda_end=50.0), want_steps=False, want_accepted=False,
                               pinned_results=False)
    assert s2 is None and a2 is None and np.array_equal(e2, end[:m]) and np.array_equal(f2, flags[:m])
    # (the whole set through the pageable form too: five chunks, the last one ragged)
    e4, f4, s4, a4 = ctx.trace(k0, CAM, _params(r_s=1.0, lambda_end=50.0), pinned_results=False)
    assert np.array_equal(e4, end) and np.array_equal(f4, flags) and np.array_equal(s4, steps) and np.array_equal(a4, acc)
    # per-ray origins take the same road
    x0 = np.broadcast_to(CAM, (m, 3)).copy()
    e3, f3, s3, a3 = ctx.trace(k0[:m], x0, _params(r_s=1.0, lambda_end=50.0), pinned_results=False)
    assert np.array_equal(e3, end[:m]) and np.array_equal(s3, steps[:m]) and np.array_equal(a3, acc[:m])


def test_calc_trajectory_adaptor(ctx, oracle):
    """The per-ray drop-in for RelativisticRenderEngine.py:293-313, with the sampled curve the
    reference requests (nr_points_curve) compared with the oracle's solve_ivp-style t_eval sampling."""
    from blackhole_geodesic_calculator_amd import GeodesicIntegratorSchwarzschild
    gi = GeodesicIntegratorSchwarzschild(mass=0.5, time_like=False, verbose=False, context=ctx)
    for b, npts in ((0.5, 10000), (2.0, 10000), (2.62, 777), (5.0, 10000)):
        k0 = np.array([b / 30.0, 0.0, -1.0])
        k0 /= np.linalg.norm(k0)
        k_xyz, x_xyz, result = gi.calc_trajectory(k0, CAM, max_step=np.inf, curve_end=50, nr_points_curve=npts, verbose=False)
        o = oracle.trace(k0[None], CAM, r_s=1.0, lambda_end=50.0)
        tr, nv, fl = oracle.trajectory(k0[None], CAM, npts, r_s=1.0, lambda_end=50.0)
        x, y, z = x_xyz                      # :299
        k_x, k_y, k_z = k_xyz                # :300
        assert result["start_inside_hole"] is False
        assert result["hit_blackhole"] == bool(o["flags"][0] & 1)
        assert len(x) == nv[0] and (len(x) == npts) == (not result["hit_blackhole"])
        assert np.abs(x_xyz - tr[0, 0:3, :nv[0]]).max() < 1e-8 and np.abs(k_xyz - tr[0, 3:6, :nv[0]]).max() < 1e-8
        assert np.array_equal(x_xyz[:, 0], CAM) and np.array_equal(k_xyz[:, 0], k0)
        # exact end state rides along; for escaping rays it is also the last sample (:307-308)
        assert np.abs(result["end_loc"] - o["end"][0, 0:3]).max() < 1e-8
        if not result["hit_blackhole"]:
            end_loc = np.array([x[-1], y[-1], z[-1]])
            end_dir = np.array([k_x[-1], k_y[-1], k_z[-1]])
            assert np.abs(end_loc - o["end"][0, 0:3]).max() < 1e-8 and np.abs(end_dir - o["end"][0, 3:6]).max() < 1e-8
    k_xyz, x_xyz, result = gi.calc_trajectory(np.array([0, 0, -1.0]), np.array([0.1, 0.1, 0.1]))
    assert result["start_inside_hole"] is True and result["hit_blackhole"] is True and x_xyz.shape == (3, 0)


def test_trajectory_kernel_shapes_give_the_same_bits(ctx):
    """bhg_trajectory runs one WAVE per ray up to 2048 rays (the engine's literal call: one ray, 10,000 samples, shared
    out over the 64 lanes) and one LANE per ray above: same arithmetic, so the same rays sampled either way must come
    back bit for bit, sample counts and NaN padding included; sample counts that are not multiples of 64 and a
    trajectory cut short by the horizon are in the set."""
    k = frame_rays(2100, seed=52)
    for kw, T in ((dict(r_s=1.0, lambda_end=50.0), 37), (dict(r_s=1.0, lambda_end=50.0, rhs_form=1, r_exit=31.0), 130),
                  (dict(r_s=1.0, lambda_end=40.0, rhs_form=2, spin=0.45), 65)):
        cam = CAM if kw.get("rhs_form") != 2 else np.array([2.0, -24.0, 14.0])
        kk = k if kw.get("rhs_form") != 2 else (k @ np.array([[1, 0, 0], [0, 0.5, 0.866], [0, -0.866, 0.5]]))
        big = ctx.trajectory(kk, cam, _params(**kw), T)           # 2100 rays: one lane per ray
        sub = ctx.trajectory(kk[:500], cam, _params(**kw), T)     # 500 rays: one wave per ray
        for a, b in zip(big, sub):
            assert np.array_equal(a[:500], b, equal_nan=True)
        nv, flags = sub[1], sub[3]
        assert (nv == T).any() and ((nv < T) & ((flags & 1) != 0)).any()
    # ... and up to 64 rays with 1024 samples or more, FOUR waves per ray share the samples (all four integrate the ray, every
    # lane alike): the same bits again, horizon rays and a sample count that is no multiple of 256 included
    k5 = frame_rays(70, seed=53, fov=0.16)
    for kw in (dict(r_s=1.0, lambda_end=50.0), dict(r_s=1.0, lambda_end=40.0, rhs_form=2, spin=0.45), dict(r_s=1.0, lambda_end=50.0, method=1, h_fixed=0.25)):
        one = ctx.trajectory(k5, CAM if kw.get("rhs_form") != 2 else np.array([2.0, -24.0, 14.0]), _params(**kw), 1500)          # 70 rays: one wave each
        four = ctx.trajectory(k5[:40], CAM if kw.get("rhs_form") != 2 else np.array([2.0, -24.0, 14.0]), _params(**kw), 1500)    # 40 rays: four waves each
        for a, b in zip(one, four):
            assert np.array_equal(a[:40], b, equal_nan=True)
        assert (four[1] == 1500).any() and ("rhs_form" in kw or (four[1] < 1500).any())    # (the Kerr camera looks past the hole)
    # the literal call's size, against a per-sample restatement of t_eval's rule: sample j is there iff its time
    # j * dt (the last one: curve_end) does not lie beyond where the ray ends
    k3 = np.array([[b / 30.0, 0.0, -1.0] for b in (5.0, 8.0, 12.0)])
    k3 /= np.linalg.norm(k3, axis=1)[:, None]
    traj, nv, end, flags = ctx.trajectory(k3, CAM, _params(r_s=1.0, lambda_end=50.0), 10000)
    assert np.all(nv == 10000) and np.isfinite(traj).all() and np.abs(traj[:, :, -1] - end).max() < 1e-9
    # consecutive samples are dt apart along a smooth curve: no sample was skipped or written twice by the lane split
    step = np.linalg.norm(np.diff(traj[:, 0:3, :], axis=2), axis=1)
    assert step.min() > 0.5 * 50.0 / 9999 * 0.9 and step.max() < 2.0 * 50.0 / 9999


def test_trajectories_batch_match_trace_and_oracle(ctx, oracle):
    k = frame_rays(300, seed=51)
    for kw in (dict(r_s=1.0, lambda_end=50.0), dict(r_s=1.0, lambda_end=50.0, rhs_form=1, r_exit=31.0),
               dict(r_s=1.0, lambda_end=40.0, rhs_form=2, spin=0.45)):
        cam = CAM if kw.get("rhs_form") != 2 else np.array([2.0, -24.0, 14.0])
        kk = k if kw.get("rhs_form") != 2 else (k @ np.array([[1, 0, 0], [0, 0.5, 0.866], [0, -0.866, 0.5]]))
        T = 64
        traj, nv, end, flags = ctx.trajectory(kk, cam, _params(**kw), T)
        e2, f2, s2, a2 = ctx.trace(kk, cam, _params(**kw))
        # same controller: rays that run to curve_end agree bit for bit; event rays were located by two searches to the
        # same 4-eps tolerance (brentq in the one-lane sampled path; the frame path's certified Newton where the event
        # function is monotone over the step) -- the roots differ by rounding, the end states by 1e-12 at most
        ev = (flags & (1 | 8)) != 0
        assert np.array_equal(flags, f2) and np.array_equal(end[~ev], e2[~ev])
        ex = (flags & 8) != 0
        assert np.abs(end[ex] - e2[ex]).max(initial=0.0) < (1e-11 if kw.get("rhs_form") != 2 else 1e-9)
        assert np.abs(end[ev & ~ex] - e2[ev & ~ex]).max(initial=0.0) < (1e-8 if kw.get("rhs_form") != 2 else 1e-5)   # horizon: k diverges there
        tr, onv, ofl = oracle.trajectory(kk, cam, T, **kw)
        assert np.array_equal(flags, ofl)
        same = nv == onv
        assert same.mean() > 0.99
        dmax = []
        for i in np.nonzero(same)[0][:120]:
            dmax.append(np.abs(traj[i, :, :nv[i]] - tr[i, :, :nv[i]]).max())
            assert np.isnan(traj[i, :, nv[i]:]).all()
        if kw.get("rhs_form") == 2:   # Boyer-Lindquist rays near the axis / horizon amplify rounding (see the Kerr tests)
            assert np.median(dmax) < 1e-8 and max(dmax) < 1e-2
        else:
            assert max(dmax) < 1e-8


def test_trajectories_with_fixed_step_rk4(ctx, oracle):
    """bhg_trajectory with BHG_METHOD_RK4 (ABI 7; VERDICT r04 missing #5): fixed steps h_fixed, samples on the step's cubic
    Hermite interpolant -- the interpolant the fixed-step kernels locate their events on.  Flags and end states: what bhg_trace
    gives for the same parameters (bit for bit for rays that run to curve_end); samples: the oracle's.  Wave-per-ray and
    lane-per-ray forms, exit sphere, disk, Kerr."""
    inc = np.radians(70.0)
    cam = np.array([30 * np.sin(inc), 0.0, 30 * np.cos(inc)])
    rot = np.array([[np.cos(inc), 0, np.sin(inc)], [0, 1, 0], [-np.sin(inc), 0, np.cos(inc)]])
    for n, kw in ((300, dict(r_s=1.0, lambda_end=45.0, method=1, h_fixed=0.25)),
                  (2300, dict(r_s=1.0, lambda_end=60.0, method=1, h_fixed=0.5, r_exit=38.0, disk_r_in=3.0, disk_r_out=9.0, rhs_form=1)),
                  (200, dict(r_s=1.0, lambda_end=40.0, method=1, h_fixed=0.2, rhs_form=2, spin=0.45, r_exit=38.0))):
        k = frame_rays(n, seed=59, fov=0.9) @ rot.T
        T = 80
        kerr = kw.get("rhs_form") == 2
        traj, nv, end, flags = ctx.trajectory(k, cam, _params(**kw), T)
        e2, f2, s2, a2 = ctx.trace(k, cam, _params(**kw))
        if kerr:   # fixed steps through the Boyer-Lindquist horizon: whether the crossing is seen before the state turns NaN is rounding
            ok = (flags == f2) | (((flags | f2) & ~np.uint8(1 | 64)) == 0)
            assert ok.all() and (flags == f2).mean() > 0.95
        else:
            assert np.array_equal(flags, f2)
        ran = (flags == 4) & (f2 == 4)
        assert ran.sum() > 0.2 * n and np.array_equal(end[ran], e2[ran])
        same_f = flags == f2
        ev = same_f & ((flags & (8 | 128)) != 0)
        assert np.abs(end[ev] - e2[ev]).max(initial=0.0) < (1e-10 if not kerr else 1e-8)
        tr, onv, ofl = oracle.trajectory(k, cam, T, **kw)
        agree = (flags == ofl)
        assert agree.mean() > (0.999 if not kerr else 0.9)
        same = agree & (nv == onv)
        assert same.mean() > (0.99 if not kerr else 0.85)
        worst = 0.0
        compared = 0
        for i in np.nonzero(same & ((flags & (1 | 64)) == 0))[0][:150]:
            m = nv[i]
            assert np.isnan(traj[i, :, m:]).all()
            if kerr and not np.abs(tr[i, :, :m]).max() < 200.0:
                continue      # (a fixed step that jumped the 1 / Delta singularity left garbage on both sides: nothing to compare)
            compared += 1
            worst = max(worst, np.abs(traj[i, :, :m] - tr[i, :, :m]).max())
        assert compared > 50 and worst < (1e-9 if not kerr else 1e-5), (compared, worst)
        # the curve's last sample of a ray that ran to curve_end IS its end state
        full = np.nonzero(ran & (nv == T))[0][:50]
        assert len(full) > 10
        for i in full:
            assert np.abs(traj[i, :, T - 1] - end[i]).max() < 1e-12


def test_trajectories_with_the_thin_disk_event(ctx, oracle):
    """bhg_trajectory with the disk (ABI 7; VERDICT r04 missing #5): the Limited engine finds its disk hit on the SAMPLED
    path (checkHitDisk walks x_SW, y_SW, z_SW, LimitedRelativisticRenderEngine.py:284, :413-438).  A ray that ends on the
    disk: flag HIT_DISK, the curve sampled up to the crossing and NaN beyond, end = the crossing point in the plane and
    inside the annulus -- the flags bhg_trace gives for the same parameters, the oracle's samples; a crossing outside
    the annulus is no event (the ray carries on through the plane).  Wave-per-ray form (<= 2048 rays) and lane-per-ray
    form; Schwarzschild from an inclined camera, and Kerr."""
    from blackhole_geodesic_calculator_amd import GeodesicIntegratorSchwarzschild
    inc = np.radians(70.0)
    cam = np.array([30 * np.sin(inc), 0.0, 30 * np.cos(inc)])
    rot = np.array([[np.cos(inc), 0, np.sin(inc)], [0, 1, 0], [-np.sin(inc), 0, np.cos(inc)]])
    for n, kw in ((400, dict(r_s=1.0, lambda_end=70.0, r_exit=40.0, disk_r_in=3.0, disk_r_out=9.0)),
                  (2600, dict(r_s=1.0, lambda_end=70.0, disk_r_in=4.5, disk_r_out=10.5, rhs_form=1)),
                  (300, dict(r_s=1.0, lambda_end=70.0, r_exit=40.0, disk_r_in=3.0, disk_r_out=9.0, rhs_form=2, spin=0.45))):
        k = frame_rays(n, seed=57, fov=0.9) @ rot.T
        T = 96
        traj, nv, end, flags = ctx.trajectory(k, cam, _params(**kw), T)
        e2, f2, s2, a2 = ctx.trace(k, cam, _params(**kw))
        kerr = kw.get("rhs_form") == 2
        assert np.array_equal(flags, f2)
        disk = flags == 128
        assert disk.sum() > 0.1 * n and ((flags & 1) != 0).sum() > 0 and (flags == (8 if "r_exit" in kw else 4)).sum() > 0.2 * n
        R = np.hypot(end[disk, 0], end[disk, 1])
        assert np.abs(end[disk, 2]).max() < 1e-9 and R.min() >= kw["disk_r_in"] - 1e-9 and R.max() <= kw["disk_r_out"] + 1e-9
        # the two searches (Brent here, the frame path's certified Newton) land on the same crossing
        assert np.abs(end[disk] - e2[disk]).max() < (1e-10 if not kerr else 1e-8)
        tr, onv, ofl = oracle.trajectory(k, cam, T, **kw)
        assert np.array_equal(flags, ofl)
        same = nv == onv
        assert same.mean() > 0.99
        t_eval = np.linspace(0.0, kw["lambda_end"], T)
        for i in np.nonzero(same & disk)[0][:150]:
            m = nv[i]
            assert 0 < m < T and np.isnan(traj[i, :, m:]).all()
            assert np.abs(traj[i, :, :m] - tr[i, :, :m]).max() < (1e-8 if not kerr else 1e-4)
            # the curve ends ON the disk: from its last sample the straight line to the end point does not cross the plane again
            assert traj[i, 2, m - 1] * (traj[i, 2, m - 1] - end[i, 2]) >= 0.0
        # a ray that crosses the plane outside the annulus carries on: its samples change the sign of z
        through = np.nonzero(~disk & (flags != 1) & (nv > 3))[0]
        crossed = [i for i in through[:400] if np.nanmin(traj[i, 2, :nv[i]]) < 0 < np.nanmax(traj[i, 2, :nv[i]])]
        assert len(crossed) > 10
        for i in crossed[:50]:
            z, x, y = traj[i, 2, :nv[i]], traj[i, 0, :nv[i]], traj[i, 1, :nv[i]]
            j = int(np.nonzero(np.sign(z[1:]) != np.sign(z[:-1]))[0][0])
            s_ = z[j] / (z[j] - z[j + 1])
            Rc = np.hypot(x[j] + s_ * (x[j + 1] - x[j]), y[j] + s_ * (y[j + 1] - y[j]))      # checkHitDisk's own interpolation (:419-421)
            assert Rc < kw["disk_r_in"] + 0.5 or Rc > kw["disk_r_out"] - 0.5
    # the adaptor: calc_trajectory(..., disk=) ends the curve on the disk
    gi = GeodesicIntegratorSchwarzschild(mass=0.5, context=ctx)
    i = int(np.nonzero(disk)[0][0]) if not kerr else 0
    k_s = frame_rays(400, seed=57, fov=0.9) @ rot.T
    tr0 = ctx.trace(k_s, cam, _params(r_s=1.0, lambda_end=70.0, disk_r_in=3.0, disk_r_out=9.0))
    i = int(np.nonzero(tr0[1] == 128)[0][0])
    k_xyz, x_xyz, res = gi.calc_trajectory(k_s[i], cam, curve_end=70.0, nr_points_curve=200, disk=(3.0, 9.0))
    assert res["hit_disk"] and not res["hit_blackhole"] and abs(res["end_loc"][2]) < 1e-9 and x_xyz.shape[1] < 200
    assert np.array_equal(res["end_loc"], tr0[0][i, 0:3]) or np.abs(res["end_loc"] - tr0[0][i, 0:3]).max() < 1e-10


def test_steps_run_ahead_edge_cases(ctx, oracle):
    """Round 6: in the exit-sphere kernels without the disk a ray whose NEXT step is expected to leave the sphere is handed to the
    short drain before that step is computed (EV_AHEAD, csrc/geodesic_kernels.hip): the drain runs the whole step -- clamp, error
    norm, accept / reject, event tests.  Where a step is computed must never show: every ray against the checker -- flags and both
    step counts identical -- where the drain's copy of the loop's logic is exercised hardest: a step budget that runs out at the
    handed-over step, tolerances at which it is rejected, a step cap, lambda_end inside it, the camera close to the sphere (the
    FIRST step leaves: never predicted), object spheres in its way, a sphere so small that nearly every step is a last step."""
    k = frame_rays(6000, seed=71, fov=1.2)
    cam = np.array([0.5, -1.0, 26.0])
    for kw in (dict(r_s=1.0, lambda_end=90.0, r_exit=35.0),
               dict(r_s=1.0, lambda_end=90.0, r_exit=35.0, rhs_form=1),
               dict(r_s=1.0, lambda_end=90.0, r_exit=35.0, max_steps=6),
               dict(r_s=1.0, lambda_end=90.0, r_exit=35.0, max_steps=9),
               dict(r_s=1.0, lambda_end=90.0, r_exit=35.0, max_step=3.0),
               dict(r_s=1.0, lambda_end=90.0, r_exit=35.0, rtol=1e-7, atol=1e-10, rhs_form=1),
               dict(r_s=1.0, lambda_end=90.0, r_exit=35.0, rtol=3e-2, atol=1e-4),
               dict(r_s=1.0, lambda_end=57.0, r_exit=35.0),              # lambda_end is reached near the sphere: either may end the ray
               dict(r_s=1.0, lambda_end=90.0, r_exit=26.5),              # the camera sits just inside: the first step leaves
               dict(r_s=1.0, lambda_end=90.0, r_exit=27.5, max_step=0.4),
               dict(r_s=1.0, lambda_end=90.0, r_exit=35.0, spheres=[[0.0, 0.0, -30.0, 6.0], [3.0, 2.0, 10.0, 1.5]]),
               dict(r_s=0.0, lambda_end=90.0, r_exit=35.0)):
        end, flags, steps, d = _compare(ctx, oracle, k[:3000] if kw.get("max_step") == 0.4 else k, cam, **kw)
        assert (flags == 8).sum() > 0.3 * len(flags) or kw.get("max_steps") or kw.get("lambda_end") == 57.0, kw
    # per-ray origins on the sphere's inside, rays in all directions
    rng = np.random.default_rng(72)
    x0 = rng.normal(size=(4000, 3))
    x0 *= (rng.uniform(4.0, 19.5, 4000) / np.linalg.norm(x0, axis=1))[:, None]
    kk = rng.normal(size=(4000, 3))
    kk /= np.linalg.norm(kk, axis=1)[:, None]
    _compare(ctx, oracle, kk, x0, r_s=1.0, lambda_end=60.0, r_exit=20.0)
    _compare(ctx, oracle, kk, x0, r_s=1.0, lambda_end=60.0, r_exit=20.0, time_like=1)   # (the time-like unit: one run-time event variant, no ahead steps)


def test_trajectories_with_object_spheres(ctx, oracle):
    """bhg_trajectory_objects (ABI 8; VERDICT r05 missing #4): the engine's literal per-ray call is exactly where the reference
    put its collision stub (RelativisticRenderEngine.py:293-305).  On the rays of the committed `objects` golden set and on
    seeded frames (wave-per-ray and lane-per-ray shapes; Schwarzschild both forms, Kerr, fixed-step RK4): the flags, sphere
    ids and entry points of the batch trace (bhg_trace_objects) and of the golden set; the oracle's samples; the curve
    NaN behind the entry point, every sample in front of it outside every sphere, and bit for bit the samples of the same
    call without spheres up to there; n_spheres = 0 is bhg_trajectory."""
    from blackhole_geodesic_calculator_amd import GeodesicIntegratorSchwarzschild
    g = load_golden("objects")
    sph = np.asarray(g["spheres"], dtype=np.float64).reshape(-1, 4)
    kw = dict(r_s=1.0, lambda_end=70.0, max_step=0.25, r_exit=35.0, disk_r_in=3.0, disk_r_out=7.0)
    T = 128
    traj, nv, end, flags, obj = ctx.trajectory(g["k0"], g["x0"], _params(**kw), T, spheres=sph)
    assert np.array_equal(flags, g["flags"]) and np.array_equal(obj, g["object_id"]) and np.abs(end - g["end"]).max() <= 1e-8
    e2, f2, s2, a2, o2 = ctx.trace(g["k0"], g["x0"], _params(**kw), spheres=sph)
    assert np.array_equal(flags, f2) and np.array_equal(obj, o2) and np.abs(end - e2).max() < 1e-10
    # ... and the Boyer-Lindquist golden set (scipy terminal events on the Cartesian image of the Kerr solve)
    gk = load_golden("kerr_objects")
    kwk = dict(r_s=1.0, lambda_end=60.0, max_step=0.5, rhs_form=2, spin=float(gk["spin"]))
    _, nvk, endk, flk, objk = ctx.trajectory(gk["k0"], gk["x0"], _params(**kwk), 64, spheres=gk["spheres"])
    assert np.array_equal(flk, gk["flags"]) and np.array_equal(objk, gk["object_id"]) and np.abs(endk - gk["end"]).max() < 1e-6
    assert np.all(nvk[flk == 0x88] < 64)
    cam = np.array([4.0, -24.0, 13.0])
    rng = np.random.default_rng(61)
    spheres = np.array([[5.0, 0.0, 0.0, 1.5], [0.0, -6.0, 2.0, 1.2], [0.2, 0.1, 7.0, 1.0], [-4.0, 3.0, -3.0, 1.3]])
    for n, kw in ((600, dict(r_s=1.0, lambda_end=60.0)),
                  (2500, dict(r_s=1.0, lambda_end=60.0, rhs_form=1, r_exit=30.0, disk_r_in=2.0, disk_r_out=9.0)),
                  (300, dict(r_s=1.0, lambda_end=60.0, rhs_form=2, spin=0.45)),
                  (300, dict(r_s=1.0, lambda_end=40.0, method=1, h_fixed=0.1))):
        k = (-cam / np.linalg.norm(cam))[None, :] + rng.normal(size=(n, 3)) * 0.2
        k /= np.linalg.norm(k, axis=1)[:, None]
        kerr, rk4 = kw.get("rhs_form") == 2, kw.get("method") == 1
        T = 96
        traj, nv, end, flags, obj = ctx.trajectory(k, cam, _params(**kw), T, spheres=spheres)
        e2, f2, s2, a2, o2 = ctx.trace(k, cam, _params(**kw), spheres=spheres)
        hit = flags == 0x88
        assert np.array_equal(flags, f2) and np.array_equal(obj, o2)
        assert hit.sum() > 0.03 * n and np.all(obj[hit] >= 0) and np.all(obj[~hit] == -1)
        assert np.abs(end[hit] - e2[hit]).max() < (1e-10 if not kerr else 1e-8)
        c = spheres[obj[hit]]
        assert np.abs(np.linalg.norm(end[hit, 0:3] - c[:, 0:3], axis=1) - c[:, 3]).max() < 1e-9      # the ray ends ON its sphere
        tr, onv, ofl = oracle.trajectory(k, cam, T, spheres=spheres, **kw)
        assert np.array_equal(flags, ofl) and (nv == onv).mean() > 0.99
        plain = ctx.trajectory(k, cam, _params(**kw), T)
        for i in np.nonzero(hit & (nv == onv))[0][:120]:
            m = nv[i]
            assert 0 < m < T and np.isnan(traj[i, :, m:]).all()
            assert np.abs(traj[i, :, :m] - tr[i, :, :m]).max() < (1e-8 if not (kerr or rk4) else 1e-4)
            # the same steps as the call without spheres up to the entry point: the same samples, bit for bit
            assert np.array_equal(traj[i, :, :m], plain[0][i, :, :m])
            # ... all of them outside every sphere (the curve stops in front of the one it enters)
            dist = np.linalg.norm(traj[i, 0:3, :m].T[:, None, :] - spheres[None, :, 0:3], axis=2) - spheres[None, :, 3]
            assert dist.min() > -1e-9
        # rays that meet no sphere: the plain call's results altogether
        free = ~hit
        assert np.array_equal(traj[free], plain[0][free], equal_nan=True) and np.array_equal(nv[free], plain[1][free])
        assert np.array_equal(end[free], plain[2][free], equal_nan=True) and np.array_equal(flags[free], plain[3][free])
    # n_spheres = 0 through the objects entry point is bhg_trajectory
    k = frame_rays(200, seed=3)
    t0 = ctx.trajectory(k, CAM, _params(r_s=1.0, lambda_end=50.0), 64)
    t1 = ctx.trajectory(k, CAM, _params(r_s=1.0, lambda_end=50.0), 64, spheres=np.zeros((0, 4)))
    assert all(np.array_equal(a, b, equal_nan=True) for a, b in zip(t0, t1[:4])) and np.all(t1[4] == -1)
    # the adaptor: calc_trajectory(spheres=) -- the engine's literal call with the collision test the reference left out
    gi = GeodesicIntegratorSchwarzschild(mass=0.5, context=ctx)
    kk = (-cam / np.linalg.norm(cam))[None, :] + np.random.default_rng(62).normal(size=(400, 3)) * 0.2
    kk /= np.linalg.norm(kk, axis=1)[:, None]
    tr0 = gi.trace(kk, cam, curve_end=60.0, spheres=spheres)
    i = int(np.nonzero(tr0["flags"] == 0x88)[0][0])
    k_xyz, x_xyz, res = gi.calc_trajectory(kk[i], cam, curve_end=60.0, nr_points_curve=10000, spheres=spheres)
    assert res["hit_object"] and res["object_id"] == int(tr0["object_id"][i]) and not res["hit_blackhole"]
    assert np.abs(res["end_loc"] - tr0["ray_end"][i, 0:3]).max() < 1e-10 and 0 < x_xyz.shape[1] < 10000
    # the sampled curve runs up to the sphere: its last sample is within one sample spacing of the entry point
    assert np.linalg.norm(x_xyz[:, -1] - res["end_loc"]) < 2.0 * 60.0 / 9999 * np.linalg.norm(k_xyz[:, -1]) + 1e-9
    j = int(np.nonzero(tr0["flags"] != 0x88)[0][0])
    _, _, res_j = gi.calc_trajectory(kk[j], cam, curve_end=60.0, nr_points_curve=100, spheres=spheres)
    assert not res_j["hit_object"] and res_j["object_id"] == -1
    # validation: the spheres are checked like bhg_trace_objects' (a radius <= 0 is refused)
    with pytest.raises(Exception):
        ctx.trajectory(k, CAM, _params(r_s=1.0, lambda_end=50.0), 64, spheres=[[1.0, 2.0, 3.0, -1.0]])


# ------------------------------------------------------------------------------------------------------------------------
# time_like=True (the solver object's other constructor value, RelativisticRenderEngine.py:134): massive particles
# ------------------------------------------------------------------------------------------------------------------------
def _orbits(n, seed):
    """Massive-particle start states around r_s = 1: radii 2.5 ... 14, tangential speeds 0 ... 1.7 x circular, a radial part."""
    rng = np.random.default_rng(seed)
    x0 = rng.normal(size=(n, 3))
    r0 = rng.uniform(2.5, 14.0, n)
    x0 *= (r0 / np.linalg.norm(x0, axis=1))[:, None]
    e_r = x0 / r0[:, None]
    e_t = np.cross(e_r, rng.normal(size=(n, 3)))
    e_t /= np.linalg.norm(e_t, axis=1)[:, None]
    v = np.sqrt(0.5 / np.maximum(r0 - 1.5, 0.8)) * rng.uniform(0.0, 1.7, n)
    return v[:, None] * e_t + rng.normal(0.0, 0.08, n)[:, None] * e_r, x0


def test_timelike_golden_and_oracle(ctx, oracle):
    g = load_golden("timelike")
    T = float(g["lambda_end"])
    end, flags, steps, acc = ctx.trace(g["k0"], g["x0"], _params(r_s=1.0, lambda_end=T, time_like=1))
    assert np.array_equal(flags, g["flags"]) and np.array_equal(steps, g["n_attempted"]) and np.array_equal(acc, g["n_accepted"])
    assert np.abs(end - g["end"]).max() < 1e-8
    end, flags, steps, acc = ctx.trace(g["k0"], g["x0"], _params(r_s=1.0, lambda_end=T, time_like=1, rhs_form=2, spin=float(g["spin"])))
    assert np.array_equal(flags, g["kerr_flags"]) and np.array_equal(steps, g["kerr_n_attempted"])
    d = np.abs(end - g["kerr_end"]).max(1)
    assert d[flags == 4].max() < 1e-8 and d.max() < 1e-5
    # at scale, per-ray origins: 6,000 orbits against the checker, every flag and every step count
    k0, x0 = _orbits(6000, 91)
    _, f1, _, d1 = _compare(ctx, oracle, k0, x0, r_s=1.0, lambda_end=120.0, time_like=1)
    assert 0.1 < ((f1 & 1) != 0).mean() < 0.7 and np.median(d1) < 1e-10
    _compare(ctx, oracle, k0[:2000], x0[:2000], r_s=1.0, lambda_end=120.0, time_like=1, r_exit=15.0, disk_r_in=3.0, disk_r_out=9.0)
    _compare(ctx, oracle, k0[:2000], x0[:2000], r_s=1.0, lambda_end=120.0, time_like=1, spheres=[[6.0, 0.0, 0.0, 1.5], [0.0, -7.0, 2.0, 1.0]])
    _compare(ctx, oracle, k0[:2000], x0[:2000], r_s=1.0, lambda_end=40.0, time_like=1, method=1, h_fixed=0.05, allow_flips=True)
    _compare(ctx, oracle, k0[:3000], x0[:3000], r_s=1.0, lambda_end=120.0, time_like=1, rhs_form=2, spin=0.45, step_flips=6)
    # the right-hand side itself
    for kw in (dict(), dict(rhs_form=2, spin=0.45)):
        q = x0[:500] if not kw else np.stack([np.linalg.norm(x0[:500], axis=1), np.arccos(x0[:500, 2] / np.linalg.norm(x0[:500], axis=1)),
                                              np.arctan2(x0[:500, 1], x0[:500, 0])], -1)
        a_g = ctx.acceleration(q, k0[:500], _params(r_s=1.0, time_like=1, **kw))
        a_o = oracle.acceleration(q, k0[:500], r_s=1.0, time_like=1, **kw)
        a_n = oracle.acceleration(q, k0[:500], r_s=1.0, **kw)
        assert np.abs(a_g - a_o).max() < 1e-12 * max(1.0, np.abs(a_o).max()) and np.abs(a_o - a_n).max() > 1e-3
    from blackhole_geodesic_calculator_amd import _ffi
    with pytest.raises(_ffi.BhgError):
        ctx.trace(k0[:4], x0[:4], _params(r_s=1.0, time_like=1, rhs_form=1))       # the reduced form is the null closed form


def test_timelike_adaptor_orbits(ctx, oracle):
    """GeodesicIntegratorSchwarzschild(time_like=True): a circular orbit comes back to its start after one proper period,
    sampled through calc_trajectory it stays on its circle, and g(k, k) = -1 along an eccentric one."""
    from blackhole_geodesic_calculator_amd import GeodesicIntegratorKerr, GeodesicIntegratorSchwarzschild
    gi = GeodesicIntegratorSchwarzschild(mass=0.5, time_like=True, verbose=False, context=ctx, rtol=1e-10, atol=1e-12)
    r0 = 4.0
    v = np.sqrt(0.5 / (r0 - 1.5))
    T = 2 * np.pi * r0 / v
    x0 = np.array([r0 * 0.6, 0.0, r0 * 0.8])
    k0 = np.array([0.0, v, 0.0])
    k_xyz, x_xyz, res = gi.calc_trajectory(k0, x0, curve_end=T, nr_points_curve=400)
    assert res["hit_blackhole"] is False and x_xyz.shape == (3, 400)
    assert np.abs(np.linalg.norm(x_xyz, axis=0) - r0).max() < 1e-7 and np.abs(res["end_loc"] - x0).max() < 1e-6
    tr, nv, fl = oracle.trajectory(k0[None], x0, 400, r_s=1.0, lambda_end=T, rtol=1e-10, atol=1e-12, time_like=1)
    assert nv[0] == 400 and np.abs(x_xyz - tr[0, 0:3]).max() < 1e-8 and np.abs(k_xyz - tr[0, 3:6]).max() < 1e-8
    # eccentric: the norm along the sampled curve, E = f k^t fixed at the start
    k1 = np.array([-0.1, 0.8 * v, 0.05])
    k_xyz, x_xyz, res = gi.calc_trajectory(k1, x0, curve_end=200.0, nr_points_curve=1000)
    r = np.linalg.norm(x_xyz, axis=0); f = 1 - 1 / r; h = 1 / (r - 1)
    nk = (x_xyz * k_xyz).sum(0) / r
    E2 = f * ((k_xyz * k_xyz).sum(0) + h * nk * nk + 1.0)
    assert x_xyz.shape[1] > 50 and np.abs(E2 - E2[0]).max() < 1e-7
    # Kerr, a/M = 0.9: prograde and retrograde equatorial orbits started alike end differently; with a = 0 the
    # Boyer-Lindquist solve agrees with the Cartesian one
    g0 = GeodesicIntegratorKerr(mass=0.5, a=0.0, time_like=True, context=ctx, rtol=1e-10, atol=1e-12)
    k2 = np.array([0.05, 1.1 * v, 0.03])       # bound, stays outside (k1 above ends on the horizon, which the two forms put 1e-3 apart)
    a0 = g0.trace(k2[None], x0, curve_end=60.0)
    s0 = gi.trace(k2[None], x0, curve_end=60.0)
    assert a0["flags"][0] == s0["flags"][0] == 4 and np.abs(a0["ray_end"] - s0["ray_end"]).max() < 1e-6
    g9 = GeodesicIntegratorKerr(mass=0.5, a=0.9, time_like=True, context=ctx, rtol=1e-10, atol=1e-12)
    xe = np.array([5.0, 0.0, 0.0])
    pro = g9.trace(np.array([[0.0, 0.3, 0.0]]), xe, curve_end=80.0)["ray_end"]
    ret = g9.trace(np.array([[0.0, -0.3, 0.0]]), xe, curve_end=80.0)["ray_end"]
    assert np.abs(pro[0, 0] - ret[0, 0]) > 1e-3 or np.abs(pro[0, 1] + ret[0, 1]) > 1e-3


@pytest.mark.filterwarnings("error:invalid value encountered in reduce")     # (an unmasked NaN reduction in the parity path: VERDICT r05)
def test_kerr_object_spheres_golden_and_oracle(ctx, oracle):
    """Object spheres with the Boyer-Lindquist form (round 4): the spheres live in the Cartesian frame, the chord rule runs on
    the images of a step's ends, the root search on the image of the dense output.  scipy golden (terminal events on the
    Cartesian image of the Kerr solve), then seeded rays against the checker incl. exit sphere + disk, RK4 and a sphere
    that straddles the rotation axis."""
    g = load_golden("kerr_objects")
    kw = dict(r_s=1.0, lambda_end=60.0, max_step=0.5, rhs_form=2, spin=float(g["spin"]))
    end, flags, steps, acc, obj = ctx.trace(g["k0"], g["x0"], _params(**kw), spheres=g["spheres"])
    assert np.array_equal(flags, g["flags"]) and np.array_equal(obj, g["object_id"]) and np.array_equal(acc, g["n_accepted"])
    assert (flags == 0x88).sum() >= 15 and np.abs(end - g["end"]).max() < 1e-6
    hit = flags == 0x88
    c = g["spheres"][obj[hit]]
    assert np.abs(np.linalg.norm(end[hit, 0:3] - c[:, 0:3], axis=1) - c[:, 3]).max() < 1e-9     # the ray ends ON its sphere
    cam = np.array([4.0, -24.0, 13.0])
    rng = np.random.default_rng(43)
    k = (-cam / np.linalg.norm(cam))[None, :] + rng.normal(size=(6000, 3)) * 0.2
    k /= np.linalg.norm(k, axis=1)[:, None]
    spheres = [[5.0, 0.0, 0.0, 1.5], [0.0, -6.0, 2.0, 1.2], [0.2, 0.1, 7.0, 1.0], [-4.0, 3.0, -3.0, 1.3]]
    _, f1, _, _ = _compare(ctx, oracle, k, cam, r_s=1.0, lambda_end=60.0, rhs_form=2, spin=0.45, spheres=spheres, step_flips=4)
    assert 0.02 < (f1 == 0x88).mean() < 0.6
    _compare(ctx, oracle, k[:3000], cam, r_s=1.0, lambda_end=60.0, rhs_form=2, spin=-0.3, spheres=spheres, r_exit=30.0,
             disk_r_in=2.0, disk_r_out=9.0, step_flips=4)
    _compare(ctx, oracle, k[:1500], cam, r_s=1.0, lambda_end=40.0, rhs_form=2, spin=0.45, spheres=spheres, method=1, h_fixed=0.1,
             allow_flips=0.15)
    # time-like as well: a massive particle stopped by a sphere
    k0, x0 = _orbits(1500, 92)
    _compare(ctx, oracle, k0, x0, r_s=1.0, lambda_end=100.0, rhs_form=2, spin=0.45, time_like=1, spheres=spheres, step_flips=4)


@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, (1 << 20) + 1])
def test_host_entry_points_on_empty_and_ragged_sizes(ctx, oracle, n):
    """Every host-buffer entry point on n = 0 (a no-op that touches nothing), one ray, the sizes either side of a 64-ray
    batch and one ray past a 2^20-ray pipeline chunk: bhg_trace, bhg_trace_objects, bhg_trajectory, bhg_acceleration,
    bhg_rays_create / bhg_rays_trace -- against the oracle."""
    from blackhole_geodesic_calculator_amd import _ffi
    k = frame_rays(max(n, 1), seed=77)[:n]
    kw = dict(r_s=1.0, lambda_end=50.0)
    end, flags, steps, acc = ctx.trace(k, CAM, _params(**kw))
    assert end.shape == (n, 6) and flags.shape == (n,) and steps.shape == (n,) and acc.shape == (n,)
    sph = [[2.5, 1.0, 10.0, 1.5]]
    e2, f2, s2, a2, obj = ctx.trace(k, CAM, _params(**kw), spheres=sph)
    assert e2.shape == (n, 6) and obj.shape == (n,)
    a = ctx.acceleration(np.zeros((n, 3)) + np.array([3.0, 1.0, 2.0]), k, _params(r_s=1.0))
    assert a.shape == (n, 3)
    if n <= 65:
        traj, nv, te, tf = ctx.trajectory(k, CAM, _params(**kw), 16)
        assert traj.shape == (n, 6, 16) and nv.shape == (n,)
        if n:
            assert np.array_equal(tf, flags) and np.array_equal(te[flags == 4], end[flags == 4])
    if n == 0:
        return
    o = oracle.trace(k, CAM, **kw)
    assert np.array_equal(flags, o["flags"]) and np.array_equal(steps, o["n_attempted"]) and np.array_equal(acc, o["n_accepted"])
    esc = flags == 4
    assert np.abs(end[esc] - o["end"][esc]).max(initial=0.0) <= STATED["escaped"][0]
    o2 = oracle.trace(k, CAM, spheres=sph, **kw)
    assert np.array_equal(f2, o2["flags"]) and np.array_equal(obj, o2["object_id"]) and np.array_equal(s2, o2["n_attempted"])
    assert np.allclose(a, oracle.acceleration(np.zeros((n, 3)) + np.array([3.0, 1.0, 2.0]), k, r_s=1.0), rtol=1e-12, atol=1e-15)


def test_resident_rays_of_an_empty_pixel_list(ctx):
    """bhg_rays_create with an EMPTY pixel list (a shard without tiles): a ray set of 0 rays; tracing it is a no-op."""
    from blackhole_geodesic_calculator_amd import _ffi
    rs = _ffi.RaySet(ctx, 64, 64, 2, 0.6, 0.6, CAM, None, None, False, np.zeros(0, np.int64))
    assert rs.n == 0
    out = rs.trace(_params(r_s=1.0, lambda_end=50.0), want=("end_dir", "flags"))
    assert out["end_dir"].shape == (0, 3) and out["flags"].shape == (0,)
    rs.close()
