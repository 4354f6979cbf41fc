"""CPU tests: the C oracle (oracle/geodesic_oracle.c) against the committed scipy golden vectors,
against scipy.solve_ivp run live, and against physics known answers.
The oracle is the checker for the GPU path; this file is what pins the checker."""
import math

import numpy as np
import pytest

from conftest import CAM, GOLDEN_TRACE_SETS, frame_rays, golden_kwargs, load_golden


@pytest.mark.parametrize("name", GOLDEN_TRACE_SETS)
def test_oracle_matches_scipy_golden(oracle, name):
    g = load_golden(name)
    form = 1 if name.endswith("reduced") else 0
    o = oracle.trace(g["k0"], g["x0"], **golden_kwargs(g, form))
    assert np.array_equal(o["flags"], g["flags"])
    assert np.array_equal(o["n_attempted"], g["n_attempted"])  # same accept/reject decisions as scipy
    assert np.array_equal(o["n_accepted"], g["n_accepted"])
    # tolerance: fp64 reassociation noise amplified along the curve; 5e-10 absolute on O(10) values
    assert np.abs(o["end"] - g["end"]).max() < 5e-10
    assert np.abs(o["t_end"] - g["t_end"]).max() < 1e-10


def test_oracle_matches_live_scipy(oracle):
    """scipy is importable wherever the tests run: compare step-for-step on fresh rays."""
    from oracle import scipy_reference as sr
    k = frame_rays(24, seed=11)
    ref = sr.trace_rays(k, CAM, r_s=1.0, lambda_end=50.0, form="christoffel")
    o = oracle.trace(k, CAM, r_s=1.0, lambda_end=50.0, rhs_form=0)
    assert np.array_equal(o["flags"], ref["flags"])
    assert np.array_equal(o["n_attempted"], ref["n_attempted"])
    assert np.abs(o["end"] - ref["end"]).max() < 5e-10


def test_rtol_below_100_eps_is_raised_like_scipy_does(oracle):
    """validate_tol (scipy _ivp/common.py:44-51): `rtol < 100 eps` is set to 100 eps with a warning -- every solve_ivp call
    does that, so the restatement does: rtol = 1e-15 and 1e-18 are the solve at rtol = 100 eps, bit for bit, and what scipy
    itself returns for rtol = 1e-15.  (At that tolerance the error estimate is rounding noise: step counts agree with scipy's
    to a fraction of a percent, not step for step.)"""
    import warnings
    from oracle import scipy_reference as sr
    cam = np.array([0.5, 0.0, 8.0])
    k = frame_rays(4, seed=78, fov=0.9)
    kw = dict(r_s=1.0, lambda_end=12.0, atol=1e-30)
    floor = oracle.trace(k, cam, rtol=100 * np.finfo(float).eps, **kw)
    for rtol in (1e-15, 1e-18):
        o = oracle.trace(k, cam, rtol=rtol, **kw)
        for key in ("end", "flags", "n_attempted", "n_accepted"):
            assert np.array_equal(o[key], floor[key], equal_nan=(key == "end"))
    assert not np.array_equal(oracle.trace(k, cam, rtol=1e-13, **kw)["n_attempted"], floor["n_attempted"])
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        ref = sr.trace_rays(k, cam, rtol=1e-15, form="christoffel", **kw)
    assert any("rtol" in str(x.message) for x in w)
    esc = (ref["flags"] == 4) & (floor["flags"] == 4)
    assert esc.sum() >= 2
    assert np.abs(floor["end"][esc] - ref["end"][esc]).max() < 1e-7
    assert np.all(np.abs(floor["n_attempted"][esc].astype(float) / ref["n_attempted"][esc] - 1.0) < 0.02)


def test_rhs_forms_agree(oracle):
    rng = np.random.default_rng(3)
    x = rng.normal(size=(500, 3)) * 6
    x = x[np.linalg.norm(x, axis=1) > 1.3]
    k = rng.normal(size=x.shape)
    a0 = oracle.acceleration(x, k, r_s=1.0, rhs_form=0)
    a1 = oracle.acceleration(x, k, r_s=1.0, rhs_form=1)
    scale = np.abs(a1).max(1)
    assert (np.abs(a0 - a1).max(1) / scale).max() < 1e-12


def test_rhs_matches_sympy_contraction(oracle):
    from oracle import scipy_reference as sr
    fn = sr.sympy_christoffel_rhs()
    rng = np.random.default_rng(4)
    x = rng.normal(size=(100, 3)) * 5
    x = x[np.linalg.norm(x, axis=1) > 1.3]
    k = rng.normal(size=x.shape)
    a = oracle.acceleration(x, k, r_s=1.0, rhs_form=0)
    for i in range(len(x)):
        ref = np.array(fn(x[i, 0], x[i, 1], x[i, 2], k[i, 0], k[i, 1], k[i, 2], 1.0))
        assert np.abs(a[i] - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max())


def test_flat_limit_is_straight_line(oracle):
    k = frame_rays(50, seed=2)
    o = oracle.trace(k, CAM, r_s=0.0, lambda_end=50.0, rhs_form=1)
    assert np.all(o["flags"] == oracle.FLAG_REACHED_END)
    assert np.abs(o["end"][:, 0:3] - (CAM + 50.0 * k)).max() < 1e-12
    assert np.abs(o["end"][:, 3:6] - k).max() < 1e-15


def test_fig5_deflection_known_answers(oracle):
    """README Fig. 5 geometry: x0 = -15 r_s, k = (1,0,0); asymptotic deflection angles."""
    g = load_golden("fig5")
    want = {3.0: 98.075, 4.0: 48.98, 5.0: 33.62, 10.0: 13.16, 19.0: 5.875}
    o = oracle.trace(g["k0"], g["x0"], r_s=1.0, lambda_end=4000.0, rtol=1e-11, atol=1e-13, rhs_form=1)
    defl = np.degrees(np.arctan2(-o["end"][:, 4], o["end"][:, 3]))
    assert np.abs(defl - g["deflection_deg_converged"]).max() < 1e-5
    for i, y0 in enumerate(g["x0"][:, 1]):
        if float(y0) in want:
            assert abs(defl[i] - want[float(y0)]) < 6e-3


def test_capture_threshold(oracle):
    """b_c = 3 sqrt(3)/2 r_s = 2.598: rays below are captured, above escape."""
    g = load_golden("capture")
    o = oracle.trace(g["k0"], g["x0"], r_s=1.0, lambda_end=200.0, rtol=1e-10, atol=1e-12, rhs_form=1)
    hit = (o["flags"] & oracle.FLAG_HIT_HORIZON) != 0
    b = g["x0"][:, 1]
    assert np.array_equal(hit, b < 1.5 * math.sqrt(3.0))
    assert np.array_equal(hit.astype(np.uint8), g["hit_converged"])


def test_conserved_quantities(oracle):
    k = frame_rays(200, seed=5)
    o = oracle.trace(k, CAM, r_s=1.0, lambda_end=50.0, rtol=1e-10, atol=1e-12, rhs_form=0)
    esc = o["flags"] == oracle.FLAG_REACHED_END
    L0 = np.cross(np.broadcast_to(CAM, k.shape), k)
    L1 = np.cross(o["end"][:, 0:3], o["end"][:, 3:6])
    assert np.abs(L1 - L0)[esc].max() < 1e-6
    # E = f k^t with (k^t)^2 from the null condition
    def energy(x, kk):
        r = np.linalg.norm(x, axis=1)
        n = x / r[:, None]
        nk = (n * kk).sum(1)
        f = 1 - 1.0 / r
        kt2 = ((kk * kk).sum(1) + (1.0 / (r - 1.0)) * nk * nk) / f
        return f * np.sqrt(kt2)
    e0 = energy(np.broadcast_to(CAM, k.shape), k)
    e1 = energy(o["end"][:, 0:3], o["end"][:, 3:6])
    assert np.abs(e1 / e0 - 1)[esc].max() < 1e-7


def test_rotation_equivariance(oracle):
    k = frame_rays(64, seed=6)
    th = 0.7
    R = np.array([[math.cos(th), -math.sin(th), 0], [math.sin(th), math.cos(th), 0], [0, 0, 1]]) @ \
        np.array([[1, 0, 0], [0, math.cos(0.4), -math.sin(0.4)], [0, math.sin(0.4), math.cos(0.4)]])
    kw = dict(r_s=1.0, lambda_end=50.0, rtol=1e-10, atol=1e-12, rhs_form=1)
    a = oracle.trace(k, CAM, **kw)
    b = oracle.trace(k @ R.T, R @ CAM, **kw)
    assert np.array_equal(a["flags"], b["flags"])
    esc = a["flags"] == oracle.FLAG_REACHED_END
    assert np.abs(a["end"][:, 0:3] @ R.T - b["end"][:, 0:3])[esc].max() < 1e-5
    assert np.abs(a["end"][:, 3:6] @ R.T - b["end"][:, 3:6])[esc].max() < 1e-6


def test_start_inside_and_caps(oracle):
    k = frame_rays(4, seed=7)
    o = oracle.trace(k, np.array([0.3, 0.2, 0.1]), r_s=1.0, lambda_end=50.0)
    assert np.all(o["flags"] == (oracle.FLAG_START_INSIDE | oracle.FLAG_HIT_HORIZON))
    assert np.all(o["n_attempted"] == 0)
    o = oracle.trace(k, CAM, r_s=1.0, lambda_end=50.0, max_steps=3)
    assert np.all(o["flags"] == oracle.FLAG_MAX_STEPS) and np.all(o["n_attempted"] == 3)


def test_rk4_converges_to_dp54(oracle):
    k = frame_rays(32, seed=8)
    a = oracle.trace(k, CAM, r_s=1.0, lambda_end=50.0, method=1, h_fixed=0.02, rhs_form=1)
    b = oracle.trace(k, CAM, r_s=1.0, lambda_end=50.0, rtol=1e-11, atol=1e-13, rhs_form=1)
    same = a["flags"] == b["flags"]
    assert same.mean() > 0.9
    esc = same & (a["flags"] == oracle.FLAG_REACHED_END)
    assert np.median(np.abs(a["end"] - b["end"])[esc].max(1)) < 1e-5


def test_oracle_disk_crossing_matches_scipy_golden(oracle):
    """Thin disk (LimitedRelativisticRenderEngine.py:413-438) as a scipy event g = z."""
    g = load_golden("disk")
    o = oracle.trace(g["k0"], g["x0"], r_s=1.0, lambda_end=80.0, disk_r_in=float(g["disk_r_in"]),
                     disk_r_out=float(g["disk_r_out"]))
    assert np.array_equal(o["flags"], g["flags"])
    assert np.array_equal(o["n_accepted"], g["n_accepted"])
    # a crossing time is only as well determined as the ray is steep: dt = dz / |k_z|
    steep = np.abs(g["end"][:, 5]) / np.linalg.norm(g["end"][:, 3:6], axis=1)
    assert np.all(np.abs(o["end"] - g["end"]).max(1) < 5e-9 + 1e-12 / np.maximum(steep, 1e-12))
    disk = o["flags"] == oracle.FLAG_HIT_DISK
    assert disk.sum() > 30
    R = np.hypot(o["end"][disk, 0], o["end"][disk, 1])
    assert np.abs(o["end"][disk, 2]).max() < 1e-12 and R.min() >= 4.5 and R.max() <= 10.5


# ---- Kerr (BASELINE.json config 5): sympy-derived Boyer-Lindquist Christoffels --------------------
def test_oracle_kerr_matches_scipy_golden(oracle):
    g = load_golden("kerr_a09")
    o = oracle.trace(g["k0"], g["x0"], r_s=1.0, lambda_end=60.0, rhs_form=oracle.RHS_KERR_BL, spin=float(g["spin"]))
    assert np.array_equal(o["flags"], g["flags"])
    assert np.array_equal(o["n_attempted"], g["n_attempted"])
    assert np.array_equal(o["n_accepted"], g["n_accepted"])
    d = np.abs(o["end"] - g["end"]).max(1)
    # the near-axis camera (first 48 rays, x = 1e-4 as in the reference's pickle names) sits on the
    # coordinate singularity theta = 0: phi amplifies rounding by ~1/sin(theta)
    assert d[48:].max() < 1e-5 and np.median(d[48:]) < 1e-9 and d[:48].max() < 5e-2


def test_kerr_conserves_killing_constants_and_null_norm(oracle):
    from oracle import scipy_reference as sr
    M, a = 0.5, 0.45
    cam = np.array([0.0, -25.0, 12.0])
    rng = np.random.default_rng(12)
    k = (-cam / np.linalg.norm(cam))[None, :] + rng.normal(size=(40, 3)) * 0.1
    k /= np.linalg.norm(k, axis=1)[:, None]
    o = oracle.trace(k, cam, r_s=1.0, lambda_end=60.0, rtol=1e-10, atol=1e-12, rhs_form=oracle.RHS_KERR_BL, spin=a)
    esc = o["flags"] == oracle.FLAG_REACHED_END
    assert esc.sum() > 10
    for i in np.nonzero(esc)[0]:
        q0, u0 = sr.cart_to_bl(cam, k[i], a)
        E0, L0, _ = sr.kerr_constants(q0, u0, M, a)
        q1, u1 = sr.cart_to_bl(o["end"][i, 0:3], o["end"][i, 3:6], a)
        E1, L1, _ = sr.kerr_constants(q1, u1, M, a)  # null condition re-solved at the end point
        assert abs(E1 - E0) < 1e-7 and abs(L1 - L0) < 1e-6
        # Carter constant Q = p_th^2 + cos^2 th (L^2 / sin^2 th - a^2 E^2)
        def carter(q, u, E, L):
            Sig = q[0] ** 2 + a * a * np.cos(q[1]) ** 2
            return (Sig * u[1]) ** 2 + np.cos(q[1]) ** 2 * (L * L / np.sin(q[1]) ** 2 - a * a * E * E)
        assert abs(carter(q1, u1, E1, L1) - carter(q0, u0, E0, L0)) < 1e-5


def test_kerr_zero_spin_is_schwarzschild(oracle):
    k = frame_rays(60, seed=13)
    cam = np.array([3.0, -20.0, 14.0])
    kw = dict(r_s=1.0, lambda_end=50.0, rtol=1e-11, atol=1e-13)
    a = oracle.trace(k, cam, rhs_form=oracle.RHS_KERR_BL, spin=1e-12, **kw)
    b = oracle.trace(k, cam, rhs_form=oracle.RHS_REDUCED, **kw)
    esc = (a["flags"] == 4) & (b["flags"] == 4)
    assert esc.sum() > 40 and np.array_equal(a["flags"] & 1, b["flags"] & 1)
    assert np.abs(a["end"] - b["end"])[esc].max() < 1e-6


def test_kerr_frame_dragging_breaks_the_mirror_symmetry(oracle):
    """Equatorial camera: prograde and retrograde rays with mirrored impact parameters differ."""
    cam = np.array([0.0, -30.0, 0.0])
    kp = np.array([+2.4 / 30, 1.0, 0.0]); km = np.array([-2.4 / 30, 1.0, 0.0])
    k = np.stack([kp / np.linalg.norm(kp), km / np.linalg.norm(km)])
    o = oracle.trace(k, cam, r_s=1.0, lambda_end=80.0, rtol=1e-9, atol=1e-11, rhs_form=oracle.RHS_KERR_BL, spin=0.45)
    s = oracle.trace(k, cam, r_s=1.0, lambda_end=80.0, rtol=1e-9, atol=1e-11, rhs_form=oracle.RHS_REDUCED)
    assert np.array_equal(s["flags"], [1, 1])      # b = 2.4 < 2.598: both captured without spin
    assert sorted(o["flags"].tolist()) == [1, 4]   # with a/M = 0.9 the prograde side escapes


def test_default_tolerance_error_against_converged_solution(oracle):
    """T2 of SURVEY section 7: how far the DEFAULT controller (rtol 1e-3, atol 1e-6, the engine's and
    scipy's defaults) is from a converged solve, per impact parameter -- the accuracy the reference's own
    settings deliver.  Measured constants are asserted with slack so that a regression shows."""
    bs = np.array([3.0, 4.0, 5.0, 7.0, 10.0, 14.0])
    k = np.stack([bs / 30.0, np.zeros_like(bs), -np.ones_like(bs)], 1)
    k /= np.linalg.norm(k, axis=1)[:, None]
    conv = oracle.trace(k, CAM, r_s=1.0, lambda_end=50.0, rtol=1e-12, atol=1e-14, rhs_form=1)
    for rtol, bound in ((1e-3, 0.6), (1e-5, 6e-3), (1e-7, 6e-5)):
        o = oracle.trace(k, CAM, r_s=1.0, lambda_end=50.0, rtol=rtol, atol=rtol * 1e-3, rhs_form=0)
        err = np.abs(o["end"] - conv["end"]).max(1)
        assert np.all(o["flags"] == 4) and err.max() < bound, (rtol, err)
    # and the order of the method shows: 100x tighter tolerance -> ~100x smaller error
    e3 = np.abs(oracle.trace(k, CAM, r_s=1.0, lambda_end=50.0, rtol=1e-5, atol=1e-8)["end"] - conv["end"]).max()
    e5 = np.abs(oracle.trace(k, CAM, r_s=1.0, lambda_end=50.0, rtol=1e-7, atol=1e-10)["end"] - conv["end"]).max()
    assert 10 < e3 / e5 < 1000


def test_oracle_objects_match_scipy_golden(oracle):
    """Object spheres as scipy terminal events (tests/golden/make_golden.py main_objects) beside horizon,
    exit sphere and disk; max_step is small, so scipy's sign-change detection sees every entry."""
    g = load_golden("objects")
    o = oracle.trace(g["k0"], g["x0"], r_s=1.0, lambda_end=70.0, max_step=0.25, r_exit=35.0, disk_r_in=3.0,
                     disk_r_out=7.0, spheres=g["spheres"])
    assert np.array_equal(o["flags"], g["flags"]) and np.array_equal(o["object_id"], g["object_id"])
    assert np.array_equal(o["n_accepted"], g["n_accepted"])
    nd = g["flags"] != oracle.FLAG_HIT_DISK
    assert np.array_equal(o["n_attempted"][nd], g["n_attempted"][nd])
    assert np.abs(o["end"] - g["end"]).max() < 1e-9 and np.abs(o["t_end"] - g["t_end"]).max() < 1e-10
    kinds = set(int(f) for f in g["flags"])
    assert {1, 8, 128, 0x88} <= kinds and len(set(g["object_id"][g["object_id"] >= 0])) >= 3


@pytest.mark.parametrize("method", [0, 1])
def test_oracle_objects_flat_space_exact(oracle, method):
    """r_s = 0: straight lines; DP5(4) steps are exact and long, so spheres are mostly passed through inside
    one step (the chord rule); entry points are the closed-form ray-sphere intersections."""
    rng = np.random.default_rng(0)
    n = 4000
    k = np.stack([rng.uniform(-0.3, 0.3, n), rng.uniform(-0.3, 0.3, n), -np.ones(n)], 1)
    k /= np.linalg.norm(k, axis=1)[:, None]
    sph = [(1.0, 0.5, 10.0, 1.5), (-2.0, 1.0, 0.0, 2.0), (0.5, -3.0, -12.0, 1.0)]
    o = oracle.trace(k, CAM, r_s=0.0, lambda_end=60.0, rhs_form=1, method=method, h_fixed=0.37, spheres=sph)
    best = np.full(n, np.inf)
    bid = np.full(n, -1)
    for j, (cx, cy, cz, rho) in enumerate(sph):
        oc = CAM - np.array([cx, cy, cz])
        b = k @ oc
        disc = b * b - (oc @ oc - rho * rho)
        t = np.where(disc > 0, -b - np.sqrt(np.maximum(disc, 0)), np.inf)
        upd = (t > 0) & (t < best)
        best = np.where(upd, t, best)
        bid = np.where(upd, j, bid)
    hit = best < 60.0
    assert hit.sum() > 300
    assert np.array_equal(o["flags"] == oracle.FLAG_HIT_OBJECT, hit)
    assert np.array_equal(o["object_id"], np.where(hit, bid, -1))
    assert np.abs(o["end"][hit, 0:3] - (CAM + best[hit, None] * k[hit])).max() < 1e-11
    if method == 0:
        assert o["n_attempted"].max() <= 8   # a handful of steps each: the spheres lie INSIDE steps


def test_oracle_objects_start_inside_sphere_and_ordering(oracle):
    """A ray that starts inside a sphere leaves it without an event; of two overlapping spheres the one
    entered first wins; a sphere behind the horizon is never reached."""
    k = np.array([[0.0, 0.0, -1.0]])
    x0 = np.array([0.0, 5.0, 20.0])
    o = oracle.trace(k, x0, r_s=1.0, lambda_end=60.0, spheres=[(0.0, 5.0, 20.0, 2.0), (0.0, 5.0, 8.0, 1.0), (0.0, 5.0, 9.0, 1.5)])
    assert o["flags"][0] == oracle.FLAG_HIT_OBJECT and o["object_id"][0] == 2
    assert abs(np.linalg.norm(o["end"][0, 0:3] - np.array([0.0, 5.0, 9.0])) - 1.5) < 1e-9
    # impact parameter 2 < 2.6 r_s: captured, whatever lies behind the hole
    o = oracle.trace(k, np.array([2.0, 0.0, 20.0]), r_s=1.0, lambda_end=60.0, spheres=[(0.0, 0.0, -6.0, 2.0), (2.0, 0.0, -6.0, 1.0)])
    assert o["flags"][0] == oracle.FLAG_HIT_HORIZON and o["object_id"][0] == -1


def test_scene_shade_reference_known_answers():
    """oracle/shade_reference.py's disk and object colours on hand-computed cases."""
    from oracle import shade_reference as sh
    # disk: R = 6 between 3 and 9 -> scale 0.5; white texture; mean 0.5 -> exp(0) / sqrt(2 pi sigma)
    e = np.array([[6.0, 0.0, 0.0, 0, 0, -1.0]])
    c = sh.disk_colour(e, 3.0, 9.0, None, phase=0.0, mean=0.5, stddev=0.3, intensity=2.0)
    assert np.allclose(c, 2.0 / np.sqrt(2 * np.pi * 0.3))
    # object: lamp straight above the hit point at distance 2, intensity 10 -> 100 * 1 / 4
    sph = [[0.0, 0.0, 0.0, 1.0], [0.0, 0.0, 2.0, 0.5]]
    e = np.array([[0.0, 0.0, 1.0, 0, 0, -1.0]])
    c = sh.object_colour(e, np.array([0]), sph[:1], [[1.0, 0.5, 0.25]], [[0.0, 0.0, 3.0, 10.0]])
    assert np.allclose(c, [[25.0, 12.5, 6.25]])
    # a second sphere between the point and the lamp: shadow; lamp behind the surface: nothing
    assert np.all(sh.object_colour(e, np.array([0]), sph, np.ones((2, 3)), [[0.0, 0.0, 3.0, 10.0]]) == 0.0)
    assert np.all(sh.object_colour(e, np.array([0]), sph[:1], np.ones((1, 3)), [[0.0, 0.0, -3.0, 10.0]]) == 0.0)


def test_oracle_kerr_disk_matches_scipy_golden(oracle):
    """Kerr with the thin disk in the equatorial plane: scipy event g = cos(theta) on the Boyer-Lindquist solve."""
    g = load_golden("kerr_disk")
    o = oracle.trace(g["k0"], g["x0"], r_s=1.0, lambda_end=80.0, rhs_form=oracle.RHS_KERR_BL, spin=float(g["spin"]),
                     disk_r_in=float(g["disk_r_in"]), disk_r_out=float(g["disk_r_out"]))
    assert np.array_equal(o["flags"], g["flags"]) and np.array_equal(o["n_accepted"], g["n_accepted"])
    disk = o["flags"] == oracle.FLAG_HIT_DISK
    assert disk.sum() > 30 and (o["flags"] == 1).sum() > 3
    tol = np.where((g["flags"] & 1) != 0, 1e-5, 1e-8)   # horizon: the Boyer-Lindquist end state is singular
    assert np.all(np.abs(o["end"] - g["end"]).max(1) < tol)
    R = np.hypot(o["end"][disk, 0], o["end"][disk, 1])
    assert np.abs(o["end"][disk, 2]).max() < 1e-12 and R.min() >= 3.0 and R.max() <= 10.0


def test_kerr_rhs_against_hamiltonian_form(oracle):
    """An independent judge for the generated Kerr right-hand side (oracle/kerr_rhs.inc comes from
    tools/gen_kerr_rhs.py's Christoffel derivation, and so do the scipy goldens): the same null geodesics from
    Hamilton's equations of H = 1/2 g^{mu nu} p_mu p_nu with the textbook INVERSE Boyer-Lindquist metric -- no
    Christoffel symbols, no sympy -- integrated with DOP853 at rtol 1e-12 (d g^{ab} / d(r, theta) by the
    complex-step derivative, exact to rounding).  p_t = -E and p_phi = L are constants; x-dot = g^{mu nu} p_nu uses the same affine parameter."""
    from scipy.integrate import solve_ivp
    from oracle import scipy_reference as sr
    M, a = 0.5, 0.45

    def ginv(r, th):
        s2, c2 = np.sin(th) ** 2, np.cos(th) ** 2
        Sig, Del = r * r + a * a * c2, r * r - 2 * M * r + a * a
        gtt = -((r * r + a * a) ** 2 - Del * a * a * s2) / (Sig * Del)
        gtp = -2 * M * a * r / (Sig * Del)
        return gtt, gtp, Del / Sig, 1.0 / Sig, (Del - a * a * s2) / (Sig * Del * s2)

    def ham(r, th, pr, pth, E, L):
        gtt, gtp, grr, gthth, gpp = ginv(r, th)
        return 0.5 * (gtt * E * E - 2 * gtp * E * L + grr * pr * pr + gthth * pth * pth + gpp * L * L)

    cam = np.array([2.0, -25.0, 12.0])
    rng = np.random.default_rng(5)
    k = (-cam / np.linalg.norm(cam))[None, :] + rng.normal(size=(16, 3)) * 0.12
    k /= np.linalg.norm(k, axis=1)[:, None]
    lam = 45.0
    o = oracle.trace(k, cam, r_s=2 * M, lambda_end=lam, rtol=1e-11, atol=1e-13, rhs_form=oracle.RHS_KERR_BL, spin=a)
    checked = 0
    for i in np.nonzero(o["flags"] == oracle.FLAG_REACHED_END)[0]:
        q0, u0 = sr.cart_to_bl(cam, k[i], a)
        E, L, _ = sr.kerr_constants(q0, u0, M, a)
        gtt, gtp, grr, gthth, gpp = sr.kerr_metric(q0[0], q0[1], M, a)
        y0 = [q0[0], q0[1], q0[2], grr * u0[0], gthth * u0[1]]

        def rhs(_t, y):
            r, th, _ph, pr, pth = y
            gtt_, gtp_, grr_, gthth_, gpp_ = ginv(r, th)
            h = 1e-30
            dHr = ham(r + 1j * h, th, pr, pth, E, L).imag / h
            dHth = ham(r, th + 1j * h, pr, pth, E, L).imag / h
            return [grr_ * pr, gthth_ * pth, -gtp_ * E + gpp_ * L, -dHr, -dHth]

        sol = solve_ivp(rhs, (0.0, lam), y0, method="DOP853", rtol=1e-12, atol=1e-14)
        assert sol.success and abs(ham(*sol.y[[0, 1, 3, 4], -1], E, L)) < 1e-9          # still null
        q1, _ = sr.cart_to_bl(o["end"][i, 0:3], o["end"][i, 3:6], a)
        dphi = (sol.y[2, -1] - q1[2] + np.pi) % (2 * np.pi) - np.pi
        assert abs(sol.y[0, -1] - q1[0]) < 2e-6 and abs(sol.y[1, -1] - q1[1]) < 2e-6 and abs(dphi) < 2e-6
        checked += 1
    assert checked >= 8


def test_kerr_snippet_is_the_generators_output_and_matches_the_contraction():
    """tools/gen_kerr_rhs.py --check: re-derives the Kerr Christoffel contraction with sympy, checks the structured
    omega / chi statements against it in exact rational arithmetic, and requires both committed copies of
    kerr_rhs.inc (oracle's and the kernel's) to be byte-for-byte what it emits."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "gen_kerr_rhs.py"), "--check"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]


def test_kerr_acceleration_probe_against_hamiltonian_form(oracle):
    """The oracle's Kerr right-hand side, point by point (not through trajectories): bhgo_acceleration with
    rhs_form = Kerr takes Boyer-Lindquist (r, theta, phi) and velocities, fixes E and L by the null condition at the
    point and evaluates the generated snippet; judged by Hamilton's equations with the inverse metric
    (tests/kerr_hamiltonian.py), for two masses (the generator's own check used to fix M = 1/2)."""
    import kerr_hamiltonian as kh
    for M, a in ((0.5, 0.45), (1.3, -0.9), (0.5, 0.0)):
        q, u = kh.sample_points(400, M, a, seed=int(M * 10))
        got = oracle.acceleration(q, u, r_s=2 * M, rhs_form=oracle.RHS_KERR_BL, spin=a)
        ref = np.array([kh.acceleration(q[i], u[i], M, a) for i in range(len(q))])
        scale = np.abs(ref).max(1) + 1e-300
        rel = np.abs(got - ref).max(1) / scale
        # close to the horizon Delta -> 0 and both forms lose digits to cancellation
        r_plus = M + np.sqrt(M * M - a * a)
        tol = 1e-9 + 1e-12 / ((q[:, 0] - r_plus) / r_plus) ** 2
        assert np.all(rel < tol), (M, a, float((rel / tol).max()))
        assert np.median(rel) < 1e-12


# ------------------------------------------------------------------------------------------------------------------------
# time_like=True: the solver object's other constructor value (RelativisticRenderEngine.py:134) -- massive particles
# ------------------------------------------------------------------------------------------------------------------------
def test_oracle_timelike_matches_scipy_golden(oracle):
    g = load_golden("timelike")
    o = oracle.trace(g["k0"], g["x0"], r_s=1.0, lambda_end=float(g["lambda_end"]), time_like=1)
    assert np.array_equal(o["flags"], g["flags"]) and set(o["flags"].tolist()) == {1, 4}
    assert np.array_equal(o["n_attempted"], g["n_attempted"]) and np.array_equal(o["n_accepted"], g["n_accepted"])
    assert np.abs(o["end"] - g["end"]).max() < 5e-10
    k = oracle.trace(g["k0"], g["x0"], r_s=1.0, lambda_end=float(g["lambda_end"]), time_like=1, rhs_form=oracle.RHS_KERR_BL,
                     spin=float(g["spin"]))
    assert np.array_equal(k["flags"], g["kerr_flags"])
    assert np.array_equal(k["n_attempted"], g["kerr_n_attempted"]) and np.array_equal(k["n_accepted"], g["kerr_n_accepted"])
    d = np.abs(k["end"] - g["kerr_end"]).max(1)
    assert d[k["flags"] == 4].max() < 1e-9 and d.max() < 1e-6      # (horizon ends: k diverges at r_plus)


def test_timelike_known_answers(oracle):
    """Circular orbits of the Schwarzschild metric (r_s = 1, M = 1/2): r dphi/dtau = sqrt(M / (r - 3M)), proper period
    2 pi r / that; stable above r = 6M = 3, so after one period the particle is back where it started -- in the Christoffel
    form, in the reduced form (with its Newtonian term) and in Boyer-Lindquist coordinates with a = 0.  A particle dropped
    from rest at r0 falls radially and crosses the horizon after tau = integral_{r_s}^{r0} dr / sqrt(r_s/r - r_s/r0)."""
    M = 0.5
    kw = dict(r_s=1.0, rtol=1e-11, atol=1e-13, time_like=1)
    for r0 in (3.5, 4.0, 8.0):
        v = math.sqrt(M / (r0 - 3 * M))
        T = 2 * math.pi * r0 / v
        x0 = np.array([r0, 0.0, 0.0]) @ np.array([[0.8, 0.0, 0.6], [0.0, 1.0, 0.0], [-0.6, 0.0, 0.8]])   # a tilted plane
        e_t = np.array([0.0, 1.0, 0.0])
        for rf in (0, 1, 2):
            o = oracle.trace((v * e_t)[None], x0, lambda_end=T, rhs_form=rf, spin=0.0, **kw)
            assert o["flags"][0] == 4
            assert np.abs(o["end"][0, 0:3] - x0).max() < 2e-7 * r0 and np.abs(o["end"][0, 3:6] - v * e_t).max() < 1e-7, (r0, rf)
    from scipy.integrate import quad
    r0 = 6.0
    tau, _ = quad(lambda r: 1.0 / math.sqrt(1.0 / r - 1.0 / r0), 1.0, r0, epsabs=1e-12, epsrel=1e-12, limit=200)
    for rf in (0, 1):
        o = oracle.trace(np.zeros((1, 3)), np.array([0.0, 0.0, r0]), lambda_end=2 * tau, rhs_form=rf, **kw)
        assert o["flags"][0] == 1 and abs(o["t_end"][0] - tau) < 1e-6 * tau, (rf, o["t_end"][0], tau)
    # the norm stays -1: -E^2 / f + |k|^2 + h (n.k)^2 with E = f k^t fixed at the start (f = 1 - r_s/r, h = r_s / (r - r_s))
    rng = np.random.default_rng(3)
    x0 = rng.normal(size=3); x0 *= 7.0 / np.linalg.norm(x0)
    k0 = rng.normal(size=(30, 3)) * 0.25
    def E2(x, k):
        r = np.linalg.norm(x, axis=-1); f = 1 - 1 / r; h = 1 / (r - 1)
        nk = (x * k).sum(-1) / r
        return f * ((k * k).sum(-1) + h * nk * nk + 1.0)          # = (f k^t)^2 from g(k, k) = -1
    o = oracle.trace(k0, x0, lambda_end=60.0, **kw)
    esc = o["flags"] == 4
    assert esc.sum() > 5
    assert np.abs(E2(o["end"][esc, 0:3], o["end"][esc, 3:6]) - E2(x0[None], k0[esc])).max() < 1e-8   # E conserved <=> norm kept
    Lv0, Lv1 = np.cross(x0[None], k0[esc]), np.cross(o["end"][esc, 0:3], o["end"][esc, 3:6])
    assert np.abs(Lv1 - Lv0).max() < 1e-8
    # and a null ray is NOT a time-like one: the flag changes the answer
    n = oracle.trace(k0, x0, r_s=1.0, lambda_end=60.0, rtol=1e-11, atol=1e-13)
    assert np.abs(n["end"] - o["end"])[esc & (n["flags"] == 4)].max() > 1e-2


def test_oracle_kerr_objects_match_scipy_golden(oracle):
    """Object spheres met by the Boyer-Lindquist solve in the Cartesian frame (round 4): the checker against scipy's terminal
    events on the Cartesian image of the Kerr state."""
    g = load_golden("kerr_objects")
    o = oracle.trace(g["k0"], g["x0"], r_s=1.0, lambda_end=60.0, max_step=0.5, rhs_form=oracle.RHS_KERR_BL, spin=float(g["spin"]),
                     spheres=g["spheres"])
    assert np.array_equal(o["flags"], g["flags"]) and np.array_equal(o["object_id"], g["object_id"])
    assert np.array_equal(o["n_accepted"], g["n_accepted"]) and (o["flags"] == 0x88).sum() >= 15
    assert np.abs(o["t_end"] - g["t_end"]).max() < 1e-9
    d = np.abs(o["end"] - g["end"]).max(1)
    assert d[o["flags"] != 1].max() < 1e-9 and d.max() < 1e-6
    hit = o["flags"] == 0x88
    c = g["spheres"][o["object_id"][hit]]
    assert np.abs(np.linalg.norm(o["end"][hit, 0:3] - c[:, 0:3], axis=1) - c[:, 3]).max() < 1e-9


def test_sampled_curves_are_solve_ivps_t_eval_output(oracle):
    """Row a2's literal semantics -- calc_trajectory(..., nr_points_curve=T) (RelativisticRenderEngine.py:293-294) -- pinned to
    scipy ITSELF: solve_ivp(..., t_eval=linspace(0, curve_end, T), events=[horizon (, exit sphere)]) against the C
    restatement's sampled curves.  Same number of samples per ray (a ray that ends on an event yields the grid points up to
    the root and no more, ivp.py:706-723), same values to rounding; rays to curve_end, horizon rays (the Fig. 5 / capture
    geometry) and exit-sphere rays; both right-hand-side forms."""
    from oracle import scipy_reference as sr
    g = load_golden("fig5")
    cases = [(g["k0"], g["x0"], dict(r_s=1.0, lambda_end=60.0), 121),
             (frame_rays(24, seed=93), CAM, dict(r_s=1.0, lambda_end=50.0), 50),
             (frame_rays(24, seed=94, fov=0.25), CAM, dict(r_s=1.0, lambda_end=70.0, r_exit=31.0), 77),
             (frame_rays(12, seed=95, fov=0.25), CAM, dict(r_s=1.0, lambda_end=50.0, max_step=0.5, rhs_form=1), 33)]
    seen = {1: 0, 4: 0, 8: 0}
    for k, x0, kw, T in cases:
        k = np.atleast_2d(k)
        tr, nv, fl = oracle.trajectory(k, x0, T, **kw)
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
            want = np.stack([sol.y[1], sol.y[3], sol.y[5], sol.y[0], sol.y[2], sol.y[4]])     # state order k_x, x, k_y, y, k_z, z (:301)
            d = np.abs(tr[i, :, :m] - want).max()
            assert d < (1e-9 if not (int(fl[i]) & 1) else 1e-6), (i, d)
            assert np.isnan(tr[i, :, m:]).all()
    assert seen[1] >= 5 and seen[4] >= 20 and seen[8] >= 10, seen
    # the thin disk: scipy's crossing event is non-terminal (the annulus test comes afterwards), so its samples run on past the
    # disk; the restatement's curve is scipy's up to the crossing and ends there
    inc = np.radians(70.0)
    cam = np.array([30 * np.sin(inc), 0.0, 30 * np.cos(inc)])
    rot = np.array([[np.cos(inc), 0, np.sin(inc)], [0, 1, 0], [-np.sin(inc), 0, np.cos(inc)]])
    k = frame_rays(40, seed=96, fov=0.8) @ rot.T
    T, kw = 90, dict(r_s=1.0, lambda_end=70.0, disk_r_in=3.0, disk_r_out=9.0)
    tr, nv, fl = oracle.trajectory(k, cam, T, **kw)
    t_eval = np.linspace(0.0, 70.0, T)
    hits = 0
    for i in range(len(k)):
        r = sr.trace_ray(k[i], cam, r_s=1.0, lambda_end=70.0, disk=(3.0, 9.0), nr_points_curve=T)
        assert int(fl[i]) == r["flags"]
        sol = r["sol"]
        m = int(np.searchsorted(t_eval, r["t_end"], side="right")) if r["flags"] == 128 else sol.y.shape[1]
        hits += r["flags"] == 128
        assert nv[i] == m, (i, int(nv[i]), m)
        want = np.stack([sol.y[1], sol.y[3], sol.y[5], sol.y[0], sol.y[2], sol.y[4]])[:, :m]
        assert np.abs(tr[i, :, :m] - want).max() < (1e-9 if not (int(fl[i]) & 1) else 1e-6)
    assert hits >= 8


def test_fixed_step_trajectories_are_the_hermite_interpolant_of_the_rk4_trace(oracle):
    """oracle.trajectory with method = RK4: samples on each fixed step's cubic Hermite interpolant.  The last sample of a ray
    that runs to lambda_end is the end state the trace gives; samples that fall ON step ends (t_eval a multiple of h) are the
    step ends of the trace at that lambda (traced again with lambda_end = that time); and with a small step the curve agrees
    with the adaptive solver's at tight tolerance."""
    k = frame_rays(60, seed=91)
    kw = dict(r_s=1.0, lambda_end=40.0, method=1, h_fixed=0.5)
    T = 81                                     # t_eval = 0, 0.5, 1.0, ...: every sample sits on a step end
    tr, nv, fl = oracle.trajectory(k, CAM, T, **kw)
    o = oracle.trace(k, CAM, **kw)
    assert np.array_equal(fl, o["flags"])
    ran = fl == 4
    assert ran.sum() > 30 and np.all(nv[ran] == T)
    assert np.abs(tr[ran][:, :, T - 1] - o["end"][ran]).max() < 1e-13
    i = int(np.nonzero(ran)[0][0])
    for j in (1, 7, 40):
        mid = oracle.trace(k[i:i + 1], CAM, **dict(kw, lambda_end=0.5 * j))
        assert np.abs(tr[i, :, j] - mid["end"][0]).max() < 1e-12
    fine = dict(r_s=1.0, lambda_end=40.0, method=1, h_fixed=0.02)
    t2, nv2, _ = oracle.trajectory(k[ran][:10], CAM, 33, **fine)
    t3, nv3, _ = oracle.trajectory(k[ran][:10], CAM, 33, r_s=1.0, lambda_end=40.0, rtol=1e-11, atol=1e-13)
    assert np.array_equal(nv2, nv3) and np.abs(t2 - t3).max() < 1e-5


def test_oracle_suite_under_sanitizers():
    """The checker is what every parity claim rests on: its C restatement built with -fsanitize=address,undefined
    (`make -C oracle asan`) must pass THIS file's tests in a child process with the sanitizer runtimes preloaded (a finding
    aborts the child).  OpenMP paths, the Brent searches, dense-output sampling and the Kerr right-hand side all run."""
    import os
    import subprocess
    import sys
    if os.environ.get("BHG_ORACLE_LIB"):
        pytest.skip("already the child run")
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    subprocess.check_call(["make", "-C", os.path.join(root, "oracle"), "-s", "asan"])
    lib = os.path.join(root, "oracle", "libgeodesic_oracle_asan.so")
    rts = []
    for name in ("libasan.so", "libubsan.so"):
        p = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
        if not os.path.isabs(p) or not os.path.exists(p):
            pytest.skip(f"{name} not found next to gcc")
        rts.append(p)
    env = dict(os.environ, BHG_ORACLE_LIB=lib, LD_PRELOAD=" ".join(rts), ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    # the child really runs the sanitizer build ...
    probe = ("import sys; sys.path.insert(0, %r); from oracle import oracle as oc; import numpy as np; "
             "oc.trace(np.array([[0.05, 0.02, -1.0]]), np.array([1e-4, 0.0, 30.0])); m = open('/proc/self/maps').read(); "
             "assert 'libgeodesic_oracle_asan.so' in m and 'libasan' in m and 'libgeodesic_oracle.so' not in m") % root
    subprocess.check_call([sys.executable, "-c", probe], env=env)
    # ... and this file's tests pass against it
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-p", "no:cacheprovider"], env=env,
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert " passed" in r.stdout and "failed" not in r.stdout
