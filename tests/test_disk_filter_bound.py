"""The thin-disk pre-filter of the trace kernels is a BOUND: property test on the CPU.

A step whose chord crosses z = 0 is parked for the event search only when the crossing may lie in the annulus; a step
the filter rejects carries on for good, so a too-small excursion figure loses a real disk hit
(raytracer/LimitedRelativisticRenderEngine.py:413-438 is the semantics at stake).  Here: scipy's RK45 steps (the
vectorised restatement in disk_filter_model.py, first checked against scipy.integrate.RK45 itself) from cameras at
inclinations 60 ... 89.999 degrees with aims down to sigma_z = 0.03 -- grazing crossings -- at rtol 1e-3 ... 1e-1;
for EVERY plane crossing of the dense output, |R_dense - R_chord| <= eps.
"""
import numpy as np
import pytest
from scipy.integrate import RK45

import disk_filter_model as dfm

R_S, T_END, CAM_R = 1.0, 70.0, 30.0


def _rays(n, seed, inc_lo=60.0, inc_hi=89.999, sigma=(9.0, 9.0, 0.03)):
    rng = np.random.default_rng(seed)
    # half of the cameras within a degree of the plane, where the crossings graze
    inc = np.where(rng.random(n) < 0.5, rng.uniform(89.0, inc_hi, n), rng.uniform(inc_lo, inc_hi, n))
    inc = np.deg2rad(inc)
    cam = CAM_R * np.stack([np.sin(inc), np.zeros(n), np.cos(inc)], -1)
    aim = rng.normal(0.0, 1.0, (n, 3)) * np.asarray(sigma)
    k = aim - cam
    k /= np.linalg.norm(k, axis=1, keepdims=True)
    return np.concatenate([cam, k], -1)


def test_vectorised_stepper_is_scipys_rk45():
    y0 = _rays(6, 1)
    for form in ("reduced", "christoffel"):
        steps = list(dfm.rk45_steps(y0, T_END, 1e-3, 1e-6, R_S, form))
        for i in range(len(y0)):
            def rhs(_t, y):
                return np.concatenate([y[3:6], dfm.accel(y[0:3], y[3:6], R_S, form)])
            sol = RK45(rhs, 0.0, y0[i], T_END, rtol=1e-3, atol=1e-6)
            mine = [(yo[i], yn[i], h[i], Q[i]) for ok, yo, yn, h, Q in steps if ok[i]]
            j = 0
            while sol.status == "running" and j < len(mine):
                sol.step()
                yo, yn, h, Q = mine[j]
                assert abs(h - (sol.t - sol.t_old)) <= 1e-9 * abs(h)   # (summation order differs: rounding drift only)
                np.testing.assert_allclose(yn, sol.y, rtol=1e-9, atol=1e-10)
                d = sol.dense_output()
                np.testing.assert_allclose(Q, d.Q, rtol=1e-6, atol=1e-9)
                j += 1
                if np.linalg.norm(sol.y[0:3]) <= 1.02 * R_S:
                    break
            assert j == len(mine)


def _crossings(n, seed, rtol, form, **ray_kw):
    """Every dense-output plane crossing of n rays: (R_dense, R_chord, eps, old_delta, step data ...)."""
    rows = []
    for ok, yo, yn, h, Q in dfm.rk45_steps(_rays(n, seed, **ray_kw), T_END, rtol, rtol * 1e-3, R_S, form):
        z0, z1 = yo[:, 2], yn[:, 2]
        cr = ok & (((z0 <= 0) & (z1 >= 0)) | ((z0 >= 0) & (z1 <= 0))) & (z0 != z1)
        if not cr.any():
            continue
        yo, yn, h, Q = yo[cr], yn[cr], h[cr], Q[cr]
        hq3 = h[:, None] * Q[:, 0:3, 3]
        Rc, eps = dfm.crossing_bound(yo[:, 0:3], yo[:, 3:6], yn[:, 0:3], yn[:, 3:6], h, hq3)
        old = dfm.old_delta(yo[:, 0:3], yo[:, 3:6], yn[:, 0:3], yn[:, 3:6], h)
        for i, roots in enumerate(dfm.dense_plane_crossings(yo, h, Q)):
            for th in roots:
                p = dfm.dense_eval(yo[i:i + 1], h[i:i + 1], Q[i:i + 1], np.array([th]))[0]
                rows.append((np.hypot(p[0], p[1]), Rc[i], eps[i], old[i], yo[i], yn[i], h[i], hq3[i]))
    return rows


@pytest.mark.parametrize("rtol", [1e-3, 3e-3, 1e-2, 1e-1])
def test_every_dense_crossing_lies_within_the_bound_of_the_chords(rtol):
    rows = _crossings(6000, int(rtol * 1e4), rtol, "reduced") + _crossings(1500, 7 + int(rtol * 1e4), rtol, "christoffel")
    assert len(rows) >= 5000, len(rows)
    Rd, Rc, eps, old = (np.array([r[j] for r in rows]) for j in range(4))
    ratio = np.abs(Rd - Rc) / eps
    assert ratio.max() <= 1.0, (rtol, ratio.max(), int(ratio.argmax()))
    # what the bound buys: how much of it the worst crossing uses, and what round 3's figure did with the same crossing
    worst_old = (np.abs(Rd - Rc) / np.maximum(old, 1e-300)).max()
    print(f"rtol {rtol:g}: {len(rows)} crossings, max |R_dense - R_chord| / eps = {ratio.max():.3f} "
          f"(median {np.median(ratio):.3f}); against round 3's delta: {worst_old:.2f}")
    # the device decision (division-free form): a crossing inside an annulus is never filtered out
    yo = np.array([r[4] for r in rows]); yn = np.array([r[5] for r in rows])
    h = np.array([r[6] for r in rows]); hq3 = np.array([r[7] for r in rows])
    rng = np.random.default_rng(5)
    for _ in range(8):
        # annuli with an edge right next to the dense crossing: the cases a loose filter gets wrong
        edge = Rd + rng.normal(0.0, 1.0, len(Rd)) * np.maximum(eps, 1e-9)
        inner = rng.random(len(Rd)) < 0.5
        r_in = np.where(inner, np.maximum(edge, 0.0), np.maximum(Rd - rng.uniform(0.5, 5.0, len(Rd)), 0.0))
        r_out = np.where(inner, Rd + rng.uniform(0.5, 5.0, len(Rd)), edge)
        inside = (Rd >= r_in) & (Rd <= r_out)
        keep = dfm.may_hit(yo[:, 0:3], yo[:, 3:6], yn[:, 0:3], yn[:, 3:6], h, hq3, r_in, r_out)
        assert not (inside & ~keep).any()


def test_round3_delta_was_not_a_bound():
    """Documents the hole this test closes: the chord-velocity figure of round 3 is exceeded on grazing crossings."""
    rows = _crossings(6000, 100, 1e-2, "reduced", inc_lo=89.7)     # the regime the review found it in
    Rd, Rc, eps, old = (np.array([r[j] for r in rows]) for j in range(4))
    assert (np.abs(Rd - Rc) / old).max() > 1.0
    assert (np.abs(Rd - Rc) / eps).max() <= 1.0


# ------------------------------------------------------------------------------------------------------------------------
# The Boyer-Lindquist forms of the filter (Kerr + thin disk): the chord bound in (r, theta) and the sharp test that follows it
# ------------------------------------------------------------------------------------------------------------------------
def _kerr_crossings(n, seed, rtol, M=0.5, a=0.45):
    """Every crossing of the equatorial plane by the dense output of scipy-RK45 steps on the Boyer-Lindquist Kerr system
    (the lambdified right-hand side the goldens were made with), from near-equatorial cameras: one row per crossing."""
    from oracle import scipy_reference as sr
    fn = sr.kerr_rhs_lambdified()
    rng = np.random.default_rng(seed)
    rows = []
    for _ in range(n):
        inc = np.deg2rad(rng.uniform(89.0, 89.99) if rng.random() < 0.5 else rng.uniform(70.0, 89.99))
        cam = CAM_R * np.array([np.sin(inc), 0.0, np.cos(inc)]) + np.array([0.0, 3.0, 0.0])
        aim = rng.normal(0.0, 1.0, 3) * np.array([9.0, 9.0, 0.1])
        k = (aim - cam) / np.linalg.norm(aim - cam)
        q0, u0 = sr.cart_to_bl(cam, k, a)
        E, L, _ = sr.kerr_constants(q0, u0, M, a)

        def rhs(_t, y):
            ar, ath, aph, _kt = fn(y[1], y[3], y[0], y[2], y[4], E, L, M, a)
            return np.array([ar, y[0], ath, y[2], aph, y[4]])

        sol = RK45(rhs, 0.0, np.array([u0[0], q0[0], u0[1], q0[1], u0[2], q0[2]]), 90.0, rtol=rtol, atol=rtol * 1e-3)
        r_h = (M + np.sqrt(M * M - a * a)) * 1.02
        while sol.status == "running":
            y_old = sol.y.copy()
            sol.step()
            if sol.status == "failed" or sol.y[1] <= r_h:
                break
            if dfm.bl_plane_index(np.array(y_old[3])) == dfm.bl_plane_index(np.array(sol.y[3])):
                continue
            dn = sol.dense_output()
            h, Q = dn.h, dn.Q
            qa, ua = y_old[[1, 3, 5]], y_old[[0, 2, 4]]
            qb, ub = sol.y[[1, 3, 5]], sol.y[[0, 2, 4]]
            hq3 = h * Q[[1, 3, 5], 3]
            for kk in range(int(min(dfm.bl_plane_index(qa[1]), dfm.bl_plane_index(qb[1]))) + 1,
                            int(max(dfm.bl_plane_index(qa[1]), dfm.bl_plane_index(qb[1]))) + 1):
                th_star = np.pi * kk + dfm.HALF_PI
                c = np.array([h * Q[3, 3], h * Q[3, 2], h * Q[3, 1], h * Q[3, 0], y_old[3] - th_star])
                roots = np.roots(c)
                roots = roots[np.abs(roots.imag) < 1e-9].real
                for th in roots[(roots >= -1e-12) & (roots <= 1.0 + 1e-12)].clip(0.0, 1.0):
                    p = th ** np.arange(1, 5)
                    r_d = y_old[1] + h * (Q[1] @ p)
                    rows.append((r_d, qa, ua, qb, ub, h, hq3))
    return rows


@pytest.mark.parametrize("rtol", [1e-3, 1e-2])
def test_boyer_lindquist_filters_are_bounds(rtol):
    a = 0.45
    rows = _kerr_crossings(1500, 40 + int(rtol * 1e4), rtol, a=a)
    assert len(rows) >= 1300, len(rows)
    r_d = np.array([r[0] for r in rows])
    q0, u0, q1, u1 = (np.array([r[j] for r in rows]) for j in (1, 2, 3, 4))
    h, hq3 = np.array([r[5] for r in rows]), np.array([r[6] for r in rows])
    one, _, r_lin, eps = dfm.bl_chord_bound(q0, u0, q1, u1, h, hq3)
    ratio = np.abs(r_d - r_lin)[one] / eps[one]
    assert one.mean() > 0.9 and ratio.max() <= 1.0, (rtol, ratio.max())
    decides, r_h, er = dfm.sharp_bl(q0, u0, q1, u1, h, hq3)
    sratio = np.abs(r_d - r_h)[decides] / er[decides]
    assert decides.mean() > 0.5 and sratio.max() <= 1.0, (rtol, sratio.max())
    print(f"rtol {rtol:g}: {len(rows)} Kerr crossings; chord bound: worst uses {ratio.max():.3f} of eps (median {np.median(ratio):.3f}); "
          f"sharp test decides {decides.mean():.2f} of them, worst uses {sratio.max():.3f} of its margin (median {np.median(sratio):.4f})")
    # the device decisions: a crossing inside the annulus (in sqrt(r^2 + a^2)) is never filtered out, by either test
    R_d = np.sqrt(r_d * r_d + a * a)
    rng = np.random.default_rng(6)
    for _ in range(8):
        edge = R_d + rng.normal(0.0, 1.0, len(R_d)) * np.maximum(np.where(decides, er, eps), 1e-9)
        inner = rng.random(len(R_d)) < 0.5
        r_in = np.where(inner, np.maximum(edge, 0.0), np.maximum(R_d - rng.uniform(0.5, 5.0, len(R_d)), 0.0))
        r_out = np.where(inner, R_d + rng.uniform(0.5, 5.0, len(R_d)), edge)
        inside = (R_d >= r_in) & (R_d <= r_out)
        assert not (inside & ~dfm.may_hit_bl(q0, u0, q1, u1, h, hq3, a, r_in, r_out)).any()
        assert not (inside & ~dfm.may_hit_sharp_bl(q0, u0, q1, u1, h, hq3, a, r_in, r_out)).any()
