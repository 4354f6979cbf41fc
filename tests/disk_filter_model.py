"""numpy restatement of the thin-disk pre-filter of the trace kernels, and a vectorised scipy-RK45 stepper to feed it.

TEST INFRASTRUCTURE.  The device functions it restates:
    disk_crossing_may_hit      blackhole_geodesic_calculator_amd/csrc/geodesic_kernels.hip
    disk_crossing_may_hit_bl   (Boyer-Lindquist form)
Semantics of the event they pre-filter: first z-sign change between consecutive samples with the crossing inside
the annulus (raytracer/LimitedRelativisticRenderEngine.py:413-438); scipy locates it on the step's dense output
(ivp.py:51-76, :109-126).

The bound.  One accepted DP5(4) step runs from (x0, v0) at theta = 0 to (x1, v1) at theta = 1; its dense output
D(theta) is a quartic in every position component with D(0) = x0, D(1) = x1, D'(0) = h v0, D'(1) = h v1 (the
Dormand-Prince interpolant is C1), leading coefficient h q3 (q3 = column 3 of rk.py's Q = K^T P).  With H the cubic
Hermite interpolant of the same end data and C the chord,
    D - H = h q3 theta^2 (1 - theta)^2                        -> |D_c - H_c| <= |h q3_c| / 16
    H - C = e0 theta (1 - theta)^2 - e1 theta^2 (1 - theta)   -> |H_c - C_c| <= (4/27) (|e0_c| + |e1_c|)
with e0 = h v0 - (x1 - x0), e1 = h v1 - (x1 - x0).  So every component of the dense curve stays within
    delta_c = (4/27) (|e0_c| + |e1_c|) + |h q3_c| / 16
of the chord.  Any plane crossing theta_d of the dense output (D_z(theta_d) = 0) then lies within delta_z / |dz| of
the chord's crossing parameter, and the crossing POINT within
    eps = delta_x + delta_y + (|dx| + |dy|) delta_z / |dz|
(1-norm, >= its length, >= the difference of the two cylindrical radii) of the chord's crossing point.
"""
from __future__ import annotations

import numpy as np
from scipy.integrate._ivp.rk import RK45

A_, B_, C_, E_, P_ = RK45.A, RK45.B, RK45.C, RK45.E, RK45.P


def component_bound(x0, v0, x1, v1, h, hq3):
    """delta_c [..., 3]: how far the dense output strays from the chord, per component."""
    dx = x1 - x0
    e0 = h[..., None] * v0 - dx
    e1 = h[..., None] * v1 - dx
    return (4.0 / 27.0) * (np.abs(e0) + np.abs(e1)) + np.abs(hq3) / 16.0


def crossing_bound(x0, v0, x1, v1, h, hq3):
    """(R_chord, eps): cylindrical radius of the chord's z = 0 crossing and the bound on |R_dense - R_chord|."""
    d = component_bound(x0, v0, x1, v1, h, hq3)
    dx = x1 - x0
    s = x0[..., 2] / (x0[..., 2] - x1[..., 2])
    pc = x0[..., 0:2] + s[..., None] * dx[..., 0:2]
    eps = d[..., 0] + d[..., 1] + (np.abs(dx[..., 0]) + np.abs(dx[..., 1])) * d[..., 2] / np.abs(dx[..., 2])
    return np.hypot(pc[..., 0], pc[..., 1]), eps


def old_delta(x0, v0, x1, v1, h):
    """Round 3's excursion figure (not a bound: VERDICT r03, weak #2), kept to show what the test catches."""
    dx = x1 - x0
    return (np.abs(h[..., None] * v0 - dx) + np.abs(h[..., None] * v1 - dx)).sum(-1)


def may_hit(x0, v0, x1, v1, h, hq3, r_in, r_out, slack=1.0 + 1e-9, fuzz=1e-11):
    """The device decision, division-free as the kernel writes it: compare |P| with R |D| -+ E, everything multiplied
    through by |D| = |z0 - z1|."""
    d = component_bound(x0, v0, x1, v1, h, hq3)
    dx = x1 - x0
    D = np.abs(x0[..., 2] - x1[..., 2])
    px = x0[..., 2] * x1[..., 0] - x1[..., 2] * x0[..., 0]
    py = x0[..., 2] * x1[..., 1] - x1[..., 2] * x0[..., 1]
    P2 = px * px + py * py
    E = ((d[..., 0] + d[..., 1]) * D + (np.abs(dx[..., 0]) + np.abs(dx[..., 1])) * d[..., 2]) * slack + fuzz * D
    lo = r_in * D - E
    hi = r_out * D + E
    return ~(((lo > 0.0) & (P2 < lo * lo)) | (P2 > hi * hi))


# ------------------------------------------------------------------------------------------------------------------
# scipy's RK45, many rays at once (rk.py:111-176 with select_initial_step, common.py:68-134): every ray has its own
# step size and accept / reject history; state layout [x, y, z, vx, vy, vz].  test_disk_filter_bound checks it
# against scipy.integrate.RK45 itself, step for step, before using it.
# ------------------------------------------------------------------------------------------------------------------
def accel(x, v, r_s, form):
    r2 = (x * x).sum(-1)
    kk = (v * v).sum(-1)
    xk = (x * v).sum(-1)
    r = np.sqrt(r2)
    if form == "reduced":
        c = -1.5 * r_s * (r2 * kk - xk * xk) / (r2 * r2 * r)
    else:
        nk = xk / r
        f = 1.0 - r_s / r
        hh = r_s / (r - r_s)
        hp = -r_s / ((r - r_s) * (r - r_s))
        kt2 = (kk + hh * nk * nk) / f
        s = 0.5 * f * (r_s / r2) * kt2 + 0.5 * f * hp * nk * nk + (r_s / r2) * (kk - nk * nk)
        c = -s / r
    return c[..., None] * x


def _f(y, r_s, form):
    return np.concatenate([y[..., 3:6], accel(y[..., 0:3], y[..., 3:6], r_s, form)], -1)


def _norm(a):
    return np.sqrt((a * a).sum(-1) / a.shape[-1])


def rk45_steps(y0, t_bound, rtol, atol, r_s=1.0, form="reduced", max_steps=400):
    """Generator over the ACCEPTED steps of all rays: yields (mask, y_old, y_new, h, Q) with Q [n, 6, 4] the dense
    output's coefficient matrix (y(theta) = y_old + h theta sum_m Q[:, m] theta^m).  Rays end at t_bound or inside
    r < 1.02 r_s."""
    y = np.array(y0, dtype=np.float64)
    n = len(y)
    t = np.zeros(n)
    f = _f(y, r_s, form)
    scale = atol + np.abs(y) * rtol
    d0, d1 = _norm(y / scale), _norm(f / scale)
    h0 = np.where((d0 < 1e-5) | (d1 < 1e-5), 1e-6, 0.01 * d0 / d1)
    h0 = np.minimum(h0, t_bound)
    y1 = y + h0[:, None] * f
    d2 = _norm((_f(y1, r_s, form) - f) / scale) / h0
    h1 = np.where((d1 <= 1e-15) & (d2 <= 1e-15), np.maximum(1e-6, h0 * 1e-3), (0.01 / np.maximum(d1, d2)) ** 0.2)
    h_abs = np.minimum(np.minimum(100 * h0, h1), t_bound)
    alive = np.ones(n, bool)
    rejected = np.zeros(n, bool)
    for _ in range(max_steps):
        if not alive.any():
            return
        t_new = np.minimum(t + h_abs, t_bound)
        h = t_new - t
        K = np.zeros((7, n, 6))
        K[0] = f
        for s in range(1, 6):
            dy = np.tensordot(A_[s, :s], K[:s], axes=(0, 0)) * h[:, None]
            K[s] = _f(y + dy, r_s, form)
        y_new = y + h[:, None] * np.tensordot(B_, K[:6], axes=(0, 0))
        f_new = _f(y_new, r_s, form)
        K[6] = f_new
        sc = atol + np.maximum(np.abs(y), np.abs(y_new)) * rtol
        err = _norm(np.tensordot(E_, K, axes=(0, 0)) * h[:, None] / sc)
        ok = alive & (err < 1.0)
        with np.errstate(divide="ignore", invalid="ignore"):
            fac_ok = np.where(err == 0.0, 10.0, np.minimum(10.0, 0.9 * err ** -0.2))
            fac_ok = np.where(rejected, np.minimum(1.0, fac_ok), fac_ok)
            fac_rej = np.maximum(0.2, 0.9 * err ** -0.2)
            h_next = np.where(ok, np.abs(h) * fac_ok, np.where(alive, np.abs(h) * fac_rej, h_abs))
        if ok.any():
            Q = np.einsum("snc,sm->ncm", K, P_)
            yield ok.copy(), y.copy(), y_new.copy(), h.copy(), Q
        h_abs = h_next
        rejected = np.where(ok, False, np.where(alive, True, rejected))
        y = np.where(ok[:, None], y_new, y)
        f = np.where(ok[:, None], f_new, f)
        t = np.where(ok, t_new, t)
        r = np.sqrt((y[:, 0:3] ** 2).sum(-1))
        alive &= (t < t_bound) & (r > 1.02 * r_s) & np.isfinite(err)


def dense_eval(y_old, h, Q, theta):
    """rk.py:552-574 for one theta per ray."""
    p = theta[:, None] ** np.arange(1, 5)[None, :]
    return y_old + h[:, None] * np.einsum("ncm,nm->nc", Q, p)


def dense_plane_crossings(y_old, h, Q):
    """All roots theta in [0, 1] of the dense output's z component, per ray: a list of arrays (the quartic's real
    roots in the interval, by numpy.roots -- independent of any bracketing search)."""
    out = []
    for i in range(len(h)):
        # z(theta) = z0 + h (Q0 th + Q1 th^2 + Q2 th^3 + Q3 th^4)
        c = np.array([h[i] * Q[i, 2, 3], h[i] * Q[i, 2, 2], h[i] * Q[i, 2, 1], h[i] * Q[i, 2, 0], y_old[i, 2]])
        r = np.roots(c)
        r = r[np.abs(r.imag) < 1e-9].real
        out.append(np.sort(r[(r >= -1e-12) & (r <= 1.0 + 1e-12)]).clip(0.0, 1.0))
    return out


# ------------------------------------------------------------------------------------------------------------------
# The Boyer-Lindquist forms (Kerr): disk_crossing_may_hit_bl (chord bound in (r, theta)) and, for what it lets through,
# disk_crossing_may_hit_sharp<BL> (the crossing of the cubic Hermite interpolant H + the exact D - H relation).
# State of a step end here: q = (r, theta, phi), u = d q / d lambda; hq3 = h * (column 3 of Q) for the three positions.
# ------------------------------------------------------------------------------------------------------------------
HALF_PI = 1.5707963267948966


def bl_plane_index(th):
    return np.floor((th - HALF_PI) * 0.3183098861837907)


def bl_chord_bound(q0, u0, q1, u1, h, hq3):
    """(crossed_one, th_star, r_lin, eps): the chord's r where it meets the plane theta* it crosses and the bound on
    |r_dense - r_lin| at ANY crossing of the dense output with that plane; crossed_one False = more than one plane (or
    none) between the step's ends: the device says "may hit" without looking further."""
    k0, k1 = bl_plane_index(q0[..., 1]), bl_plane_index(q1[..., 1])
    one = np.abs(k1 - k0) == 1.0
    th_star = np.pi * np.maximum(k0, k1) + HALF_PI
    dth, dr = q1[..., 1] - q0[..., 1], q1[..., 0] - q0[..., 0]
    with np.errstate(divide="ignore", invalid="ignore"):
        s = (th_star - q0[..., 1]) / dth
        r_lin = q0[..., 0] + s * dr
        d = component_bound(q0, u0, q1, u1, h, hq3)
        eps = d[..., 0] + np.abs(dr) * d[..., 1] / np.abs(dth)
    return one, th_star, r_lin, eps


def may_hit_bl(q0, u0, q1, u1, h, hq3, a, r_in, r_out, slack=1.0 + 1e-9, fuzz=1e-11):
    """The device decision of disk_crossing_may_hit_bl (annulus in sqrt(r^2 + a^2), compared in squares)."""
    one, _, r_lin, eps = bl_chord_bound(q0, u0, q1, u1, h, hq3)
    eps = eps * slack * (1.0 + 1e-12) + fuzz
    r_lo, r_hi = np.maximum(r_lin - eps, 0.0), r_lin + eps
    with np.errstate(invalid="ignore"):
        out = (r_hi * r_hi + a * a < r_in * r_in) | (r_lo * r_lo + a * a > r_out * r_out)
    return ~one | ~out


def sharp_bl(q0, u0, q1, u1, h, hq3):
    """disk_crossing_may_hit_sharp in Boyer-Lindquist coordinates: (decides, r_h, er) -- where `decides`, every crossing of
    the dense output with the plane lies at an r within er of r_h; elsewhere the device function returns "may hit"
    (several planes, H_theta not safely monotone)."""
    ce = 1
    k0, k1 = bl_plane_index(q0[..., 1]), bl_plane_index(q1[..., 1])
    one = np.abs(k1 - k0) == 1.0
    target = np.pi * np.maximum(k0, k1) + HALF_PI
    e4 = np.abs(hq3) * 0.0625
    d = q1 - q0
    m0, m1 = h[..., None] * u0, h[..., None] * u1
    b1, b2, b3 = m0, 3.0 * d - (2.0 * m0 + m1), -2.0 * d + (m0 + m1)
    z0, dz = q0[..., ce] - target, q1[..., ce] - q0[..., ce]
    g0, g1 = b1[..., ce], 3.0 * b3[..., ce] + (2.0 * b2[..., ce] + b1[..., ce])
    with np.errstate(divide="ignore", invalid="ignore"):
        the = -b2[..., ce] / (3.0 * b3[..., ce])
        ge = np.where((the > 0.0) & (the < 1.0), b2[..., ce] * the + b1[..., ce], g0)
        sgn = np.where(dz < 0.0, -1.0, 1.0)
        gmin = np.minimum(np.minimum(g0 * sgn, g1 * sgn), ge * sgn)
        decides = one & (gmin > 0.125 * np.abs(dz))
        th = np.clip(-z0 / dz, 0.0, 1.0)
        for _ in range(2):
            f = ((b3[..., ce] * th + b2[..., ce]) * th + b1[..., ce]) * th + z0
            df = (3.0 * b3[..., ce] * th + 2.0 * b2[..., ce]) * th + b1[..., ce]
            th = np.clip(th - f / df, 0.0, 1.0)
        f = ((b3[..., ce] * th + b2[..., ce]) * th + b1[..., ce]) * th + z0
        dth = (1.5 * e4[..., ce] + np.abs(f)) / gmin
        r_h = ((b3[..., 0] * th + b2[..., 0]) * th + b1[..., 0]) * th + q0[..., 0]
        er = ((np.abs(b1[..., 0]) + (2.0 * np.abs(b2[..., 0]) + 3.0 * np.abs(b3[..., 0]))) * dth + e4[..., 0]) * (1.0 + 1e-6)
    return decides, r_h, er


def may_hit_sharp_bl(q0, u0, q1, u1, h, hq3, a, r_in, r_out):
    decides, r_h, er = sharp_bl(q0, u0, q1, u1, h, hq3)
    r_lo, r_hi = np.maximum(r_h - er, 0.0), r_h + er
    with np.errstate(invalid="ignore"):
        out = (r_hi * r_hi + a * a < r_in * r_in) | (r_lo * r_lo + a * a > r_out * r_out)
    return ~decides | ~out
