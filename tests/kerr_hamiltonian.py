"""An independent judge for the Kerr right-hand side: the coordinate acceleration of a null geodesic in
Boyer-Lindquist coordinates from Hamilton's equations of H = 1/2 g^{mu nu} p_mu p_nu with the textbook INVERSE
metric -- no Christoffel symbols, no sympy, nothing generated.  Derivatives of g^{ab} by the complex step
(exact to rounding).  Test infrastructure (CPU); shared by tests/test_oracle.py and the GPU parity tests."""
import numpy as np


def ginv(r, th, M, a):
    s2, c2 = np.sin(th) ** 2, np.cos(th) ** 2
    Sig, Del = r * r + a * a * c2, r * r - 2 * M * r + a * a
    gtt = -((r * r + a * a) ** 2 - Del * a * a * s2) / (Sig * Del)
    gtp = -2 * M * a * r / (Sig * Del)
    return gtt, gtp, Del / Sig, 1.0 / Sig, (Del - a * a * s2) / (Sig * Del * s2)


def metric(r, th, M, a):
    s2, c2 = np.sin(th) ** 2, np.cos(th) ** 2
    Sig, Del = r * r + a * a * c2, r * r - 2 * M * r + a * a
    return (-(1 - 2 * M * r / Sig), -2 * M * a * r * s2 / Sig, Sig / Del, Sig,
            (r * r + a * a + 2 * M * a * a * r * s2 / Sig) * s2)


def constants(q, u, M, a):
    """E = -k_t, L = k_phi of the null ray through (q, u): future-directed root of the null condition."""
    gtt, gtp, grr, gthth, gpp = metric(q[0], q[1], M, a)
    S = grr * u[0] ** 2 + gthth * u[1] ** 2 + gpp * u[2] ** 2
    B = gtp * u[2]
    kt = (-B - np.sqrt(B * B - gtt * S)) / gtt
    return -(gtt * kt + gtp * u[2]), gtp * kt + gpp * u[2]


def acceleration(q, u, M, a):
    """d^2 (r, theta, phi) / dlambda^2 at Boyer-Lindquist position q = (r, theta, phi) with velocity u, null ray."""
    r, th = float(q[0]), float(q[1])
    E, L = constants(q, u, M, a)
    _, _, grr_c, gthth_c, _ = metric(r, th, M, a)
    pr, pth = grr_c * u[0], gthth_c * u[1]
    h = 1e-30

    def ham(rr, tt):
        gtt, gtp, grr, gthth, gpp = ginv(rr, tt, M, a)
        return 0.5 * (gtt * E * E - 2 * gtp * E * L + grr * pr * pr + gthth * pth * pth + gpp * L * L)

    dpr = -ham(r + 1j * h, th).imag / h
    dpth = -ham(r, th + 1j * h).imag / h
    g0 = ginv(r, th, M, a)
    d_r = [x.imag / h for x in ginv(r + 1j * h, th, M, a)]
    d_th = [x.imag / h for x in ginv(r, th + 1j * h, M, a)]
    rd, thd = u[0], u[1]
    acc_r = (d_r[2] * rd + d_th[2] * thd) * pr + g0[2] * dpr
    acc_th = (d_r[3] * rd + d_th[3] * thd) * pth + g0[3] * dpth
    acc_ph = (-d_r[1] * E + d_r[4] * L) * rd + (-d_th[1] * E + d_th[4] * L) * thd
    return np.array([acc_r, acc_th, acc_ph])


def sample_points(n, M, a, seed=0):
    """Boyer-Lindquist positions from just outside the horizon to r = 60 M, away from the axis, with random velocities."""
    rng = np.random.default_rng(seed)
    r_plus = M + np.sqrt(M * M - a * a)
    r = np.where(rng.random(n) < 0.3, r_plus * (1.0 + rng.uniform(2e-3, 0.5, n)), M * rng.uniform(2.2, 60.0, n))
    th = rng.uniform(0.15, np.pi - 0.15, n)
    ph = rng.uniform(-np.pi, np.pi, n)
    q = np.stack([r, th, ph], 1)
    u = np.stack([rng.normal(size=n), rng.normal(size=n) / r, rng.normal(size=n) / (r * np.sin(th))], 1)
    # inside the ergosphere (g_tt > 0) a null ray must co-rotate: the null condition has a real root k^t only if
    # g_rr ur^2 + g_thth uth^2 <= uph^2 Delta sin^2(theta) / g_tt -- scale the poloidal part down to half that
    gtt, gtp, grr, gthth, gpp = metric(r, th, M, a)
    Del = r * r - 2 * M * r + a * a
    pol = grr * u[:, 0] ** 2 + gthth * u[:, 1] ** 2
    cap = np.where(gtt > 0, 0.5 * u[:, 2] ** 2 * Del * np.sin(th) ** 2 / np.where(gtt > 0, gtt, 1.0), np.inf)
    f = np.where(pol > cap, np.sqrt(cap / np.maximum(pol, 1e-300)), 1.0)
    u[:, 0] *= f
    u[:, 1] *= f
    return q, u
