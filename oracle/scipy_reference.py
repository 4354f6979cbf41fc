"""scipy/sympy-driven restatement of the reference's per-ray geodesic solve.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package imports this file; it is used
by tests/, by tests/golden/make_golden.py (the fixture generator) and nowhere else.

PARITY UNPINNED: the reference tree (bldevries/blackhole_geodesic_calculator) holds no
solver arithmetic, no tests and no golden vectors for this path.  The arithmetic lives in
the un-vendored, un-pinned third-party package ``curvedpy`` (reference README.md:23-24,
README.md:93; import at raytracer/RelativisticRenderEngine.py:7), which is not installable
here.  What this file follows instead, line by line:

* the ODE  dk^a/dl = -Gamma^a_{mu nu} k^mu k^nu,  dx^b/dl = k^b        (README.md:198-209)
* the Schwarzschild metric, used "in cartesian coordinates"            (README.md:162-174)
* Christoffel symbols "calculated using sympy"                         (README.md:133-135, :182-184)
* integrator = scipy.integrate.solve_ivp                               (README.md:196)
* null rays (time_like=False), horizon radius r_s = 2*mass             (raytracer/RelativisticRenderEngine.py:95, :134)
* state rows ordered k_x, x, k_y, y, k_z, z                            (raytracer/RelativisticRenderEngine.py:281, :301)
* call signature / what is consumed: last sample + 'hit_blackhole'     (raytracer/RelativisticRenderEngine.py:293-313)

scipy (1.15.3 here) *is* the third-party arithmetic for the RK update, so the C restatement
in oracle/geodesic_oracle.c is checked step-for-step against ``solve_ivp`` driven from here.
"""
from __future__ import annotations

import functools

import numpy as np
from scipy.integrate import solve_ivp

# state layout used by the reference's solver (RelativisticRenderEngine.py:301)
#   y = [k_x, x, k_y, y, k_z, z]
IK = (0, 2, 4)
IX = (1, 3, 5)


# --------------------------------------------------------------------------------------
# RHS form 1: generic sympy derivation (what README.md:176-184 describes)
# --------------------------------------------------------------------------------------
@functools.lru_cache(maxsize=4)
def sympy_christoffel_rhs():
    """Derive -Gamma^i_{mu nu} k^mu k^nu for the Cartesian Schwarzschild metric with sympy.

    Returns a numpy-lambdified function acc(x, y, z, kx, ky, kz, r_s) -> (ax, ay, az) with
    k^t eliminated through the null condition g_{mu nu} k^mu k^nu = 0 (time_like=False,
    RelativisticRenderEngine.py:134).  No hand simplification is applied: the expression
    is the raw contraction, common sub-expressions shared by ``cse`` only.
    """
    import sympy as sp

    t, x, y, z, rs = sp.symbols("t x y z r_s", real=True)
    kt, kx, ky, kz = sp.symbols("k_t k_x k_y k_z", real=True)
    X = [t, x, y, z]
    K = [kt, kx, ky, kz]
    r = sp.sqrt(x * x + y * y + z * z)
    f = 1 - rs / r
    n = [x / r, y / r, z / r]
    g = sp.zeros(4, 4)
    g[0, 0] = -f
    for i in range(3):
        for j in range(3):
            g[i + 1, j + 1] = (1 if i == j else 0) + (rs / (r - rs)) * n[i] * n[j]
    ginv = sp.zeros(4, 4)
    ginv[0, 0] = 1 / g[0, 0]
    g3 = g[1:, 1:]
    # symbolic 3x3 inverse by adjugate/determinant, not simplified (the default Gaussian
    # elimination spends minutes zero-testing pivots that contain sqrt)
    ginv[1:, 1:] = g3.adjugate() / g3.det(method="berkowitz")
    dg = [[[sp.diff(g[a, b], X[c]) for c in range(4)] for b in range(4)] for a in range(4)]

    def gamma(s, m, nu):
        return sp.Rational(1, 2) * sum(
            ginv[s, rho] * (dg[nu][rho][m] + dg[rho][m][nu] - dg[m][nu][rho]) for rho in range(4)
        )

    # null condition -> (k^t)^2, kept as one shared sub-expression
    kt2 = sum(g[i, j] * K[i] * K[j] for i in range(1, 4) for j in range(1, 4)) / f
    acc = []
    for i in range(1, 4):
        expr = 0
        for m in range(4):
            for nu in range(4):
                gam = gamma(i, m, nu)
                if gam == 0:
                    continue
                if (m == 0) != (nu == 0):
                    # static, diagonal-in-t metric: Gamma^i_{t j} is structurally zero
                    raise AssertionError("unexpected non-zero Gamma^i_{tj}")
                kk = kt2 if (m == 0 and nu == 0) else K[m] * K[nu]
                expr -= gam * kk
        acc.append(expr)
    fn = sp.lambdify((x, y, z, kx, ky, kz, rs), acc, modules="numpy", cse=True)
    return fn


# --------------------------------------------------------------------------------------
# RHS form 2/3: hand-reduced forms, numpy scalar arithmetic
# --------------------------------------------------------------------------------------
def acc_christoffel(x, k, r_s, mu2=0.0):
    """a = -n [ 1/2 f f' (k^t)^2 + 1/2 f h' (n.k)^2 + (r_s/r^2)(|k|^2 - (n.k)^2) ]; mu2 = -g(k, k): 0 for the engine's
    null rays (time_like=False), 1 for massive particles (time_like=True) -- it enters through (k^t)^2 only."""
    r2 = x[0] * x[0] + x[1] * x[1] + x[2] * x[2]
    r = np.sqrt(r2)
    nk = (x[0] * k[0] + x[1] * k[1] + x[2] * k[2]) / r
    kk = k[0] * k[0] + k[1] * k[1] + k[2] * k[2]
    f = 1.0 - r_s / r
    fp = r_s / r2
    h = r_s / (r - r_s)
    hp = -r_s / ((r - r_s) * (r - r_s))
    kt2 = (kk + h * nk * nk + mu2) / f
    s = 0.5 * f * fp * kt2 + 0.5 * f * hp * nk * nk + (r_s / r2) * (kk - nk * nk)
    c = -s / r
    return np.array([c * x[0], c * x[1], c * x[2]])


def acc_reduced(x, k, r_s, mu2=0.0):
    """a = -(3/2) r_s |x cross k|^2 x / r^5, L^2 recomputed from the current state (+ the Newtonian term
    -(r_s/2) mu2 x / r^3 for a massive particle: d^2 r/dtau^2 = -M/r^2 + L^2/r^3 - 3 M L^2/r^4)."""
    r2 = x[0] * x[0] + x[1] * x[1] + x[2] * x[2]
    kk = k[0] * k[0] + k[1] * k[1] + k[2] * k[2]
    xk = x[0] * k[0] + x[1] * k[1] + x[2] * k[2]
    L2 = r2 * kk - xk * xk
    c = -1.5 * r_s * L2 / (r2 * r2 * np.sqrt(r2)) - 0.5 * r_s * mu2 / (r2 * np.sqrt(r2))
    return np.array([c * x[0], c * x[1], c * x[2]])


def make_rhs(r_s, form="christoffel", time_like=False):
    """RHS f(t, y) on the 6-D state [k_x, x, k_y, y, k_z, z]."""
    mu2 = 1.0 if time_like else 0.0
    if form == "sympy":
        assert not time_like, "the sympy-derived contraction is the null one"
        fn = sympy_christoffel_rhs()

        def rhs(_t, y):
            a = fn(y[1], y[3], y[5], y[0], y[2], y[4], r_s)
            return np.array([a[0], y[0], a[1], y[2], a[2], y[4]], dtype=float)

    else:
        acc = {"christoffel": acc_christoffel, "reduced": acc_reduced}[form]

        def rhs(_t, y):
            a = acc((y[1], y[3], y[5]), (y[0], y[2], y[4]), r_s, mu2)
            return np.array([a[0], y[0], a[1], y[2], a[2], y[4]])

    return rhs


# --------------------------------------------------------------------------------------
# The per-ray solve (what calc_trajectory does, RelativisticRenderEngine.py:293-313)
# --------------------------------------------------------------------------------------
FLAG_HIT_HORIZON = 1
FLAG_START_INSIDE = 2
FLAG_REACHED_END = 4
FLAG_EXITED_SPHERE = 8
FLAG_MAX_STEPS = 16
FLAG_STEP_TOO_SMALL = 32
FLAG_NAN = 64
FLAG_HIT_DISK = 128
FLAG_HIT_OBJECT = 0x88


def trace_ray(k0, x0, r_s=1.0, lambda_end=50.0, max_step=np.inf, rtol=1e-3, atol=1e-6,
              form="christoffel", method="RK45", r_exit=0.0, nr_points_curve=None, disk=None, spheres=None, time_like=False):
    """Integrate one null (time_like=True: time-like, parameter = proper time) geodesic with scipy.solve_ivp; returns a dict.

    Events: horizon r - r_s = 0 (terminal, any direction); optional outward sphere exit
    r - r_exit = 0 (terminal, direction +1) as the Limited engine's ray_trace does
    (LimitedRelativisticRenderEngine.py:273-278).
    """
    k0 = np.asarray(k0, float)
    x0 = np.asarray(x0, float)
    out = {"nfev": 0, "n_accepted": 0, "n_attempted": 0, "t_end": 0.0}
    if np.sqrt(np.dot(x0, x0)) <= r_s:
        out.update(flags=FLAG_START_INSIDE | FLAG_HIT_HORIZON, end=np.concatenate([x0, k0]))
        return out
    y0 = np.array([k0[0], x0[0], k0[1], x0[1], k0[2], x0[2]])

    def ev_horizon(_t, y):
        return np.sqrt(y[1] * y[1] + y[3] * y[3] + y[5] * y[5]) - r_s

    ev_horizon.terminal = True
    events = [ev_horizon]
    if r_exit > 0.0:
        def ev_exit(_t, y):
            return np.sqrt(y[1] * y[1] + y[3] * y[3] + y[5] * y[5]) - r_exit

        ev_exit.terminal = True
        ev_exit.direction = 1.0
        events.append(ev_exit)
    if disk is not None:
        # thin disk in z = 0 (LimitedRelativisticRenderEngine.py:413-438): a NON-terminal scipy event
        # g = z; afterwards the crossings are visited in time order and the first one inside the
        # annulus R_in <= R <= R_out ends the ray there (what came after is discarded)
        def ev_disk(_t, y):
            return y[5]

        events.append(ev_disk)
    n_obj = 0
    if spheres is not None:
        # object spheres inside the curved region (the reference's collision stub,
        # RelativisticRenderEngine.py:304-305): terminal events g_j = |x - c_j| - rho_j, entering only.
        # scipy sees only sign changes between step ends; the C oracle's chord rule adds the
        # pass-through-in-one-step case, so the two agree when max_step is small against rho_j.
        # Inserted BEFORE the disk event so that the disk stays events[-1].
        def make_ev(c, rho):
            def ev(_t, y):
                return np.sqrt((y[1] - c[0]) ** 2 + (y[3] - c[1]) ** 2 + (y[5] - c[2]) ** 2) - rho
            ev.terminal = True
            ev.direction = -1.0
            return ev

        obj_events = [make_ev(np.asarray(sp[:3], float), float(sp[3])) for sp in spheres]
        n_obj = len(obj_events)
        pos = len(events) - (1 if disk is not None else 0)
        events[pos:pos] = obj_events
    t_eval = None
    if nr_points_curve:
        t_eval = np.linspace(0.0, lambda_end, nr_points_curve)
    sol = solve_ivp(make_rhs(r_s, form, time_like), (0.0, lambda_end), y0, method=method, events=events,
                    max_step=max_step, rtol=rtol, atol=atol, t_eval=t_eval)
    flags = 0
    if sol.status == 1:
        # the terminal event that stopped the solve is the one with the LAST root time recorded
        # (ivp.py:680-690 truncates the roots at the first terminal one)
        cands = [(sol.t_events[i][-1], i) for i in range(len(events) - (1 if disk is not None else 0))
                 if len(sol.t_events[i]) > 0]
        te, i_ev = min(cands)
        ye = sol.y_events[i_ev][-1]
        i_obj0 = 1 + (1 if r_exit > 0.0 else 0)
        if i_ev == 0:
            flags |= FLAG_HIT_HORIZON
        elif i_ev < i_obj0:
            flags |= FLAG_EXITED_SPHERE
        else:
            flags |= FLAG_HIT_OBJECT
            out["object_id"] = i_ev - i_obj0
    elif sol.status == 0:
        flags |= FLAG_REACHED_END
        te, ye = (sol.t[-1], sol.y[:, -1]) if t_eval is None else (lambda_end, None)
    else:
        flags |= FLAG_STEP_TOO_SMALL
        te, ye = (sol.t[-1], sol.y[:, -1])
    if ye is None:  # t_eval path: last grid point == lambda_end
        ye = sol.y[:, -1]
    if disk is not None:
        r_in, r_out = disk
        for td, yd in zip(sol.t_events[-1], sol.y_events[-1]):
            R = np.sqrt(yd[1] * yd[1] + yd[3] * yd[3])
            if r_in <= R <= r_out and td <= te:
                flags, te, ye = FLAG_HIT_DISK, td, yd
                # steps up to and including the one that contains the crossing
                n_acc = int(np.searchsorted(sol.t, td, side="left"))
                out["n_accepted_disk"] = n_acc
                break
    out.update(
        flags=flags,
        end=np.array([ye[1], ye[3], ye[5], ye[0], ye[2], ye[4]]),
        t_end=float(te),
        nfev=int(sol.nfev),
        n_attempted=(int(sol.nfev) - 2) // 6 if method == "RK45" else -1,
        n_accepted=(len(sol.t) - 1) if t_eval is None else -1,
        sol=sol,
    )
    if "n_accepted_disk" in out:
        out["n_accepted"] = out["n_accepted_disk"]
        out["n_attempted"] = -1  # scipy kept integrating past the disk: attempted count not comparable
    return out


def trace_rays(k0, x0, **kw):
    """Loop trace_ray over k0[N,3]; x0 is [3] (shared) or [N,3]."""
    k0 = np.atleast_2d(np.asarray(k0, float))
    x0 = np.asarray(x0, float)
    n = k0.shape[0]
    end = np.zeros((n, 6))
    flags = np.zeros(n, np.uint8)
    natt = np.zeros(n, np.uint32)
    nacc = np.zeros(n, np.uint32)
    tend = np.zeros(n)
    for i in range(n):
        xi = x0 if x0.ndim == 1 else x0[i]
        r = trace_ray(k0[i], xi, **kw)
        end[i] = r["end"]
        flags[i] = r["flags"]
        natt[i] = max(r["n_attempted"], 0)
        nacc[i] = max(r["n_accepted"], 0)
        tend[i] = r["t_end"]
    return {"end": end, "flags": flags, "n_attempted": natt, "n_accepted": nacc, "t_end": tend}


# --------------------------------------------------------------------------------------
# Converged solve for physics known-answer tests
# --------------------------------------------------------------------------------------
def trace_ray_converged(k0, x0, r_s=1.0, lambda_end=50.0, form="reduced"):
    return trace_ray(k0, x0, r_s=r_s, lambda_end=lambda_end, rtol=1e-12, atol=1e-14,
                     form=form, method="DOP853")


# --------------------------------------------------------------------------------------
# Kerr, Boyer-Lindquist coordinates (BASELINE.json config 5).  No reference arithmetic exists
# (README.md:218 lists Kerr as a goal; `a = 0.9` at CamEdition.py:210); the method is the
# reference's: metric -> sympy Christoffels -> solve_ivp.  The derivation lives in
# tools/gen_kerr_rhs.py and is shared with the C snippet the oracle and the kernel compile.
# --------------------------------------------------------------------------------------
@functools.lru_cache(maxsize=2)
def kerr_rhs_lambdified():
    import importlib.util
    import os

    import sympy as sp
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "gen_kerr_rhs.py")
    spec = importlib.util.spec_from_file_location("gen_kerr_rhs", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    syms, kt_expr, acc = mod.derive()
    r, th, ur, uth, uph, E, L, M, a, kt = syms
    acc = [e.subs(kt, kt_expr) for e in acc]
    return sp.lambdify((r, th, ur, uth, uph, E, L, M, a), acc + [kt_expr], modules="math", cse=True)


def kerr_metric(r, th, M, a):
    s2, c2 = np.sin(th) ** 2, np.cos(th) ** 2
    Sig = r * r + a * a * c2
    Del = r * r - 2 * M * r + a * a
    gtt = -(1 - 2 * M * r / Sig)
    gtp = -2 * M * a * r * s2 / Sig
    grr = Sig / Del
    gthth = Sig
    gpp = (r * r + a * a + 2 * M * a * a * r * s2 / Sig) * s2
    return gtt, gtp, grr, gthth, gpp


def cart_to_bl(x, k, a):
    """(x, y, z), k_cart -> (r, th, ph), (ur, uth, uph) with x = sqrt(r^2+a^2) sin th cos ph, z = r cos th."""
    x = np.asarray(x, float)
    rho2 = x @ x
    b = rho2 - a * a
    r = np.sqrt(0.5 * (b + np.sqrt(b * b + 4 * a * a * x[2] * x[2])))
    th = np.arccos(x[2] / r)
    ph = np.arctan2(x[1], x[0])
    J = bl_jacobian(r, th, ph, a)
    u = np.linalg.solve(J, np.asarray(k, float))
    return np.array([r, th, ph]), u


def bl_jacobian(r, th, ph, a):
    R = np.sqrt(r * r + a * a)
    st, ct, sp_, cp = np.sin(th), np.cos(th), np.sin(ph), np.cos(ph)
    return np.array([[r / R * st * cp, R * ct * cp, -R * st * sp_],
                     [r / R * st * sp_, R * ct * sp_, R * st * cp],
                     [ct, -r * st, 0.0]])


def bl_to_cart(q, u, a):
    r, th, ph = q
    R = np.sqrt(r * r + a * a)
    x = np.array([R * np.sin(th) * np.cos(ph), R * np.sin(th) * np.sin(ph), r * np.cos(th)])
    return x, bl_jacobian(r, th, ph, a) @ np.asarray(u)


def kerr_constants(q, u, M, a, mu2=0.0):
    """E = -k_t, L = k_phi from the norm condition g(k, k) = -mu2 at the start (future-directed root, g_tt < 0)."""
    gtt, gtp, grr, gthth, gpp = kerr_metric(q[0], q[1], M, a)
    S = grr * u[0] ** 2 + gthth * u[1] ** 2 + gpp * u[2] ** 2 + mu2
    B = gtp * u[2]
    kt = (-B - np.sqrt(B * B - gtt * S)) / gtt
    return -(gtt * kt + gtp * u[2]), gtp * kt + gpp * u[2], kt


KERR_HORIZON_MARGIN = 1e-3  # terminal event at r = r_plus (1 + margin): BL coordinates are singular at r_plus


def trace_ray_kerr(k0, x0, M=0.5, a=0.45, lambda_end=50.0, max_step=np.inf, rtol=1e-3, atol=1e-6, method="RK45",
                   disk=None, time_like=False, spheres=None):
    """One null geodesic in Kerr; state y = [ur, r, uth, th, uph, ph]; Cartesian in, Cartesian out.
    disk=(R_in, R_out): thin disk in the equatorial plane z = r cos(th) = 0, annulus in the cylindrical
    radius sqrt(x^2 + y^2) = sqrt(r^2 + a^2) |sin th|; a NON-terminal event g = cos(th), the first crossing
    inside the annulus ends the ray there (same rule as trace_ray's disk)."""
    fn = kerr_rhs_lambdified()
    q0, u0 = cart_to_bl(x0, k0, a)
    r_plus = M + np.sqrt(M * M - a * a)
    r_h = r_plus * (1 + KERR_HORIZON_MARGIN)
    out = {"nfev": 0, "n_accepted": 0, "n_attempted": 0, "t_end": 0.0}
    if q0[0] <= r_h:
        out.update(flags=FLAG_START_INSIDE | FLAG_HIT_HORIZON, end=np.concatenate([x0, k0]))
        return out
    E, L, _ = kerr_constants(q0, u0, M, a, 1.0 if time_like else 0.0)

    def rhs(_t, y):
        ar, ath, aph, _kt = fn(y[1], y[3], y[0], y[2], y[4], E, L, M, a)
        return np.array([ar, y[0], ath, y[2], aph, y[4]])

    def ev(_t, y):
        return y[1] - r_h

    ev.terminal = True
    events = [ev]
    n_obj = 0
    if spheres is not None:
        # object spheres (trace_ray's rule) met in the Cartesian frame: g_j = |x(r, th, ph) - c_j| - rho_j, entering only
        def make_ev(c, rho):
            def evs(_t, y):
                R = np.sqrt(y[1] * y[1] + a * a)
                x = np.array([R * np.sin(y[3]) * np.cos(y[5]), R * np.sin(y[3]) * np.sin(y[5]), y[1] * np.cos(y[3])])
                return np.sqrt(((x - c) ** 2).sum()) - rho
            evs.terminal = True
            evs.direction = -1.0
            return evs

        events += [make_ev(np.asarray(sp[:3], float), float(sp[3])) for sp in spheres]
        n_obj = len(spheres)
    if disk is not None:
        def ev_disk(_t, y):
            return np.cos(y[3])

        events.append(ev_disk)
    y0 = np.array([u0[0], q0[0], u0[1], q0[1], u0[2], q0[2]])
    sol = solve_ivp(rhs, (0.0, lambda_end), y0, method=method, events=events, max_step=max_step, rtol=rtol, atol=atol)
    n_acc_disk = None
    if sol.status == 1:
        cands = [(sol.t_events[i][-1], i) for i in range(1 + n_obj) if len(sol.t_events[i]) > 0]
        te, i_ev = min(cands)
        ye = sol.y_events[i_ev][-1]
        flags = FLAG_HIT_HORIZON if i_ev == 0 else FLAG_HIT_OBJECT
        if i_ev > 0:
            out["object_id"] = i_ev - 1
    elif sol.status == 0:
        flags, te, ye = FLAG_REACHED_END, sol.t[-1], sol.y[:, -1]
    else:
        flags, te, ye = FLAG_STEP_TOO_SMALL, sol.t[-1], sol.y[:, -1]
    if disk is not None:
        for td, yd in zip(sol.t_events[-1], sol.y_events[-1]):
            R = np.sqrt(yd[1] * yd[1] + a * a) * abs(np.sin(yd[3]))
            if disk[0] <= R <= disk[1] and td <= te:
                flags, te, ye = FLAG_HIT_DISK, td, yd
                n_acc_disk = int(np.searchsorted(sol.t, td, side="left"))
                break
    xe, ke = bl_to_cart((ye[1], ye[3], ye[5]), (ye[0], ye[2], ye[4]), a)
    out.update(flags=flags, end=np.concatenate([xe, ke]), end_bl=np.array([ye[1], ye[3], ye[5], ye[0], ye[2], ye[4]]),
               t_end=float(te), nfev=int(sol.nfev), n_attempted=(int(sol.nfev) - 2) // 6 if method == "RK45" else -1,
               n_accepted=len(sol.t) - 1, E=E, L=L, sol=sol)
    if n_acc_disk is not None:
        out["n_accepted"] = n_acc_disk
        out["n_attempted"] = -1  # scipy integrated on past the disk
    return out
