/*
 * geodesic_oracle.c -- CPU restatement (plain C) of the reference's per-ray null-geodesic solve.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the *checker*: only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may build, load or call it.  The product path
 * (blackhole_geodesic_calculator_amd/) never links or imports it and has no CPU fallback.
 *
 * PARITY UNPINNED.  The reference tree holds no solver arithmetic, no tests and no golden
 * vectors for this path: the arithmetic is in the un-vendored, un-pinned third-party package
 * `curvedpy` (reference README.md:23-24, :93; raytracer/RelativisticRenderEngine.py:7), which
 * cannot be installed here.  This restatement therefore follows
 *   - the ODE and metric stated in the reference README            (README.md:162-174, :198-209)
 *   - the call-site contract of calc_trajectory                    (raytracer/RelativisticRenderEngine.py:134, :281, :293-313)
 *   - the sphere-exit semantics of the Limited engine's ray_trace  (raytracer/LimitedRelativisticRenderEngine.py:273-278)
 *   - its thin-disk test: first z sign change whose crossing point lies in the annulus
 *     R_in <= R <= R_out                                            (raytracer/LimitedRelativisticRenderEngine.py:413-438)
 *     -- located here on the step's dense output (a scipy event g = z) instead of by linear
 *     interpolation between trajectory samples (:419-421)
 *   - Kerr (BASELINE.json config 5; a goal of the reference, README.md:218, hinted at by `a = 0.9`,
 *     raytracer/RelativisticRenderEngineCamEdition.py:210): no reference arithmetic exists; the
 *     reference's METHOD (metric -> sympy Christoffels -> solve_ivp) is applied to the Kerr metric in
 *     Boyer-Lindquist coordinates by tools/gen_kerr_rhs.py, whose generated snippet kerr_rhs.inc is
 *     compiled here
 *   - scipy 1.15.3's RK45 (the integrator README.md:196 names):    scipy/integrate/_ivp/rk.py:14-71 (rk_step),
 *     :111-176 (_step_impl), :377-404 (tableau, dense output P), common.py:63-134 (norm,
 *     select_initial_step), ivp.py:51-76 (brentq event root), :109-126 (find_active_events),
 *     :673-694 (event handling), base.py:175-206 (step / finished test)
 * and it is pinned against scipy.integrate.solve_ivp itself (run in the build container, fixtures
 * committed under tests/golden/ with the generating script) and against physics known answers.
 *
 * Arithmetic mirrors scipy's order of operations (no FMA contraction: build with
 * -ffp-contract=off) so that accept/reject decisions and step counts agree with solve_ivp.
 *
 * State layout on the boundary: x0[3], k0[3] in; end[6] = {x, y, z, k_x, k_y, k_z} out
 * (the Cam edition's ray_end[..., 0:3] = position, [..., 3:6] = direction,
 *  raytracer/RelativisticRenderEngineCamEdition.py:225-228).
 */
#include <math.h>
#include <stdio.h>
static int bhgo_debug = 0;
void bhgo_set_debug(int v) { bhgo_debug = v; }
#include <float.h>
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define BHGO_FLAG_HIT_HORIZON 1u
#define BHGO_FLAG_START_INSIDE 2u
#define BHGO_FLAG_REACHED_END 4u
#define BHGO_FLAG_EXITED_SPHERE 8u
#define BHGO_FLAG_MAX_STEPS 16u
#define BHGO_FLAG_STEP_TOO_SMALL 32u
#define BHGO_FLAG_NAN 64u
#define BHGO_FLAG_HIT_DISK 128u
#define BHGO_FLAG_HIT_OBJECT 0x88u /* composite of EXITED_SPHERE | HIT_DISK, which cannot co-occur otherwise */
#define BHGO_MAX_SPHERES 8

#define BHGO_METHOD_DP54 0
#define BHGO_METHOD_RK4 1
#define BHGO_RHS_CHRISTOFFEL 0
#define BHGO_RHS_REDUCED 1
#define BHGO_RHS_KERR_BL 2
#define BHGO_KERR_HORIZON_MARGIN 1e-3

typedef struct {
    double r_s;        /* horizon radius = 2*mass (RelativisticRenderEngine.py:95) */
    double lambda_end; /* curve_end (RelativisticRenderEngine.py:62, :294) */
    double max_step;   /* max_step; +inf when the scene property is -1 (:59-60) */
    double rtol;       /* scipy default 1e-3 (rk.py:86) */
    double atol;       /* scipy default 1e-6 (rk.py:86) */
    double h_fixed;    /* RK4 only */
    double r_exit;     /* 0 = off; outward sphere exit radius (Limited engine) */
    int32_t method;    /* BHGO_METHOD_* */
    int32_t rhs_form;  /* BHGO_RHS_* */
    uint32_t max_steps; /* cap on attempted steps per ray, 0 = no cap */
    uint32_t time_like; /* 0: null geodesics, g(k, k) = 0 (time_like=False, RelativisticRenderEngine.py:134); 1: massive
                           particles, g(k, k) = -1 with the proper time as parameter (the constructor's other value) */
    double disk_r_in;  /* thin disk in the plane z = 0: annulus R_in <= R <= R_out; off when R_out <= 0 */
    double disk_r_out; /* (LimitedRelativisticRenderEngine.py:283-286, :413-438) */
    double spin;       /* Kerr a (length units, |a| < M = r_s/2); used by BHGO_RHS_KERR_BL only */
    /* objects inside the curved region: spheres {cx, cy, cz, radius} in BH-centred coordinates.  The
       reference only holds a stub for this ("NOW YOU DO COLLISION DETECTION", hit = False,
       RelativisticRenderEngine.py:304-305; README.md:225 lists it as done elsewhere); the build's rule:
       a ray that is outside sphere j at the start of an accepted step and either ends the step inside it or
       whose chord passes through it enters the sphere in that step; the entry point is the first root of
       |x(t) - c_j| - rho_j on the step's dense output, and the earliest terminal event of the step wins */
    int32_t n_spheres;
    int32_t reserved2;
    double spheres[BHGO_MAX_SPHERES][4];
} bhgo_params;

/* per-ray context: Schwarzschild-Cartesian rays need none of it; Kerr rays are integrated in
   Boyer-Lindquist (r, theta, phi) with k^t from the Killing constants E, L fixed at the camera */
typedef struct {
    int kerr;
    double E, L, M, a;
    double r_hor; /* horizon event radius: r_s, or r_plus (1 + margin) in Boyer-Lindquist */
} rayctx;

/* ---------------------------------------------------------------------------------------
 * RHS: the spatial acceleration -Gamma^i_{mu nu} k^mu k^nu with k^t from the null condition
 * (README.md:198-209; time_like=False at RelativisticRenderEngine.py:134).
 * ------------------------------------------------------------------------------------- */
static void acc_christoffel(const double x[3], const double k[3], double r_s, double mu2, double a[3])
{
    /* (mu2 = -g(k, k): 0 for the null rays of the engine, 1 for time_like=True -- the only place the norm enters) */
    /* a = -n [ 1/2 f f' (k^t)^2 + 1/2 f h' (n.k)^2 + (r_s/r^2)(|k|^2 - (n.k)^2) ],
       f = 1 - r_s/r, f' = r_s/r^2, h = r_s/(r - r_s), h' = -r_s/(r - r_s)^2,
       (k^t)^2 = (|k|^2 + h (n.k)^2)/f.  Singular at r = r_s, like the lambdified contraction. */
    double r2 = x[0] * x[0] + x[1] * x[1] + x[2] * x[2];
    double r = sqrt(r2);
    double nk = (x[0] * k[0] + x[1] * k[1] + x[2] * k[2]) / r;
    double kk = k[0] * k[0] + k[1] * k[1] + k[2] * k[2];
    double f = 1.0 - r_s / r;
    double fp = r_s / r2;
    double h = r_s / (r - r_s);
    double hp = -r_s / ((r - r_s) * (r - r_s));
    double kt2 = (kk + h * nk * nk + mu2) / f;
    double s = 0.5 * f * fp * kt2 + 0.5 * f * hp * nk * nk + (r_s / r2) * (kk - nk * nk);
    double c = -s / r;
    a[0] = c * x[0];
    a[1] = c * x[1];
    a[2] = c * x[2];
}

static void acc_reduced(const double x[3], const double k[3], double r_s, double mu2, double a[3])
{
    /* algebraically identical for null rays: a = -(3/2) r_s |x cross k|^2 x / r^5; a massive particle (mu2 = 1) feels
       the Newtonian term as well: d^2 r / dtau^2 = -M / r^2 + L^2 / r^3 - 3 M L^2 / r^4, i.e. - (r_s / 2) mu2 x / r^3 */
    double r2 = x[0] * x[0] + x[1] * x[1] + x[2] * x[2];
    double kk = k[0] * k[0] + k[1] * k[1] + k[2] * k[2];
    double xk = x[0] * k[0] + x[1] * k[1] + x[2] * k[2];
    double L2 = r2 * kk - xk * xk;
    double c = -1.5 * r_s * L2 / (r2 * r2 * sqrt(r2)) - 0.5 * r_s * mu2 / (r2 * sqrt(r2));
    a[0] = c * x[0];
    a[1] = c * x[1];
    a[2] = c * x[2];
}

/* y = [k_x, x, k_y, y, k_z, z]  (RelativisticRenderEngine.py:301) */
static void acc_kerr_bl(const rayctx *rc, const double q[3], const double u[3], double acc[3])
{
    const double r = q[0], th = q[1], ur = u[0], uth = u[1], uph = u[2];
    const double E = rc->E, L = rc->L, M = rc->M, a = rc->a;
    double ar, ath, aph, ktv;
#define KERR_RCP(x) (1.0 / (x))
#define KERR_RCP3(o0, x0, o1, x1, o2, x2) const double o0 = 1.0 / (x0), o1 = 1.0 / (x1), o2 = 1.0 / (x2)
#define KERR_SIN(x) sin(x)
#define KERR_COS(x) cos(x)
#include "kerr_rhs.inc"
#undef KERR_RCP
#undef KERR_RCP3
#undef KERR_SIN
#undef KERR_COS
    (void)ktv;
    acc[0] = ar;
    acc[1] = ath;
    acc[2] = aph;
}

static void rhs(const bhgo_params *p, const rayctx *rc, const double y[6], double dy[6])
{
    double x[3] = {y[1], y[3], y[5]};
    double k[3] = {y[0], y[2], y[4]};
    double a[3];
    if (rc->kerr) {
        acc_kerr_bl(rc, x, k, a);
        dy[0] = a[0];
        dy[1] = k[0];
        dy[2] = a[1];
        dy[3] = k[1];
        dy[4] = a[2];
        dy[5] = k[2];
        return;
    }
    if (p->rhs_form == BHGO_RHS_REDUCED)
        acc_reduced(x, k, p->r_s, p->time_like ? 1.0 : 0.0, a);
    else
        acc_christoffel(x, k, p->r_s, p->time_like ? 1.0 : 0.0, a);
    dy[0] = a[0];
    dy[1] = k[0];
    dy[2] = a[1];
    dy[3] = k[1];
    dy[4] = a[2];
    dy[5] = k[2];
}

static double radius_cart(const double y[6])
{
    return sqrt(y[1] * y[1] + y[3] * y[3] + y[5] * y[5]);
}

/* the radial coordinate the horizon / exit events watch */
static double radius(const rayctx *rc, const double y[6])
{
    return rc->kerr ? y[1] : radius_cart(y);
}

/* ---------------------------------------------------------------------------------------
 * Dormand-Prince 5(4) tableau -- scipy rk.py:377-404
 * ------------------------------------------------------------------------------------- */
/* C (stage abscissae) is not needed: the RHS is autonomous */
static const double A_[6][5] = {
    {0, 0, 0, 0, 0},
    {1.0 / 5, 0, 0, 0, 0},
    {3.0 / 40, 9.0 / 40, 0, 0, 0},
    {44.0 / 45, -56.0 / 15, 32.0 / 9, 0, 0},
    {19372.0 / 6561, -25360.0 / 2187, 64448.0 / 6561, -212.0 / 729, 0},
    {9017.0 / 3168, -355.0 / 33, 46732.0 / 5247, 49.0 / 176, -5103.0 / 18656}};
static const double B_[6] = {35.0 / 384, 0, 500.0 / 1113, 125.0 / 192, -2187.0 / 6784, 11.0 / 84};
static const double E_[7] = {-71.0 / 57600, 0,           71.0 / 16695, -71.0 / 1920,
                             17253.0 / 339200, -22.0 / 525, 1.0 / 40};
static const double P_[7][4] = {
    {1, -8048581381.0 / 2820520608.0, 8663915743.0 / 2820520608.0, -12715105075.0 / 11282082432.0},
    {0, 0, 0, 0},
    {0, 131558114200.0 / 32700410799.0, -68118460800.0 / 10900136933.0, 87487479700.0 / 32700410799.0},
    {0, -1754552775.0 / 470086768.0, 14199869525.0 / 1410260304.0, -10690763975.0 / 1880347072.0},
    {0, 127303824393.0 / 49829197408.0, -318862633887.0 / 49829197408.0, 701980252875.0 / 199316789632.0},
    {0, -282668133.0 / 205662961.0, 2019193451.0 / 616988883.0, -1453857185.0 / 822651844.0},
    {0, 40617522.0 / 29380423.0, -110615467.0 / 29380423.0, 69997945.0 / 29380423.0}};

#define SAFETY 0.9
#define MIN_FACTOR 0.2
#define MAX_FACTOR 10.0

static void pack_end(const double y[6], double end[6]);

/* RMS norm -- common.py:63-65 */
static double rms6(const double v[6])
{
    double s = 0.0;
    for (int i = 0; i < 6; i++) s += v[i] * v[i];
    return sqrt(s) / sqrt(6.0);
}

/* common.py:68-134 */
static double select_initial_step(const bhgo_params *p, const rayctx *rc, const double y0[6], const double f0[6],
                                  double t0, double t_bound, uint32_t *nfev)
{
    const int order = 4; /* error_estimator_order, rk.py:96-98 */
    double interval_length = fabs(t_bound - t0);
    if (interval_length == 0.0) return 0.0;
    double scale[6], v[6];
    for (int i = 0; i < 6; i++) scale[i] = p->atol + fabs(y0[i]) * p->rtol;
    for (int i = 0; i < 6; i++) v[i] = y0[i] / scale[i];
    double d0 = rms6(v);
    for (int i = 0; i < 6; i++) v[i] = f0[i] / scale[i];
    double d1 = rms6(v);
    double h0;
    if (d0 < 1e-5 || d1 < 1e-5)
        h0 = 1e-6;
    else
        h0 = 0.01 * d0 / d1;
    if (interval_length < h0) h0 = interval_length;
    double y1[6], f1[6];
    for (int i = 0; i < 6; i++) y1[i] = y0[i] + h0 * f0[i];
    rhs(p, rc, y1, f1);
    (*nfev)++;
    for (int i = 0; i < 6; i++) v[i] = (f1[i] - f0[i]) / scale[i];
    double d2 = rms6(v) / h0;
    double h1;
    if (d1 <= 1e-15 && d2 <= 1e-15) {
        h1 = h0 * 1e-3;
        if (h1 < 1e-6) h1 = 1e-6;
    } else {
        double dm = d1 > d2 ? d1 : d2;
        h1 = pow(0.01 / dm, 1.0 / (order + 1));
    }
    double h = 100 * h0;
    if (h1 < h) h = h1;
    if (interval_length < h) h = interval_length;
    if (p->max_step < h) h = p->max_step;
    return h;
}

/* rk.py:14-71 */
static void rk_step(const bhgo_params *p, const rayctx *rc, const double y[6], const double f[6], double h,
                    double K[7][6], double y_new[6], double f_new[6])
{
    memcpy(K[0], f, sizeof(double) * 6);
    for (int s = 1; s < 6; s++) {
        double ys[6];
        for (int i = 0; i < 6; i++) {
            double dot = 0.0;
            for (int j = 0; j < s; j++) dot += K[j][i] * A_[s][j];
            ys[i] = y[i] + dot * h;
        }
        rhs(p, rc, ys, K[s]);
    }
    for (int i = 0; i < 6; i++) {
        double dot = 0.0;
        for (int j = 0; j < 6; j++) dot += K[j][i] * B_[j];
        y_new[i] = y[i] + h * dot;
    }
    rhs(p, rc, y_new, f_new);
    memcpy(K[6], f_new, sizeof(double) * 6);
}

/* rk.py:552-574 with Q = K^T P (rk.py:176-178) */
typedef struct {
    double t_old, h;
    double y_old[6];
    double Q[6][4];
} dense_t;

static void dense_build(dense_t *d, double t_old, double t, const double y_old[6], double K[7][6])
{
    d->t_old = t_old;
    d->h = t - t_old;
    memcpy(d->y_old, y_old, sizeof(double) * 6);
    for (int i = 0; i < 6; i++)
        for (int m = 0; m < 4; m++) {
            double s = 0.0;
            for (int j = 0; j < 7; j++) s += K[j][i] * P_[j][m];
            d->Q[i][m] = s;
        }
}

static void dense_eval(const dense_t *d, double t, double y[6])
{
    double x = (t - d->t_old) / d->h;
    double pw[4];
    pw[0] = x;
    for (int m = 1; m < 4; m++) pw[m] = pw[m - 1] * x;
    for (int i = 0; i < 6; i++) {
        double s = 0.0;
        for (int m = 0; m < 4; m++) s += d->Q[i][m] * pw[m];
        y[i] = d->h * s + d->y_old[i];
    }
}

/* Hermite cubic used only by the fixed-step RK4 mode (scipy has no RK4; this mode is the
   build's own "R-fine" regime) */
typedef struct {
    double t_old, h;
    double y0[6], y1[6], f0[6], f1[6];
} hermite_t;

static void hermite_eval(const hermite_t *d, double t, double y[6])
{
    double s = (t - d->t_old) / d->h;
    double s2 = s * s, s3 = s2 * s;
    double h00 = 2 * s3 - 3 * s2 + 1, h10 = s3 - 2 * s2 + s, h01 = -2 * s3 + 3 * s2, h11 = s3 - s2;
    for (int i = 0; i < 6; i++)
        y[i] = h00 * d->y0[i] + h10 * d->h * d->f0[i] + h01 * d->y1[i] + h11 * d->h * d->f1[i];
}

/* event function g(t) = r(y(t)) - R evaluated through an interpolant */
typedef struct {
    int kind; /* 0 dense (DP54), 1 hermite (RK4) */
    const dense_t *dn;
    const hermite_t *hm;
    double R;
    int zmode; /* 0: g = r - R; 1: g = z (disk plane); 2: g = |x - c| - R (object sphere) */
    const rayctx *rc;
    double c[3];
} evfun_t;

/* position of a state in the Cartesian frame the object spheres live in: the state's own x, y, z, or -- Boyer-Lindquist
   states (r, theta, phi) -- x = sqrt(r^2 + a^2) sin th cos ph, y = ... sin ph, z = r cos th */
static void cart_position(const rayctx *rc, const double y[6], double out[3])
{
    if (!rc->kerr) {
        out[0] = y[1];
        out[1] = y[3];
        out[2] = y[5];
        return;
    }
    const double R = sqrt(y[1] * y[1] + rc->a * rc->a), st = sin(y[3]);
    out[0] = R * st * cos(y[5]);
    out[1] = R * st * sin(y[5]);
    out[2] = y[1] * cos(y[3]);
}

static double ev_eval(const evfun_t *e, double t)
{
    double y[6];
    if (e->kind == 0)
        dense_eval(e->dn, t, y);
    else
        hermite_eval(e->hm, t, y);
    if (e->zmode == 1) return e->rc->kerr ? cos(y[3]) : y[5]; /* z = r cos(theta), r > 0 */
    if (e->zmode == 2) {
        double xc[3];
        cart_position(e->rc, y, xc);
        double dx = xc[0] - e->c[0], dy = xc[1] - e->c[1], dz = xc[2] - e->c[2];
        return sqrt(dx * dx + dy * dy + dz * dz) - e->R;
    }
    return radius(e->rc, y) - e->R;
}

/* Brent's method as scipy.optimize.brentq runs it (xtol = rtol = 4 eps, ivp.py:74-75).
 * Attribution: a restatement of SciPy's scipy/optimize/Zeros/brentq.c ("Written by Charles Harris
 * charles.harris@sdl.usu.edu"; SciPy, Copyright (c) 2001-2002 Enthought, Inc., 2003- SciPy Developers, BSD 3-Clause
 * License, https://github.com/scipy/scipy/blob/main/LICENSE.txt), kept step for step -- variable names included --
 * because the checker has to return the very iterate solve_ivp returns. */
static double brentq(const evfun_t *e, double xa, double xb)
{
    const double xtol = 4 * DBL_EPSILON, rtol = 4 * DBL_EPSILON;
    double xpre = xa, xcur = xb, xblk = 0.0, fblk = 0.0, spre = 0.0, scur = 0.0;
    double fpre = ev_eval(e, xpre), fcur = ev_eval(e, xcur);
    if (fpre == 0.0) return xpre;
    if (fcur == 0.0) return xcur;
    for (int it = 0; it < 100; it++) {
        if (fpre != 0.0 && fcur != 0.0 && (signbit(fpre) != signbit(fcur))) {
            xblk = xpre;
            fblk = fpre;
            spre = scur = xcur - xpre;
        }
        if (fabs(fblk) < fabs(fcur)) {
            xpre = xcur;
            xcur = xblk;
            xblk = xpre;
            fpre = fcur;
            fcur = fblk;
            fblk = fpre;
        }
        double delta = (xtol + rtol * fabs(xcur)) / 2;
        double sbis = (xblk - xcur) / 2;
        if (fcur == 0.0 || fabs(sbis) < delta) return xcur;
        if (fabs(spre) > delta && fabs(fcur) < fabs(fpre)) {
            double stry;
            if (xpre == xblk) {
                stry = -fcur * (xcur - xpre) / (fcur - fpre);
            } else {
                double dpre = (fpre - fcur) / (xpre - xcur);
                double dblk = (fblk - fcur) / (xblk - xcur);
                stry = -fcur * (fblk * dblk - fpre * dpre) / (dblk * dpre * (fblk - fpre));
            }
            double lim = fabs(spre);
            double alt = 3 * fabs(sbis) - delta;
            if (alt < lim) lim = alt;
            if (2 * fabs(stry) < lim) {
                spre = scur;
                scur = stry;
            } else {
                spre = sbis;
                scur = sbis;
            }
        } else {
            spre = sbis;
            scur = sbis;
        }
        xpre = xcur;
        fpre = fcur;
        if (fabs(scur) > delta)
            xcur += scur;
        else
            xcur += (sbis > 0 ? delta : -delta);
        fcur = ev_eval(e, xcur);
    }
    return xcur;
}

static void pack_end(const double y[6], double end[6])
{
    end[0] = y[1];
    end[1] = y[3];
    end[2] = y[5];
    end[3] = y[0];
    end[4] = y[2];
    end[5] = y[4];
}

/* Events after an accepted step (ivp.py:109-126, :673-694).  Horizon, sphere exit and object spheres
   are terminal; the disk-plane crossing g = z is terminal only when the crossing point lies in the annulus
   (LimitedRelativisticRenderEngine.py:423-424), otherwise the ray carries on.  Of the terminal candidates
   the earliest root wins (handle_events sorts the roots, ivp.py:111-122); ties keep the order horizon,
   exit, disk, sphere 0, 1, ...  Returns 0 = carry on, else the flag; *t_root, y_root, *obj filled. */
static int e_kerr(const evfun_t *b) { return b->rc->kerr; }

static uint32_t check_events(const bhgo_params *p, double g_h, double g_h_new, double g_e,
                             double g_e_new, const double y_old[6], const double y_new[6], const evfun_t *base,
                             double t_old, double t, double *t_root, double y_root[6], int *obj)
{
    /* disk plane z = 0: Cartesian z, or the sign of cos(theta) in Boyer-Lindquist coordinates */
    const double z_old = e_kerr(base) ? cos(y_old[3]) : y_old[5], z_new = e_kerr(base) ? cos(y_new[3]) : y_new[5];
    int hor = ((g_h <= 0) && (g_h_new >= 0)) || ((g_h >= 0) && (g_h_new <= 0));
    int ext = (p->r_exit > 0.0) && (g_e <= 0) && (g_e_new >= 0); /* direction = +1 */
    int dsk = (p->disk_r_out > 0.0) && (((z_old <= 0) && (z_new >= 0)) || ((z_old >= 0) && (z_new <= 0)));
    double best = INFINITY;
    uint32_t best_flag = 0;
    int best_obj = -1;
    evfun_t e = *base;
    if (hor) {
        e.R = e.rc->r_hor;
        e.zmode = 0;
        double r = brentq(&e, t_old, t);
        if (r < best) { best = r; best_flag = BHGO_FLAG_HIT_HORIZON; }
    }
    if (ext) {
        e.R = p->r_exit;
        e.zmode = 0;
        double r = brentq(&e, t_old, t);
        if (r < best) { best = r; best_flag = BHGO_FLAG_EXITED_SPHERE; }
    }
    if (dsk) {
        e.zmode = 1;
        double r = brentq(&e, t_old, t);
        double y[6];
        if (e.kind == 0) dense_eval(e.dn, r, y); else hermite_eval(e.hm, r, y);
        double R = e.rc->kerr ? sqrt(y[1] * y[1] + e.rc->a * e.rc->a) * fabs(sin(y[3])) : sqrt(y[1] * y[1] + y[3] * y[3]);
        if (R >= p->disk_r_in && R <= p->disk_r_out && r < best) { best = r; best_flag = BHGO_FLAG_HIT_DISK; }
    }
    double c_old[3], c_new[3]; /* the step's ends in the spheres' (Cartesian) frame */
    if (p->n_spheres > 0) {
        cart_position(e.rc, y_old, c_old);
        cart_position(e.rc, y_new, c_new);
    }
    for (int j = 0; j < p->n_spheres; j++) {
        const double *sp = p->spheres[j];
        const double rho2 = sp[3] * sp[3];
        double a0[3] = {c_old[0] - sp[0], c_old[1] - sp[1], c_old[2] - sp[2]};
        double a1[3] = {c_new[0] - sp[0], c_new[1] - sp[1], c_new[2] - sp[2]};
        double d0 = a0[0] * a0[0] + a0[1] * a0[1] + a0[2] * a0[2]; /* squared distances: no root needed to decide */
        double d1 = a1[0] * a1[0] + a1[1] * a1[1] + a1[2] * a1[2];
        if (!(d0 > rho2)) continue; /* started inside (or on) this sphere: not an entry */
        double hi = t;
        if (!(d1 <= rho2)) {
            /* both ends outside: does the chord pass through?  closest point of the chord to the centre at
               s* = b / cc in (0, 1), its squared distance d0 - b^2 / cc < rho^2 (written without the division) */
            double ch[3] = {a1[0] - a0[0], a1[1] - a0[1], a1[2] - a0[2]};
            double cc = ch[0] * ch[0] + ch[1] * ch[1] + ch[2] * ch[2];
            double bb = -(a0[0] * ch[0] + a0[1] * ch[1] + a0[2] * ch[2]);
            if (!(bb > 0 && bb < cc && (d0 - rho2) * cc < bb * bb)) continue;
            hi = t_old + (bb / cc) * (t - t_old);
            e.zmode = 2;
            e.R = sp[3];
            memcpy(e.c, sp, sizeof(double) * 3);
            if (!(ev_eval(&e, hi) < 0)) continue; /* the curve itself stays outside there: no hit */
        }
        e.zmode = 2;
        e.R = sp[3];
        memcpy(e.c, sp, sizeof(double) * 3);
        double r = brentq(&e, t_old, hi);
        if (r < best) { best = r; best_flag = BHGO_FLAG_HIT_OBJECT; best_obj = j; }
    }
    if (!best_flag) return 0;
    *t_root = best;
    if (e.kind == 0) dense_eval(e.dn, best, y_root); else hermite_eval(e.hm, best, y_root);
    if (obj) *obj = best_obj;
    return best_flag;
}

typedef struct {
    double end[6];
    double t_end;
    uint32_t flags, n_attempted, n_accepted, nfev;
    int object_id; /* sphere index for BHGO_FLAG_HIT_OBJECT, else -1 */
} ray_result;

/* optional trajectory sampler: t_eval = linspace(0, lambda_end, T) like the engine's
   nr_points_curve request (RelativisticRenderEngine.py:294); samples with t_eval <= t are emitted
   through the step's dense output after every accepted step (ivp.py:706-723) */
typedef struct {
    uint32_t T, next;
    double lambda_end;
    double *out; /* [6][T]: x, y, z, k_x, k_y, k_z rows (Kerr: converted by the caller) */
} sampler_t;

static double sampler_time(const sampler_t *sm, uint32_t j)
{
    if (j + 1 == sm->T) return sm->lambda_end;
    return (double)j * (sm->lambda_end / (double)(sm->T - 1));
}

static void sampler_emit(sampler_t *sm, const dense_t *dn, double t_upto)
{
    while (sm->next < sm->T && sampler_time(sm, sm->next) <= t_upto) {
        double y[6], e[6];
        dense_eval(dn, sampler_time(sm, sm->next), y);
        pack_end(y, e);
        for (int c = 0; c < 6; c++) sm->out[(size_t)c * sm->T + sm->next] = e[c];
        sm->next++;
    }
}

/* ---------------------------------------------------------------------------------------
 * One ray, adaptive DP5(4): RungeKutta.__init__ (rk.py:84-104) + solve_ivp loop
 * (ivp.py:654-723) + _step_impl (rk.py:111-176)
 * ------------------------------------------------------------------------------------- */
static void trace_dp54(const bhgo_params *p, const rayctx *rc, const double x0[3], const double k0[3], ray_result *res,
                       sampler_t *sm)
{
    double y[6] = {k0[0], x0[0], k0[1], x0[1], k0[2], x0[2]};
    double f[6], K[7][6], y_new[6], f_new[6];
    const double t_bound = p->lambda_end;
    double t = 0.0;
    memset(res, 0, sizeof(*res));
    res->object_id = -1;
    rhs(p, rc, y, f);
    res->nfev = 1;
    double h_abs = select_initial_step(p, rc, y, f, t, t_bound, &res->nfev);
    double g_h = radius(rc, y) - rc->r_hor;
    double g_e = radius(rc, y) - p->r_exit;
    const uint32_t cap = p->max_steps ? p->max_steps : 0xFFFFFFFFu;

    for (;;) {
        if (t == t_bound) { /* base.py:189-194 */
            res->flags |= BHGO_FLAG_REACHED_END;
            break;
        }
        double min_step = 10 * fabs(nextafter(t, INFINITY) - t);
        if (h_abs > p->max_step)
            h_abs = p->max_step;
        else if (h_abs < min_step)
            h_abs = min_step;
        int step_rejected = 0, failed = 0;
        double h, t_new;
        for (;;) {
            if (h_abs < min_step) {
                failed = BHGO_FLAG_STEP_TOO_SMALL;
                break;
            }
            if (res->n_attempted >= cap) {
                failed = BHGO_FLAG_MAX_STEPS;
                break;
            }
            h = h_abs;
            t_new = t + h;
            if (t_new - t_bound > 0) t_new = t_bound;
            h = t_new - t;
            h_abs = fabs(h);
            rk_step(p, rc, y, f, h, K, y_new, f_new);
            res->nfev += 6;
            res->n_attempted++;
            double ev[6];
            for (int i = 0; i < 6; i++) {
                double ay = fabs(y[i]), an = fabs(y_new[i]);
                double scale = p->atol + (ay > an ? ay : an) * p->rtol; /* np.maximum: NaN propagates */
                if (isnan(ay) || isnan(an)) scale = NAN;
                double dot = 0.0;
                for (int j = 0; j < 7; j++) dot += K[j][i] * E_[j];
                ev[i] = dot * h / scale;
            }
            double error_norm = rms6(ev);
            if (bhgo_debug) fprintf(stderr, "oracle att %u t %.17g h %.17g errsq %.17g\n", res->n_attempted, t, h, error_norm * error_norm);
            if (error_norm < 1) {
                double factor;
                if (error_norm == 0)
                    factor = MAX_FACTOR;
                else {
                    factor = SAFETY * pow(error_norm, -0.2);
                    if (factor > MAX_FACTOR) factor = MAX_FACTOR;
                }
                if (step_rejected && factor > 1) factor = 1;
                h_abs *= factor;
                break;
            } else {
                double fac = SAFETY * pow(error_norm, -0.2);
                /* python max(MIN_FACTOR, nan) == MIN_FACTOR */
                if (!(fac > MIN_FACTOR)) fac = MIN_FACTOR;
                h_abs *= fac;
                step_rejected = 1;
            }
        }
        if (failed) {
            res->flags |= (uint32_t)failed;
            break;
        }
        double t_old = t;
        double y_old[6];
        memcpy(y_old, y, sizeof(y));
        t = t_new;
        memcpy(y, y_new, sizeof(y));
        memcpy(f, f_new, sizeof(f));
        res->n_accepted++;

        double g_h_new = radius(rc, y) - rc->r_hor;
        double g_e_new = radius(rc, y) - p->r_exit;
        double z_old = rc->kerr ? cos(y_old[3]) : y_old[5], z_new = rc->kerr ? cos(y[3]) : y[5];
        dense_t dn;
        evfun_t base = {0, &dn, NULL, 0.0, 0, rc, {0, 0, 0}};
        int any = (((g_h <= 0) && (g_h_new >= 0)) || ((g_h >= 0) && (g_h_new <= 0))) ||
                  ((p->r_exit > 0.0) && (g_e <= 0) && (g_e_new >= 0)) ||
                  ((p->disk_r_out > 0.0) && (((z_old <= 0) && (z_new >= 0)) || ((z_old >= 0) && (z_new <= 0)))) ||
                  (p->n_spheres > 0);
        if (any || sm) dense_build(&dn, t_old, t, y_old, K);
        if (any) {
            double t_root, y_root[6];
            uint32_t fl = check_events(p, g_h, g_h_new, g_e, g_e_new, y_old, y, &base, t_old, t, &t_root, y_root, &res->object_id);
            if (fl) {
                res->flags |= fl;
                t = t_root;
                memcpy(y, y_root, sizeof(y));
                if (sm) sampler_emit(sm, &dn, t);
                break;
            }
        }
        if (sm) sampler_emit(sm, &dn, t);
        g_h = g_h_new;
        g_e = g_e_new;
        if (t - t_bound >= 0) { /* base.py:203-204 */
            res->flags |= BHGO_FLAG_REACHED_END;
            break;
        }
    }
    int bad = 0;
    for (int i = 0; i < 6; i++) bad |= !isfinite(y[i]);
    if (bad) res->flags |= BHGO_FLAG_NAN;
    pack_end(y, res->end);
    res->t_end = t;
}

/* ---------------------------------------------------------------------------------------
 * One ray, classic fixed-step RK4 (the build's own "R-fine" regime; not a scipy method)
 * ------------------------------------------------------------------------------------- */
/* samples of a fixed step: the step's cubic Hermite interpolant (the fixed-step regime's dense output, also what its
   events are located on) */
static void sampler_emit_hermite(sampler_t *sm, const hermite_t *hm, double t_upto)
{
    while (sm->next < sm->T && sampler_time(sm, sm->next) <= t_upto) {
        double y[6], e[6];
        hermite_eval(hm, sampler_time(sm, sm->next), y);
        pack_end(y, e);
        for (int c = 0; c < 6; c++) sm->out[(size_t)c * sm->T + sm->next] = e[c];
        sm->next++;
    }
}

static void trace_rk4(const bhgo_params *p, const rayctx *rc, const double x0[3], const double k0[3], ray_result *res,
                      sampler_t *sm)
{
    double y[6] = {k0[0], x0[0], k0[1], x0[1], k0[2], x0[2]};
    double f[6];
    const double t_bound = p->lambda_end;
    double t = 0.0;
    memset(res, 0, sizeof(*res));
    res->object_id = -1;
    rhs(p, rc, y, f);
    res->nfev = 1;
    double g_h = radius(rc, y) - rc->r_hor;
    double g_e = radius(rc, y) - p->r_exit;
    const uint32_t cap = p->max_steps ? p->max_steps : 0xFFFFFFFFu;
    for (;;) {
        if (t >= t_bound) {
            res->flags |= BHGO_FLAG_REACHED_END;
            break;
        }
        if (res->n_attempted >= cap) {
            res->flags |= BHGO_FLAG_MAX_STEPS;
            break;
        }
        double h = p->h_fixed;
        double t_new = t + h;
        if (t_new - t_bound > 0) t_new = t_bound;
        h = t_new - t;
        double k2[6], k3[6], k4[6], ys[6], y_new[6], f_new[6];
        for (int i = 0; i < 6; i++) ys[i] = y[i] + (0.5 * h) * f[i];
        rhs(p, rc, ys, k2);
        for (int i = 0; i < 6; i++) ys[i] = y[i] + (0.5 * h) * k2[i];
        rhs(p, rc, ys, k3);
        for (int i = 0; i < 6; i++) ys[i] = y[i] + h * k3[i];
        rhs(p, rc, ys, k4);
        for (int i = 0; i < 6; i++) y_new[i] = y[i] + (h / 6.0) * (f[i] + 2.0 * k2[i] + 2.0 * k3[i] + k4[i]);
        rhs(p, rc, y_new, f_new);
        res->nfev += 4;
        res->n_attempted++;
        res->n_accepted++;
        hermite_t hm;
        hm.t_old = t;
        hm.h = h;
        memcpy(hm.y0, y, sizeof(y));
        memcpy(hm.y1, y_new, sizeof(y));
        memcpy(hm.f0, f, sizeof(f));
        memcpy(hm.f1, f_new, sizeof(f));
        double t_old = t;
        t = t_new;
        memcpy(y, y_new, sizeof(y));
        memcpy(f, f_new, sizeof(f));
        double g_h_new = radius(rc, y) - rc->r_hor;
        double g_e_new = radius(rc, y) - p->r_exit;
        evfun_t base = {1, NULL, &hm, 0.0, 0, rc, {0, 0, 0}};
        double t_root, y_root[6];
        uint32_t fl = check_events(p, g_h, g_h_new, g_e, g_e_new, hm.y0, hm.y1, &base, t_old, t, &t_root, y_root, &res->object_id);
        if (fl) {
            res->flags |= fl;
            t = t_root;
            memcpy(y, y_root, sizeof(y));
            if (sm) sampler_emit_hermite(sm, &hm, t);
            break;
        }
        g_h = g_h_new;
        g_e = g_e_new;
        int bad = 0;
        for (int i = 0; i < 6; i++) bad |= !isfinite(y[i]);
        if (bad) break;     /* (a step that ends in a non-finite state yields no samples) */
        if (sm) sampler_emit_hermite(sm, &hm, t);
    }
    int bad = 0;
    for (int i = 0; i < 6; i++) bad |= !isfinite(y[i]);
    if (bad) res->flags |= BHGO_FLAG_NAN;
    pack_end(y, res->end);
    res->t_end = t;
}

/* Boyer-Lindquist <-> Cartesian: x = sqrt(r^2+a^2) sin th cos ph, y = ... sin ph, z = r cos th */
static void bl_jacobian(double r, double th, double ph, double a, double J[3][3])
{
    double R = sqrt(r * r + a * a), st = sin(th), ct = cos(th), sp = sin(ph), cp = cos(ph);
    J[0][0] = r / R * st * cp;
    J[0][1] = R * ct * cp;
    J[0][2] = -R * st * sp;
    J[1][0] = r / R * st * sp;
    J[1][1] = R * ct * sp;
    J[1][2] = R * st * cp;
    J[2][0] = ct;
    J[2][1] = -r * st;
    J[2][2] = 0.0;
}

static void cart_to_bl(const double x[3], const double k[3], double a, double q[3], double u[3])
{
    double rho2 = x[0] * x[0] + x[1] * x[1] + x[2] * x[2];
    double b = rho2 - a * a;
    double r = sqrt(0.5 * (b + sqrt(b * b + 4 * a * a * x[2] * x[2])));
    q[0] = r;
    q[1] = acos(x[2] / r);
    q[2] = atan2(x[1], x[0]);
    double J[3][3];
    bl_jacobian(q[0], q[1], q[2], a, J);
    /* u = J^-1 k by Cramer's rule */
    double det = J[0][0] * (J[1][1] * J[2][2] - J[1][2] * J[2][1]) - J[0][1] * (J[1][0] * J[2][2] - J[1][2] * J[2][0]) +
                 J[0][2] * (J[1][0] * J[2][1] - J[1][1] * J[2][0]);
    u[0] = (k[0] * (J[1][1] * J[2][2] - J[1][2] * J[2][1]) - J[0][1] * (k[1] * J[2][2] - J[1][2] * k[2]) +
            J[0][2] * (k[1] * J[2][1] - J[1][1] * k[2])) / det;
    u[1] = (J[0][0] * (k[1] * J[2][2] - J[1][2] * k[2]) - k[0] * (J[1][0] * J[2][2] - J[1][2] * J[2][0]) +
            J[0][2] * (J[1][0] * k[2] - k[1] * J[2][0])) / det;
    u[2] = (J[0][0] * (J[1][1] * k[2] - k[1] * J[2][1]) - J[0][1] * (J[1][0] * k[2] - k[1] * J[2][0]) +
            k[0] * (J[1][0] * J[2][1] - J[1][1] * J[2][0])) / det;
}

static void bl_to_cart(const double q[3], const double u[3], double a, double x[3], double k[3])
{
    double R = sqrt(q[0] * q[0] + a * a);
    x[0] = R * sin(q[1]) * cos(q[2]);
    x[1] = R * sin(q[1]) * sin(q[2]);
    x[2] = q[0] * cos(q[1]);
    double J[3][3];
    bl_jacobian(q[0], q[1], q[2], a, J);
    for (int i = 0; i < 3; i++) k[i] = J[i][0] * u[0] + J[i][1] * u[1] + J[i][2] * u[2];
}

static void trace_one(const bhgo_params *p_in, const double x0[3], const double k0[3], ray_result *res, sampler_t *sm)
{
    /* validate_tol (scipy _ivp/common.py:44-51): an rtol below 100 eps is raised to 100 eps (scipy warns and carries on) */
    bhgo_params pc = *p_in;
    if (pc.rtol < 100.0 * DBL_EPSILON) pc.rtol = 100.0 * DBL_EPSILON;
    const bhgo_params *p = &pc;
    rayctx rc;
    memset(&rc, 0, sizeof(rc));
    rc.r_hor = p->r_s;
    if (p->rhs_form == BHGO_RHS_KERR_BL) {
        const double M = 0.5 * p->r_s, a = p->spin;
        rc.kerr = 1;
        rc.M = M;
        rc.a = a;
        rc.r_hor = (M + sqrt(M * M - a * a)) * (1.0 + BHGO_KERR_HORIZON_MARGIN);
        double q[3], u[3];
        cart_to_bl(x0, k0, a, q, u);
        if (q[0] <= rc.r_hor) {
            memset(res, 0, sizeof(*res));
            res->flags = BHGO_FLAG_START_INSIDE | BHGO_FLAG_HIT_HORIZON;
            memcpy(res->end, x0, sizeof(double) * 3);
            memcpy(res->end + 3, k0, sizeof(double) * 3);
            return;
        }
        /* E = -k_t, L = k_phi from the norm condition g(k, k) = -mu2 at the camera (future-directed root, g_tt < 0) */
        double r = q[0], th = q[1], s2 = sin(th) * sin(th), c2 = cos(th) * cos(th);
        double Sig = r * r + a * a * c2, Del = r * r - 2 * M * r + a * a;
        double gtt = -(1 - 2 * M * r / Sig), gtp = -2 * M * a * r * s2 / Sig, grr = Sig / Del, gthth = Sig;
        double gpp = (r * r + a * a + 2 * M * a * a * r * s2 / Sig) * s2;
        double S = grr * u[0] * u[0] + gthth * u[1] * u[1] + gpp * u[2] * u[2] + (p->time_like ? 1.0 : 0.0);
        double B = gtp * u[2];
        double kt = (-B - sqrt(B * B - gtt * S)) / gtt;
        rc.E = -(gtt * kt + gtp * u[2]);
        rc.L = gtp * kt + gpp * u[2];
        if (p->method == BHGO_METHOD_RK4)
            trace_rk4(p, &rc, q, u, res, sm);
        else
            trace_dp54(p, &rc, q, u, res, sm);
        /* res->end is {r, th, ph, ur, uth, uph}: back to Cartesian */
        double xe[3], ke[3];
        bl_to_cart(res->end, res->end + 3, a, xe, ke);
        memcpy(res->end, xe, sizeof(xe));
        memcpy(res->end + 3, ke, sizeof(ke));
        if (sm)
            for (uint32_t j = 0; j < sm->next; j++) {
                double qq[3], uu[3];
                for (int c = 0; c < 3; c++) {
                    qq[c] = sm->out[(size_t)c * sm->T + j];
                    uu[c] = sm->out[(size_t)(3 + c) * sm->T + j];
                }
                bl_to_cart(qq, uu, a, xe, ke);
                for (int c = 0; c < 3; c++) {
                    sm->out[(size_t)c * sm->T + j] = xe[c];
                    sm->out[(size_t)(3 + c) * sm->T + j] = ke[c];
                }
            }
        return;
    }
    double r0 = sqrt(x0[0] * x0[0] + x0[1] * x0[1] + x0[2] * x0[2]);
    if (r0 <= p->r_s) {
        /* 'start_inside_hole' -> (False, True, [], []) at RelativisticRenderEngine.py:311-313 */
        memset(res, 0, sizeof(*res));
        res->flags = BHGO_FLAG_START_INSIDE | BHGO_FLAG_HIT_HORIZON;
        res->end[0] = x0[0];
        res->end[1] = x0[1];
        res->end[2] = x0[2];
        res->end[3] = k0[0];
        res->end[4] = k0[1];
        res->end[5] = k0[2];
        return;
    }
    if (p->method == BHGO_METHOD_RK4)
        trace_rk4(p, &rc, x0, k0, res, sm);
    else
        trace_dp54(p, &rc, x0, k0, res, sm);
}

/* ---------------------------------------------------------------------------------------
 * Public entry points (C ABI, called through ctypes from oracle/oracle.py)
 * ------------------------------------------------------------------------------------- */
int bhgo_abi_version(void) { return 1; }

int bhgo_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* k0: [n][3]; x0: [3] if x0_shared else [n][3]; end: [n][6]; any of flags/n_attempted/
   n_accepted/t_end may be NULL.  n_threads <= 0 -> OpenMP default. */
int bhgo_trace(const bhgo_params *p, const double *x0, int x0_shared, const double *k0, size_t n,
               double *end, uint8_t *flags, uint32_t *n_attempted, uint32_t *n_accepted,
               double *t_end, int n_threads)
{
    if (!p || !x0 || !k0 || !end) return -1;
#ifdef _OPENMP
    if (n_threads <= 0) n_threads = omp_get_max_threads();
#else
    n_threads = 1;
#endif
    long long nn = (long long)n;
#pragma omp parallel for schedule(dynamic, 256) num_threads(n_threads)
    for (long long i = 0; i < nn; i++) {
        ray_result r;
        const double *xi = x0_shared ? x0 : x0 + 3 * i;
        trace_one(p, xi, k0 + 3 * i, &r, NULL);
        memcpy(end + 6 * i, r.end, sizeof(double) * 6);
        if (flags) flags[i] = (uint8_t)r.flags;
        if (n_attempted) n_attempted[i] = r.n_attempted;
        if (n_accepted) n_accepted[i] = r.n_accepted;
        if (t_end) t_end[i] = r.t_end;
    }
    return 0;
}

/* same as bhgo_trace, plus the sphere index of rays that ended on an object (-1 otherwise) */
int bhgo_trace_objects(const bhgo_params *p, const double *x0, int x0_shared, const double *k0, size_t n,
                       double *end, uint8_t *flags, uint32_t *n_attempted, uint32_t *n_accepted, int8_t *object_id,
                       int n_threads)
{
    if (!p || !x0 || !k0 || !end) return -1;
#ifdef _OPENMP
    if (n_threads <= 0) n_threads = omp_get_max_threads();
#else
    n_threads = 1;
#endif
    long long nn = (long long)n;
#pragma omp parallel for schedule(dynamic, 256) num_threads(n_threads)
    for (long long i = 0; i < nn; i++) {
        ray_result r;
        r.object_id = -1;
        trace_one(p, x0_shared ? x0 : x0 + 3 * i, k0 + 3 * i, &r, NULL);
        memcpy(end + 6 * i, r.end, sizeof(double) * 6);
        if (flags) flags[i] = (uint8_t)r.flags;
        if (n_attempted) n_attempted[i] = r.n_attempted;
        if (n_accepted) n_accepted[i] = r.n_accepted;
        if (object_id) object_id[i] = (r.flags == BHGO_FLAG_HIT_OBJECT) ? (int8_t)r.object_id : (int8_t)-1;
    }
    return 0;
}

/* Sampled trajectories: traj [n][6][T], n_valid [n] = samples emitted per ray (t_eval points beyond the ray's end are
   not produced, as with solve_ivp's t_eval).  DP5(4): the step's dense output; fixed-step RK4: its cubic Hermite interpolant */
int bhgo_trajectory(const bhgo_params *p, const double *x0, int x0_shared, const double *k0, size_t n, uint32_t T,
                    double *traj, uint32_t *n_valid, uint8_t *flags)
{
    if (!p || !x0 || !k0 || !traj || !n_valid || T < 2) return -1;
    for (size_t i = 0; i < n; i++) {
        ray_result r;
        sampler_t sm = {T, 0, p->lambda_end, traj + i * 6 * (size_t)T};
        trace_one(p, x0_shared ? x0 : x0 + 3 * i, k0 + 3 * i, &r, &sm);
        n_valid[i] = sm.next;
        if (flags) flags[i] = (uint8_t)r.flags;
    }
    return 0;
}

/* RHS probe for tests: acc[n][3] from x[n][3], k[n][3] */
int bhgo_acceleration(const bhgo_params *p, const double *x, const double *k, size_t n, double *acc)
{
    for (size_t i = 0; i < n; i++) {
        if (p->rhs_form == BHGO_RHS_KERR_BL) {
            /* x = (r, theta, phi), k = d/dlambda of those; E, L from the null condition AT THIS POINT (as trace_one
               fixes them at the camera), then the generated Boyer-Lindquist right-hand side; acc in (r, theta, phi) */
            rayctx rc;
            memset(&rc, 0, sizeof(rc));
            const double M = 0.5 * p->r_s, a = p->spin;
            const double *q = x + 3 * i, *u = k + 3 * i;
            rc.kerr = 1;
            rc.M = M;
            rc.a = a;
            double r = q[0], th = q[1], s2 = sin(th) * sin(th), c2 = cos(th) * cos(th);
            double Sig = r * r + a * a * c2, Del = r * r - 2 * M * r + a * a;
            double gtt = -(1 - 2 * M * r / Sig), gtp = -2 * M * a * r * s2 / Sig, grr = Sig / Del, gthth = Sig;
            double gpp = (r * r + a * a + 2 * M * a * a * r * s2 / Sig) * s2;
            double S = grr * u[0] * u[0] + gthth * u[1] * u[1] + gpp * u[2] * u[2] + (p->time_like ? 1.0 : 0.0);
            double B = gtp * u[2];
            double kt = (-B - sqrt(B * B - gtt * S)) / gtt;
            rc.E = -(gtt * kt + gtp * u[2]);
            rc.L = gtp * kt + gpp * u[2];
            acc_kerr_bl(&rc, q, u, acc + 3 * i);
        } else if (p->rhs_form == BHGO_RHS_REDUCED)
            acc_reduced(x + 3 * i, k + 3 * i, p->r_s, p->time_like ? 1.0 : 0.0, acc + 3 * i);
        else
            acc_christoffel(x + 3 * i, k + 3 * i, p->r_s, p->time_like ? 1.0 : 0.0, acc + 3 * i);
    }
    return 0;
}
