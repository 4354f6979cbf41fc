#!/usr/bin/env python3
"""Generate the committed golden vectors under tests/golden/ (run in the build container).

The reference's own arithmetic (curvedpy) is not available, so these vectors come from the
scipy/sympy restatement in oracle/scipy_reference.py -- i.e. from scipy.integrate.solve_ivp
(the integrator README.md:196 names) applied to the README's ODE (README.md:198-209) with
the call-site parameters of raytracer/RelativisticRenderEngine.py:134, :293-294.  They pin
the C oracle (oracle/geodesic_oracle.c) and, through it, the HIP path.  PARITY UNPINNED with
respect to curvedpy itself (see oracle/geodesic_oracle.c header).

    python tests/golden/make_golden.py            # writes tests/golden/*.npz

Needs numpy, scipy, sympy only.  Takes about two minutes on one core.
"""
import math
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import scipy_reference as sr  # noqa: E402

CAM = np.array([1e-4, 0.0, 30.0])  # RelativisticRenderEngineCamEdition.py:216-221 pickle names


def euler_xyz_matrix(ax, ay, az):
    """mathutils Euler order 'XYZ' (to_euler() default): rotate about X, then Y, then Z."""
    cx, sx = math.cos(ax), math.sin(ax)
    cy, sy = math.cos(ay), math.sin(ay)
    cz, sz = math.cos(az), math.sin(az)
    rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return rz @ ry @ rx


def reference_ray_loop(width, height, samples, fov_x, fov_y, seed, euler=(0.0, 0.0, 0.0),
                       mark=None):
    """Pure-Python restatement of the direction loop, RelativisticRenderEngine.py:185-230.

    Same statement order, same `random` calls, loop order sample -> row -> column, draws only
    inside the mark window (:199, :219).  Returns directions[S, H, W, 3] (NaN where skipped).
    """
    aspectratio = height / width
    dy = aspectratio / height
    dx = 1 / width
    random.seed(seed)
    rot = euler_xyz_matrix(*euler)
    y_min, y_max, x_min, x_max = mark if mark else (0, height, 0, width)
    out = np.full((samples, height, width, 3), np.nan)
    for s in range(samples):
        for y in range(height):
            if y >= y_min and y <= y_max:
                for x in range(width):
                    if x >= x_min and x <= x_max:
                        aspectratio = height / width
                        x_render = fov_x * (x - int(width / 2)) / width
                        y_render = fov_y * (y - int(height / 2)) / height * aspectratio
                        d = np.array([x_render + dx * (random.random() - 0.5),
                                      y_render + dy * (random.random() - 0.5), -1.0])
                        d = rot @ d
                        d = d / math.sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2])
                        out[s, y, x] = d
    return out


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"wrote {path}  ({os.path.getsize(path)/1024:.1f} KiB)")


def pack(res):
    return dict(end=res["end"], flags=res["flags"], n_attempted=res["n_attempted"],
                n_accepted=res["n_accepted"], t_end=res["t_end"])


def main():
    # ---- 1. config 1: 64x64x1 frame, default controller -------------------------------
    dirs = reference_ray_loop(64, 64, 1, 0.6, 0.6, 42.0)
    k0 = dirs.reshape(-1, 3)
    res = sr.trace_rays(k0, CAM, r_s=1.0, lambda_end=50.0, form="christoffel")
    save("frame64_christoffel", k0=k0, x0=CAM, r_s=1.0, lambda_end=50.0, max_step=np.inf,
         rtol=1e-3, atol=1e-6, **pack(res))
    res = sr.trace_rays(k0, CAM, r_s=1.0, lambda_end=50.0, form="reduced")
    save("frame64_reduced", k0=k0, x0=CAM, r_s=1.0, lambda_end=50.0, max_step=np.inf,
         rtol=1e-3, atol=1e-6, **pack(res))

    # ---- 2. generic sympy contraction on a strided subset -----------------------------
    sub = k0[::16]
    res = sr.trace_rays(sub, CAM, r_s=1.0, lambda_end=50.0, form="sympy")
    save("frame64_sympy_subset", k0=sub, x0=CAM, r_s=1.0, lambda_end=50.0, max_step=np.inf,
         rtol=1e-3, atol=1e-6, **pack(res))

    # ---- 3. ray generation (a7): small frames incl. rotation, non-square, mark window --
    save("raygen",
         d_8x6x2=reference_ray_loop(8, 6, 2, 0.6, 0.45, 42.0),
         d_5x7x3_rot=reference_ray_loop(5, 7, 3, 1.0, 1.0, 7.0, euler=(0.3, -0.2, 1.1)),
         d_6x6x2_mark=reference_ray_loop(6, 6, 2, 0.6, 0.6, 42.0, mark=(1, 4, 2, 3)),
         first_draws=np.array([(random.seed(42.0), [random.random() for _ in range(8)])[1]][0]))

    # ---- 4. fine regime: max_step = 0.1 (CamEdition.py:216 pickle name) ---------------
    fine = k0[:: 64 * 4 + 3][:24]
    res = sr.trace_rays(fine, CAM, r_s=1.0, lambda_end=50.0, max_step=0.1, form="christoffel")
    save("fine_maxstep01", k0=fine, x0=CAM, r_s=1.0, lambda_end=50.0, max_step=0.1,
         rtol=1e-3, atol=1e-6, **pack(res))

    # ---- 5. Fig. 5 geometry (README.md:68-70): x0 = -15 r_s, y0 = 3..19, k = (1,0,0) ---
    y0s = np.arange(3.0, 20.0)
    x0s = np.stack([-15.0 * np.ones_like(y0s), y0s, np.zeros_like(y0s)], 1)
    kf = np.tile(np.array([1.0, 0.0, 0.0]), (len(y0s), 1))
    res = sr.trace_rays(kf, x0s, r_s=1.0, lambda_end=60.0, form="christoffel")
    defl = []
    for i in range(len(y0s)):
        c = sr.trace_ray(kf[i], x0s[i], r_s=1.0, lambda_end=4000.0, rtol=1e-12, atol=1e-14,
                         form="reduced", method="DOP853")
        kx, ky = c["end"][3], c["end"][4]
        defl.append(math.degrees(math.atan2(-ky, kx)))
    save("fig5", k0=kf, x0=x0s, r_s=1.0, lambda_end=60.0, max_step=np.inf, rtol=1e-3, atol=1e-6,
         deflection_deg_converged=np.array(defl), **pack(res))

    # ---- 6. capture threshold b_c = 3 sqrt(3)/2 r_s -----------------------------------
    bs = np.array([2.40, 2.55, 2.59, 2.60, 2.61, 2.65, 2.80])
    xs = np.stack([-40.0 * np.ones_like(bs), bs, np.zeros_like(bs)], 1)
    kc = np.tile(np.array([1.0, 0.0, 0.0]), (len(bs), 1))
    hit = []
    for i in range(len(bs)):
        c = sr.trace_ray(kc[i], xs[i], r_s=1.0, lambda_end=200.0, rtol=1e-11, atol=1e-13,
                         form="reduced", method="DOP853")
        hit.append(c["flags"] & 1)
    res = sr.trace_rays(kc, xs, r_s=1.0, lambda_end=200.0, rtol=1e-8, atol=1e-10, form="christoffel")
    save("capture", k0=kc, x0=xs, r_s=1.0, lambda_end=200.0, max_step=np.inf, rtol=1e-8,
         atol=1e-10, hit_converged=np.array(hit, np.uint8), **pack(res))

    # ---- 7. sphere exit (Limited engine: ratio 30, curve_end heuristic :277-279) -------
    rng = np.random.default_rng(5)
    n = 48
    pos = rng.normal(size=(n, 3))
    pos = 30.0 * pos / np.linalg.norm(pos, axis=1)[:, None]
    aim = rng.normal(size=(n, 3)) * 4.0  # aim near the hole
    dirs_in = aim - pos
    dirs_in /= np.linalg.norm(dirs_in, axis=1)[:, None]
    lam = 50 + 2 * 50 * (30 / 20 - 1)
    res = sr.trace_rays(dirs_in, pos, r_s=1.0, lambda_end=lam, r_exit=30.0, form="christoffel")
    save("sphere_exit", k0=dirs_in, x0=pos, r_s=1.0, lambda_end=lam, max_step=np.inf, rtol=1e-3,
         atol=1e-6, r_exit=30.0, **pack(res))

    # ---- 8. other masses / off-centre hole (bh_loc subtraction, :278) ------------------
    k8 = k0[5::97][:40]
    res = sr.trace_rays(k8, np.array([3.0, -2.0, 45.0]), r_s=2.5, lambda_end=80.0, form="christoffel")
    save("mass_1p25", k0=k8, x0=np.array([3.0, -2.0, 45.0]), r_s=2.5, lambda_end=80.0,
         max_step=np.inf, rtol=1e-3, atol=1e-6, **pack(res))


def main_disk():
    # ---- 9. thin disk in z = 0 (LimitedRelativisticRenderEngine.py:283-286, :413-438): config-3 like
    # geometry, camera at r = 30 seen from five inclinations, annulus 4.5 .. 10.5 r_s (0.15 .. 0.35 x ratio 30)
    rng = np.random.default_rng(9)
    ks, xs = [], []
    for inc_deg in (85.0, 80.0, 60.0, 30.0, 5.0):
        inc = math.radians(inc_deg)
        cam = np.array([30 * math.sin(inc), 0.0, 30 * math.cos(inc)])
        aim = rng.normal(size=(40, 3)) * np.array([9.0, 9.0, 1.0])
        d = aim - cam
        ks.append(d / np.linalg.norm(d, axis=1)[:, None])
        xs.append(np.tile(cam, (40, 1)))
    k0, x0 = np.concatenate(ks), np.concatenate(xs)
    end, flags, nacc, tend = [], [], [], []
    for i in range(len(k0)):
        r = sr.trace_ray(k0[i], x0[i], r_s=1.0, lambda_end=80.0, form="christoffel", disk=(4.5, 10.5))
        end.append(r["end"]); flags.append(r["flags"]); nacc.append(r["n_accepted"]); tend.append(r["t_end"])
    save("disk", k0=k0, x0=x0, r_s=1.0, lambda_end=80.0, max_step=np.inf, rtol=1e-3, atol=1e-6,
         disk_r_in=4.5, disk_r_out=10.5, end=np.array(end), flags=np.array(flags, np.uint8),
         n_accepted=np.array(nacc, np.uint32), t_end=np.array(tend))


def main_objects():
    # ---- 11. object spheres inside the curved region (the reference's collision stub, RelativisticRenderEngine.py:
    # 304-305): scipy terminal events g_j = |x - c_j| - rho_j (direction -1) next to horizon, exit sphere and disk.
    # max_step = 0.25 is small against every radius, so plain sign-change detection sees every entry.
    rng = np.random.default_rng(11)
    n = 240
    k0 = np.stack([rng.uniform(-0.3, 0.3, n), rng.uniform(-0.3, 0.3, n), -np.ones(n)], 1)
    k0 /= np.linalg.norm(k0, axis=1)[:, None]
    sph = np.array([(2.0, 1.0, 8.0, 1.5), (-3.0, 0.5, -1.0, 1.2), (1.0, -4.0, -10.0, 2.0), (0.0, 3.5, 2.0, 0.8),
                    (4.5, 0.0, 0.0, 1.0)])
    end, flags, natt, nacc, tend, obj = [], [], [], [], [], []
    for i in range(n):
        r = sr.trace_ray(k0[i], CAM, r_s=1.0, lambda_end=70.0, max_step=0.25, form="christoffel", r_exit=35.0,
                         disk=(3.0, 7.0), spheres=sph)
        end.append(r["end"]); flags.append(r["flags"]); natt.append(r["n_attempted"]); nacc.append(r["n_accepted"])
        tend.append(r["t_end"]); obj.append(r.get("object_id", -1) if r["flags"] == sr.FLAG_HIT_OBJECT else -1)
        natt[-1] = max(natt[-1], 0)  # disk hits: scipy integrates on past the disk, attempted count not comparable (stored 0)
    save("objects", k0=k0, x0=CAM, r_s=1.0, lambda_end=70.0, max_step=0.25, rtol=1e-3, atol=1e-6, r_exit=35.0,
         disk_r_in=3.0, disk_r_out=7.0, spheres=sph, end=np.array(end), flags=np.array(flags, np.uint8),
         n_attempted=np.array(natt, np.uint32), n_accepted=np.array(nacc, np.uint32), t_end=np.array(tend),
         object_id=np.array(obj, np.int8))


def main_kerr():
    # ---- 10. Kerr a/M = 0.9 (CamEdition.py:210), mass 0.5, Boyer-Lindquist Christoffels (config 5) ----
    M, a = 0.5, 0.45
    rng = np.random.default_rng(10)
    sets = []
    # (i) the reference's near-axis camera (x = 1e-4, CamEdition.py:216-221), (ii) an off-axis camera
    cam1 = CAM
    k1 = np.stack([rng.uniform(-0.3, 0.3, 48), rng.uniform(-0.3, 0.3, 48), -np.ones(48)], 1)
    cam2 = np.array([0.0, -25.0, 12.0])
    k2 = (-cam2 / np.linalg.norm(cam2))[None, :] + rng.normal(size=(96, 3)) * 0.08
    for cam, k in ((cam1, k1), (cam2, k2)):
        k = k / np.linalg.norm(k, axis=1)[:, None]
        for i in range(len(k)):
            r = sr.trace_ray_kerr(k[i], cam, M, a, lambda_end=60.0)
            sets.append((k[i], cam, r["end"], r["flags"], r["n_attempted"], r["n_accepted"], r["t_end"]))
    save("kerr_a09", k0=np.array([s[0] for s in sets]), x0=np.array([s[1] for s in sets]), r_s=1.0, spin=a,
         lambda_end=60.0, max_step=np.inf, rtol=1e-3, atol=1e-6, end=np.array([s[2] for s in sets]),
         flags=np.array([s[3] for s in sets], np.uint8), n_attempted=np.array([s[4] for s in sets], np.uint32),
         n_accepted=np.array([s[5] for s in sets], np.uint32), t_end=np.array([s[6] for s in sets]))


def main_kerr_disk():
    # ---- 12. Kerr a/M = 0.9 with a thin disk in the equatorial plane (config 3 x config 5): scipy event
    # g = cos(theta) on the Boyer-Lindquist solve, annulus in the cylindrical radius; three inclinations ----
    M, a = 0.5, 0.45
    rng = np.random.default_rng(12)
    ks, xs = [], []
    for inc_deg in (80.0, 60.0, 20.0):
        inc = math.radians(inc_deg)
        cam = np.array([30 * math.sin(inc), 0.5, 30 * math.cos(inc)])
        aim = rng.normal(size=(40, 3)) * np.array([8.0, 8.0, 1.0])
        d = aim - cam
        ks.append(d / np.linalg.norm(d, axis=1)[:, None])
        xs.append(np.tile(cam, (40, 1)))
    k0, x0 = np.concatenate(ks), np.concatenate(xs)
    res = [sr.trace_ray_kerr(k0[i], x0[i], M, a, lambda_end=80.0, disk=(3.0, 10.0)) for i in range(len(k0))]
    save("kerr_disk", k0=k0, x0=x0, r_s=1.0, spin=a, lambda_end=80.0, max_step=np.inf, rtol=1e-3, atol=1e-6,
         disk_r_in=3.0, disk_r_out=10.0, end=np.array([r["end"] for r in res]),
         flags=np.array([r["flags"] for r in res], np.uint8),
         n_accepted=np.array([r["n_accepted"] for r in res], np.uint32), t_end=np.array([r["t_end"] for r in res]))


def main_kerr_objects():
    # ---- 14. object spheres in Kerr (a/M = 0.9): met in the Cartesian frame by the Boyer-Lindquist solve; max_step 0.5 as for
    # the Schwarzschild object set (scipy only sees sign changes between step ends) ----
    M, a = 0.5, 0.45
    spheres = np.array([[6.0, 0.0, 0.0, 1.5], [0.0, -5.0, 1.0, 1.0], [-3.0, 4.0, -2.0, 1.2]])
    cam = np.array([2.0, -24.0, 14.0])
    rng = np.random.default_rng(14)
    aim = rng.normal(size=(60, 3)) * np.array([5.0, 5.0, 3.0])
    aim[:30] = spheres[rng.integers(0, 3, 30), :3] + rng.normal(size=(30, 3)) * 1.0     # half of them at the spheres
    k0 = aim - cam
    k0 /= np.linalg.norm(k0, axis=1)[:, None]
    res = [sr.trace_ray_kerr(k0[i], cam, M, a, lambda_end=60.0, max_step=0.5, spheres=spheres) for i in range(len(k0))]
    save("kerr_objects", k0=k0, x0=np.tile(cam, (len(k0), 1)), r_s=1.0, spin=a, lambda_end=60.0, max_step=0.5, rtol=1e-3, atol=1e-6,
         spheres=spheres, end=np.array([r["end"] for r in res]), flags=np.array([r["flags"] for r in res], np.uint8),
         object_id=np.array([r.get("object_id", -1) for r in res], np.int8),
         n_attempted=np.array([r["n_attempted"] for r in res], np.uint32),
         n_accepted=np.array([r["n_accepted"] for r in res], np.uint32), t_end=np.array([r["t_end"] for r in res]))


def main_timelike():
    # ---- 13. time_like=True (the solver object's other constructor value, RelativisticRenderEngine.py:134): massive
    # particles, g(k, k) = -1, parameter = proper time.  Orbits from four radii with a circular-orbit speed scaled by
    # 0 (radial plunge) ... 1.6 (unbound), tilted out of the equatorial plane, some with a radial component; once through
    # scipy on the Cartesian Christoffel form, once through the Boyer-Lindquist solve with a/M = 0.9 ----
    M, a = 0.5, 0.45
    ks, xs = [], []
    rng = np.random.default_rng(13)
    for r0 in (3.2, 4.0, 6.0, 10.0):
        v_circ = math.sqrt(M / (r0 - 3 * M))                 # r dphi/dtau of the circular orbit at r0 (needs r0 > 3M)
        for scale in (0.0, 0.6, 1.0, 1.25, 1.6):
            ang = rng.uniform(0.0, 2 * math.pi)
            inc = rng.uniform(0.2, 1.2)
            pos = r0 * np.array([math.cos(ang) * math.sin(inc), math.sin(ang) * math.sin(inc), math.cos(inc)])
            e_r = pos / r0
            e_t = np.cross(e_r, np.array([0.3, -0.5, 0.8]))
            e_t /= np.linalg.norm(e_t)
            vr = rng.choice([0.0, 0.0, -0.15, 0.1])
            ks.append(scale * v_circ * e_t + vr * e_r)
            xs.append(pos)
    k0, x0 = np.array(ks), np.array(xs)
    rs = [sr.trace_ray(k0[i], x0[i], r_s=1.0, lambda_end=150.0, time_like=True) for i in range(len(k0))]
    rk = [sr.trace_ray_kerr(k0[i], x0[i], M, a, lambda_end=150.0, time_like=True) for i in range(len(k0))]
    save("timelike", k0=k0, x0=x0, r_s=1.0, spin=a, lambda_end=150.0, max_step=np.inf, rtol=1e-3, atol=1e-6,
         end=np.array([r["end"] for r in rs]), flags=np.array([r["flags"] for r in rs], np.uint8),
         n_attempted=np.array([r["n_attempted"] for r in rs], np.uint32), n_accepted=np.array([r["n_accepted"] for r in rs], np.uint32),
         t_end=np.array([r["t_end"] for r in rs]),
         kerr_end=np.array([r["end"] for r in rk]), kerr_flags=np.array([r["flags"] for r in rk], np.uint8),
         kerr_n_attempted=np.array([r["n_attempted"] for r in rk], np.uint32),
         kerr_n_accepted=np.array([r["n_accepted"] for r in rk], np.uint32), kerr_t_end=np.array([r["t_end"] for r in rk]))


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "timelike"):
        main_timelike()
    if which in ("all", "kerr_objects"):
        main_kerr_objects()
    if which in ("all",):
        main()
    if which in ("all", "disk"):
        main_disk()
    if which in ("all", "kerr"):
        main_kerr()
    if which in ("all", "objects"):
        main_objects()
    if which in ("all", "kerr_disk"):
        main_kerr_disk()
