"""numpy restatement of the shading + multisample mean stage -- TEST INFRASTRUCTURE ONLY.

Follows raytracer/RelativisticRenderEngine.py:242-250 (black for horizon rays, sbuf += colour,
mean over samples) and :366-378 (equirect (u, v) from the exit direction).  The texture filter is
the BUILD'S OWN bilinear definition (u wraps, v clamps, texel centres at half-integers), because
Blender's Texture.evaluate cannot run outside Blender; it pins blackhole_geodesic_calculator_amd's
device shade kernel, not Blender's filter.  PARITY UNPINNED with respect to Blender's own filter.
"""
import numpy as np


def sky_lookup(sky, u, v):
    TH, TW = sky.shape[:2]
    fx = (u + 1.0) * 0.5 * TW - 0.5
    fy = (v + 1.0) * 0.5 * TH - 0.5
    x0f, y0f = np.floor(fx), np.floor(fy)
    ax, ay = fx - x0f, fy - y0f
    x0 = x0f.astype(np.int64) % TW
    x1 = (x0f.astype(np.int64) + 1) % TW
    y0 = np.clip(y0f.astype(np.int64), 0, TH - 1)
    y1 = np.clip(y0f.astype(np.int64) + 1, 0, TH - 1)
    s = sky.astype(np.float64)
    w00, w01, w10, w11 = (1 - ax) * (1 - ay), ax * (1 - ay), (1 - ax) * ay, ax * ay
    return (w00[:, None] * s[y0, x0, :3] + w01[:, None] * s[y0, x1, :3] + w10[:, None] * s[y1, x0, :3]
            + w11[:, None] * s[y1, x1, :3])


def shade_reduce(end, flags, n_pixels, samples, sky):
    """end [S*P, 6], flags [S*P] -> rgba [P, 4] (fp64), samples accumulated in order."""
    acc = np.zeros((n_pixels, 3))
    for s in range(samples):
        e = end[s * n_pixels:(s + 1) * n_pixels]
        f = flags[s * n_pixels:(s + 1) * n_pixels]
        d = e[:, 3:6]
        d = d * (1.0 / np.sqrt(d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2]))[:, None]
        theta = 1.0 - np.arccos(d[:, 2]) / np.pi
        phi = np.arctan2(d[:, 1], d[:, 0]) / np.pi
        rgb = sky_lookup(sky, -phi, 2.0 * theta - 1.0)
        rgb[(f & 1) != 0] = 0.0
        acc += np.nan_to_num(rgb)
    return np.concatenate([acc / samples, np.ones((n_pixels, 1))], 1)


def disk_colour(end, r_in, r_out, tex=None, phase=0.0, mean=0.2, stddev=0.3, intensity=1.0):
    """LimitedRelativisticRenderEngine.py:427-436, :300 (sign(+0) taken as +, where the reference has 0/0)."""
    x, y = end[:, 0], end[:, 1]
    R = np.sqrt(x * x + y * y)
    scale = (R - r_in) / (r_out - r_in)
    inten = intensity * np.exp(-((scale - mean) ** 2) / (2.0 * stddev * stddev)) / np.sqrt(2.0 * np.pi * stddev)
    tx = (phase + np.arccos(np.clip(x / R, -1.0, 1.0)) * np.where(y < 0.0, -1.0, 1.0)) / np.pi
    rgb = np.ones((len(end), 3)) if tex is None else sky_lookup(tex, tx, scale)
    return rgb * inten[:, None]


def object_colour(end, obj, spheres, sphere_rgb, lamps):
    """RelativisticRenderEngine.py:341-363: Lambert point lamps, straight light paths, shadowed by the other
    spheres, n.l clamped at 0 (the build's choice; the reference lets it go negative)."""
    spheres = np.asarray(spheres, float).reshape(-1, 4)
    sphere_rgb = np.asarray(sphere_rgb, float).reshape(-1, 3)
    out = np.zeros((len(end), 3))
    for i in range(len(end)):
        j = int(obj[i])
        if j < 0:
            continue
        loc = end[i, 0:3]
        n = (loc - spheres[j, 0:3]) / spheres[j, 3]
        tot = 0.0
        for lx, ly, lz, li in np.asarray(lamps, float).reshape(-1, 4):
            lv = np.array([lx, ly, lz]) - loc
            d2 = lv @ lv
            dist = np.sqrt(d2)
            ld = lv / dist
            ndl = n @ ld
            if not ndl > 0.0:
                continue
            shadow = False
            for q in range(len(spheres)):
                if q == j:
                    continue
                oc = loc - spheres[q, 0:3]
                b = oc @ ld
                disc = b * b - (oc @ oc - spheres[q, 3] ** 2)
                if disc > 0.0:
                    t0, t1 = -b - np.sqrt(disc), -b + np.sqrt(disc)
                    if (1e-5 < t0 < dist) or (t0 <= 1e-5 and t1 > 1e-5):
                        shadow = True
            if not shadow:
                tot += li * li * ndl / d2
        out[i] = sphere_rgb[j] * tot
    return out


def shade_scene(end, flags, obj, n_pixels, samples, sky, disk=None, disk_tex=None, disk_profile=None, spheres=None,
                sphere_rgb=None, lamps=None):
    """shade_reduce plus the disk and object colours; same accumulation order."""
    acc = np.zeros((n_pixels, 3))
    for s in range(samples):
        sl = slice(s * n_pixels, (s + 1) * n_pixels)
        e, f = end[sl], flags[sl]
        one = shade_reduce(e, f, n_pixels, 1, sky)[:, :3]
        if disk is not None:
            m = f == 128
            if m.any():
                one[m] = disk_colour(e[m], disk[0], disk[1], disk_tex, **(disk_profile or {}))
        if spheres is not None and len(spheres):
            m = f == 0x88
            if m.any():
                rgb = np.ones((len(spheres), 3)) if sphere_rgb is None else sphere_rgb
                one[m] = object_colour(e[m], obj[sl][m], spheres, rgb, lamps if lamps is not None else [])
        acc += np.nan_to_num(one)
    return np.concatenate([acc / samples, np.ones((n_pixels, 1))], 1)
