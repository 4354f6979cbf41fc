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
