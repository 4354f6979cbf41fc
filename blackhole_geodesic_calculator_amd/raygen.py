"""Camera-ray generation with the reference's multisample jitter (host side, numpy).

Restates raytracer/RelativisticRenderEngine.py:185-189 (setup) and :195-230 (loops):
    x_render = fov_x * (x - int(W/2)) / W
    y_render = fov_y * (y - int(H/2)) / H * (H/W)
    d = (x_render + (1/W)(u1 - 1/2),  y_render + ((H/W)/H)(u2 - 1/2),  -1)
    d.rotate(camera euler);  d = d.normalized()
with u1, u2 successive random.random() draws after random.seed(sampling_seed), loop order
sample -> row -> column, and draws consumed only inside the mark window (:199, :219).

The MT19937 stream is produced in bulk by numpy from Python's own seeded state, so it is
bit-identical to the reference's per-pixel random.random() calls.
"""
from __future__ import annotations

import math
import random

import numpy as np


def python_random_stream(seed, n: int) -> np.ndarray:
    """The first n values of random.random() after random.seed(seed), as a float64 array."""
    st = random.Random(seed).getstate()  # (version, 624 words + pos, gauss_next)
    bg = np.random.MT19937()
    bg.state = {"bit_generator": "MT19937",
                "state": {"key": np.array(st[1][:-1], dtype=np.uint32), "pos": int(st[1][-1])}}
    # numpy's MT19937 double = (a>>5 * 2^26 + b>>6) / 2^53, the same construction as CPython's
    return np.random.Generator(bg).random(int(n))


def euler_xyz_matrix(euler) -> np.ndarray:
    """Rotation matrix of a Blender Euler in the default 'XYZ' order (matrix_world.to_euler(),
    RelativisticRenderEngine.py:183): rotate about X, then Y, then Z."""
    ax, ay, az = (float(e) for e in euler)
    cx, sx = math.cos(ax), math.sin(ax)
    cy, sy = math.cos(ay), math.sin(ay)
    cz, sz = math.cos(az), math.sin(az)
    rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]], dtype=np.float64)
    ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]], dtype=np.float64)
    rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]], dtype=np.float64)
    return rz @ ry @ rx


def _directions(width, height, samples, px, py, u, fov_x, fov_y, rotation_euler):
    """The formula of :224-230 for pixels (px, py) with jitter u[S, ..., 2]: the one place it is written down."""
    W, H = int(width), int(height)
    aspect = H / W
    dy = aspect / H
    dx = 1 / W
    x_render = fov_x * (px - int(W / 2)) / W
    y_render = fov_y * (py - int(H / 2)) / H * aspect
    d = np.empty(u.shape[:-1] + (3,), dtype=np.float64)
    d[..., 0] = x_render + dx * (u[..., 0] - 0.5)
    d[..., 1] = y_render + dy * (u[..., 1] - 0.5)
    d[..., 2] = -1.0
    rot = euler_xyz_matrix(rotation_euler)
    if not np.array_equal(rot, np.eye(3)):
        d = d @ rot.T
    nrm = np.sqrt(d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1] + d[..., 2] * d[..., 2])
    return d / nrm[..., None]


def camera_directions(width, height, samples, fov_x=1.0, fov_y=1.0, seed=42.0,
                      rotation_euler=(0.0, 0.0, 0.0), mark=None, stream=None) -> np.ndarray:
    """Unit ray directions [S, H, W, 3] (float64); NaN for pixels outside the mark window.

    mark = (y_min, y_max, x_min, x_max), bounds inclusive as in the reference (:199, :219).
    `stream` may supply a pre-computed jitter stream (cacheable: the engine re-seeds with the
    same seed on every render(), :189, so every frame of a static camera draws the same rays).
    """
    W, H, S = int(width), int(height), int(samples)
    y_min, y_max, x_min, x_max = mark if mark is not None else (0, H, 0, W)
    ys = np.arange(H)
    xs = np.arange(W)
    row_in = (ys >= y_min) & (ys <= y_max)
    col_in = (xs >= x_min) & (xs <= x_max)
    R, Cn = int(row_in.sum()), int(col_in.sum())
    if stream is None:
        stream = python_random_stream(seed, 2 * S * R * Cn)
    u = np.asarray(stream, dtype=np.float64)[: 2 * S * R * Cn].reshape(S, R, Cn, 2)
    d = _directions(W, H, S, xs[col_in][None, None, :], ys[row_in][None, :, None], u, fov_x, fov_y, rotation_euler)
    if mark is None:
        return d
    out = np.full((S, H, W, 3), np.nan)
    out[np.ix_(np.arange(S), np.nonzero(row_in)[0], np.nonzero(col_in)[0])] = d
    return out


def camera_directions_for_pixels(width, height, samples, pixels, fov_x=1.0, fov_y=1.0, seed=42.0,
                                 rotation_euler=(0.0, 0.0, 0.0), stream=None) -> np.ndarray:
    """Directions [S, P, 3] for a subset of pixels (flat indices y*W + x) of the full-frame
    stream: what one GPU's tile shard needs.  Bit-identical to camera_directions()[s, y, x]."""
    W, H, S = int(width), int(height), int(samples)
    pixels = np.asarray(pixels, dtype=np.int64)
    if stream is None:
        stream = python_random_stream(seed, 2 * S * W * H)
    u = np.asarray(stream, dtype=np.float64)[: 2 * S * W * H].reshape(S, H * W, 2)[:, pixels, :]
    py, px = np.divmod(pixels, W)
    return _directions(W, H, S, px[None, :], py[None, :], u, fov_x, fov_y, rotation_euler)
