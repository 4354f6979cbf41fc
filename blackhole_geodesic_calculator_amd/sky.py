"""A deterministic synthetic sky image (there is no dataset: "data": "synthetic"); numpy only."""
import numpy as np


def synthetic_sky(width=2048, height=1024, seed=7):
    """A deterministic equirectangular test sky: smooth gradient + a few hundred gaussian stars."""
    rng = np.random.default_rng(seed)
    v, u = np.meshgrid(np.linspace(0, 1, height), np.linspace(0, 1, width), indexing="ij")
    img = np.stack([0.05 + 0.1 * u, 0.05 + 0.1 * v, 0.1 + 0.1 * np.sin(2 * np.pi * u) ** 2], -1)
    for _ in range(300):
        cx, cy, amp, sig = rng.uniform(0, width), rng.uniform(0, height), rng.uniform(0.3, 1.0), rng.uniform(1.0, 3.0)
        x0, x1 = int(max(0, cx - 4 * sig)), int(min(width, cx + 4 * sig + 1))
        y0, y1 = int(max(0, cy - 4 * sig)), int(min(height, cy + 4 * sig + 1))
        yy, xx = np.mgrid[y0:y1, x0:x1]
        img[y0:y1, x0:x1] += (amp * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * sig * sig)))[..., None]
    rgba = np.concatenate([img, np.ones((height, width, 1))], -1)
    return rgba.astype(np.float32)
