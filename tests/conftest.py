import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
CAM = np.array([1e-4, 0.0, 30.0])  # camera of BASELINE.json configs 1/2


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: g[k] for k in g.files}


def golden_kwargs(g, rhs_form=0):
    kw = dict(r_s=float(g["r_s"]), lambda_end=float(g["lambda_end"]), max_step=float(g["max_step"]),
              rtol=float(g["rtol"]), atol=float(g["atol"]), rhs_form=rhs_form)
    if "r_exit" in g:
        kw["r_exit"] = float(g["r_exit"])
    return kw


GOLDEN_TRACE_SETS = ["frame64_christoffel", "frame64_reduced", "frame64_sympy_subset", "fine_maxstep01",
                     "fig5", "capture", "sphere_exit", "mass_1p25"]


def frame_rays(n, seed=0, fov=0.6):
    """Seeded camera-like unit directions looking down -z with the given field of view."""
    rng = np.random.default_rng(seed)
    k = np.stack([rng.uniform(-fov / 2, fov / 2, n), rng.uniform(-fov / 2, fov / 2, n), -np.ones(n)], 1)
    return k / np.linalg.norm(k, axis=1)[:, None]


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as oc
    oc.build()
    return oc


@pytest.fixture(scope="session")
def ctx():
    """GPU context through the C ABI.  Fails loudly (no skip, no fallback) if the HIP library is
    missing or no device is usable."""
    from blackhole_geodesic_calculator_amd import _ffi
    c = _ffi.Context(0)
    yield c
    c.close()
