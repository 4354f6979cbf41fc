"""Latency of the literal per-ray drop-in call (RelativisticRenderEngine.py:293-294) -- dev aid."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from blackhole_geodesic_calculator_amd import GeodesicIntegratorSchwarzschild
gi = GeodesicIntegratorSchwarzschild(mass=0.5, time_like=False, verbose=False)
rng = np.random.default_rng(0)
x0 = np.array([1e-4, 0.0, 30.0])
K = np.array([0, 0, -1.0]) + rng.normal(size=(2000, 3)) * 0.15
for npts in (50, 1000, 10000):
    for rep in range(2):
        t0 = time.perf_counter()
        for i in range(500):
            k, x, res = gi.calc_trajectory(K[i], x0, max_step=1e4, curve_end=50, nr_points_curve=npts, verbose=False)
        dt = (time.perf_counter() - t0) / 500
    print("nr_points_curve %5d: %.1f us per call (%d samples kept in the last one)" % (npts, dt * 1e6, x.shape[1]))
