"""Latency of one likelihood evaluation through the C ABI (what scipy's refinement waits for)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import load_hist
from covest_amd import BasicModel, RepeatsModel
for name, cls, p in (("basic H10k_basic", BasicModel, (4000.0, 0.02)), ("repeats H10k_rep", RepeatsModel, (25.0, 0.02, 0.6, 0.5, 0.1)),
                     ("repeats sim", RepeatsModel, (10.0, 0.05, 0.8, 0.5, 0.3))):
    hist = load_hist("sim_c10_e0.05" if name.endswith("sim") else name.split()[1])
    m = cls(21, 100, hist, 0, max_error=8)
    m.compute_loglikelihood(*p)
    for n in (1, 6, 64):
        pts = np.tile(np.array(p), (n, 1)) * (1 + 1e-3 * np.arange(n))[:, None]
        m.loglikelihood_points(pts)
        t0 = time.perf_counter()
        reps = 200
        for _ in range(reps):
            m.loglikelihood_points(pts)
        dt = (time.perf_counter() - t0) / reps
        print("%-18s n=%3d  %7.1f us per call  %7.1f us per point" % (name, n, 1e6 * dt, 1e6 * dt / n))
