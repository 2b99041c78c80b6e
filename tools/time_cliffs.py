"""The two shapes that fell back to K-direct in round 1, timed on both kernels (hipEvents around the launches,
DenseGrid.profile): (a) a dense repeats grid whose threshold_o - 1 exceeds a workgroup's 512 lanes, (b) max_error
above 8 (the reference's default for k = 21 is 22 error classes).  Prints kernel ms and the largest relative
difference between the two kernels' values."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import load_hist  # noqa: E402
from covest_amd import BasicModel, DenseGrid, RepeatsModel  # noqa: E402


def timed(model, axes, kernel, reps):
    g = DenseGrid(model, axes)
    g.evaluate(kernel=kernel)
    g.argmin()
    g.profile(True)
    for _ in range(reps):
        g.evaluate(kernel=kernel)
    g.argmin()
    ms, launches = g.kernel_ms()
    ll = g.loglikelihoods()
    name = g.work()[2]
    g.close()
    return ms / reps, name, ll


cases = [
    ("repeats, H10k_rep, max_error 8, q axis 0.004..0.03 (threshold_o up to ~3000): c8 x e8 x 8 x 1 x 8",
     RepeatsModel, "H10k_rep", dict(max_error=8),
     [np.linspace(15, 30, 8), np.linspace(0.005, 0.08, 8), np.linspace(0.3, 0.95, 8), np.array([0.5]),
      np.linspace(0.004, 0.03, 8)]),
    ("repeats, H10k_rep, max_error 22 (every error class of k = 21): c16 x e16 x 8 x 1 x 8",
     RepeatsModel, "H10k_rep", dict(max_error=22),
     [np.linspace(15, 30, 16), np.linspace(0.005, 0.08, 16), np.linspace(0.3, 0.95, 8), np.array([0.5]),
      np.linspace(0.05, 0.95, 8)]),
    ("basic, H10k_basic, max_error 22: c200 x e200",
     BasicModel, "H10k_basic", dict(max_error=22),
     [np.linspace(2000.0, 6000.0, 200), np.linspace(0.001, 0.1, 200)]),
]
for label, cls, hname, kw, axes in cases:
    m = cls(21, 100, load_hist(hname), 0, **kw)
    n = int(np.prod([len(a) for a in axes]))
    fast_ms, fast_name, fast_ll = timed(m, axes, "auto", 5)
    slow_ms, slow_name, slow_ll = timed(m, axes, "direct", 1)
    both = np.isfinite(fast_ll) & np.isfinite(slow_ll)
    same_special = bool(np.all((fast_ll == slow_ll) | both | (np.isnan(fast_ll) & np.isnan(slow_ll))))
    rel = float(np.max(np.abs(fast_ll[both] - slow_ll[both]) / np.maximum(1.0, np.abs(slow_ll[both])))) if both.any() else 0.0
    label += " [%d finite, -inf/NaN identical: %s]" % (int(both.sum()), same_special)
    print("%s\n   %d points: %s %.3f ms, %s %.3f ms (x%.1f); largest relative difference %.2e" % (
        label, n, fast_name, fast_ms, slow_name, slow_ms, slow_ms / fast_ms, rel), flush=True)
    m.close()
