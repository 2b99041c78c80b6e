"""K-factored's launch time against the number of (c, e) pairs -- workgroups -- of a C3-shaped grid: 256 CUs take one
workgroup each, so n pairs are n / 256 rounds; the slope is a round, the intercept what a launch costs beside its rounds."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import load_hist, workload
from covest_amd import DenseGrid, RepeatsModel
kind, hname, axes = workload("c3", 1)
m = RepeatsModel(21, 100, load_hist(hname), 0, max_error=8)
for n_c in (4, 8, 12, 16, 24, 32, 48, 64):
    ax = [np.linspace(axes[0][0], axes[0][-1], n_c)] + list(axes[1:])
    g = DenseGrid(m, ax)
    for _ in range(30):
        g.evaluate()
    g.argmin()
    g.profile(True)
    for _ in range(40):
        g.evaluate()
    g.argmin()
    ms, n = g.kernel_ms()
    print("pairs %5d  rounds %5.2f  kernel bracket %.4f ms  per round %.4f" % (n_c * len(ax[1]), n_c * len(ax[1]) / 256.0, ms / n,
                                                                          ms / n / (n_c * len(ax[1]) / 256.0)))
    g.close()
