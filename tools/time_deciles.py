"""K-basic's launch time on the ten deciles of C2's c axis (1000 x 1000 grid: 100 000 points a decile) -- where along c
the launch's time goes: the low deciles are the doomed rows (every sum -inf, the waves leave at once)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import load_hist, workload
from covest_amd import BasicModel, DenseGrid
kind, hname, axes = workload("c2", 1)
m = BasicModel(21, 100, load_hist(hname), 0, max_error=8)
tot = 0.0
for d in range(10):
    ax = [np.asarray(axes[0])[100 * d:100 * (d + 1)], axes[1]]
    g = DenseGrid(m, ax)
    for _ in range(30):
        g.evaluate()
    g.argmin()
    g.profile(True)
    for _ in range(60):
        g.evaluate()
    g.argmin()
    ms, n = g.kernel_ms()
    ll = g.loglikelihoods()
    tot += ms / n
    print("decile %d  c %.3f .. %.3f  kernel bracket %.4f ms  -inf share %.3f" % (d, ax[0][0], ax[0][-1], ms / n, float(np.isneginf(ll).mean())))
    g.close()
print("sum of the deciles %.4f ms" % tot)
