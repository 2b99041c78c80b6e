"""Kernel time of the dense-grid kernels with and without a histogram tail (the TAIL template variants)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import load_hist, workload
from covest_amd import BasicModel, DenseGrid, RepeatsModel
for w in ("c3", "c2"):
    kind, hname, axes = workload(w, 1)
    cls = RepeatsModel if kind == "repeats" else BasicModel
    for tail in (0, 12345):
        m = cls(21, 100, load_hist(hname), tail, max_error=8)
        g = DenseGrid(m, axes)
        g.evaluate(); g.argmin()
        g.profile(True)
        for _ in range(20):
            g.evaluate()
        g.argmin()
        ms, n = g.kernel_ms()
        print(w, "tail", tail, "kernel", g.work()[2], "%.4f ms" % (ms / n), "bins evaluated", m.bins_evaluated)
