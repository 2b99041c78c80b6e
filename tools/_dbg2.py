import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from bench import load_hist, workload
from covest_amd import BasicModel, DenseGrid, RepeatsModel
for w in ("c3", "c2"):
    kind, hname, axes = workload(w, 1)
    cls = RepeatsModel if kind == "repeats" else BasicModel
    m = cls(21, 100, load_hist(hname), 0, max_error=8)
    g = DenseGrid(m, axes)
    for _ in range(30):
        g.evaluate()
    g.argmin()
    print(w, "done")
