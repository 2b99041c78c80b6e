"""Where the warm time-to-argmin goes: handle creation, plan building, launch, read-back."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import load_hist, workload
from covest_amd import BasicModel, DenseGrid, RepeatsModel
for w in ("c3", "c2"):
    kind, hname, axes = workload(w, 1)
    cls = RepeatsModel if kind == "repeats" else BasicModel
    hist = load_hist(hname)
    m0 = cls(21, 100, hist, 0, max_error=8); g0 = DenseGrid(m0, axes); g0.evaluate(); g0.argmin()
    T = {k: [] for k in ("model", "handle", "grid", "evaluate", "argmin")}
    for _ in range(10):
        t0 = time.perf_counter(); m = cls(21, 100, hist, 0, max_error=8); t1 = time.perf_counter()
        m.handle; t2 = time.perf_counter()
        g = DenseGrid(m, axes); t3 = time.perf_counter()
        g.evaluate(); t4 = time.perf_counter()
        g.argmin(); t5 = time.perf_counter()
        for k, a, b in (("model", t0, t1), ("handle", t1, t2), ("grid", t2, t3), ("evaluate", t3, t4), ("argmin", t4, t5)):
            T[k].append(1e3 * (b - a))
        g.close(); m.close()
    print(w, {k: round(float(np.median(v)), 3) for k, v in T.items()}, "total ms", round(sum(float(np.median(v)) for v in T.values()), 3))
