"""30 evaluations of the C3 and of the C2 grid (for rocprofv3 --kernel-trace --stats: per-kernel times)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import load_hist, workload  # noqa: E402
from covest_amd import BasicModel, DenseGrid, RepeatsModel  # noqa: E402

for w in (sys.argv[1:] or ["c3", "c2"]):
    kind, hname, axes = workload(w, 1)
    cls = RepeatsModel if kind == "repeats" else BasicModel
    m = cls(21, 100, load_hist(hname), 0, max_error=8)
    g = DenseGrid(m, axes)
    for _ in range(30):
        g.evaluate()
    g.argmin()
    print(w, "done")
