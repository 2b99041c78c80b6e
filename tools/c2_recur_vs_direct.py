#!/usr/bin/env python3
"""GPU box: K-basic against K-direct on the whole C2 grid -- where they differ, by route (diagnostic library:
COVEST_AMD_LIB=tools/bin/libcovest_amd_diag.so) or just the differences (shipped library)."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))


def main():
    from conftest import load_hist
    from covest_amd import BasicModel, DenseGrid
    hname = sys.argv[1] if len(sys.argv) > 1 else "H10k_basic"
    tail = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
    m = BasicModel(21, 100, load_hist(hname), tail, max_error=8)
    grid = DenseGrid(m, [np.linspace(2000.0, 6000.0, 1000), np.linspace(0.001, 0.1, 1000)])
    cls = lp = None
    if "diag" in os.environ.get("COVEST_AMD_LIB", ""):
        os.environ["COVEST_DIAG_BASIC_CLASS"] = "1"
        grid.evaluate(kernel="recur")
        cls = grid.loglikelihoods()
        os.environ["COVEST_DIAG_BASIC_CLASS"] = "2"
        grid.evaluate(kernel="recur")
        lp = grid.loglikelihoods()
        del os.environ["COVEST_DIAG_BASIC_CLASS"]
    grid.evaluate(kernel="recur")
    fast = grid.loglikelihoods()
    grid.evaluate(kernel="direct")
    ref = grid.loglikelihoods()
    bad_inf = np.flatnonzero(np.isneginf(fast) != np.isneginf(ref))
    print("points whose -inf-ness differs:", len(bad_inf))
    for i in bad_inf[:40]:
        print("  flat", int(i), "point", grid.point(int(i)), "recur", fast[i], "direct", ref[i],
              "" if cls is None else "class %g log p_min %.6f" % (cls[i], lp[i]))
    fin = np.isfinite(ref) & np.isfinite(fast)
    err = np.abs(fast[fin] - ref[fin]) / np.abs(ref[fin])
    order = np.argsort(-err)[:20]
    idx = np.flatnonzero(fin)
    print("finite in both:", int(fin.sum()), "worst rel err", float(err.max()), "above 1e-11:", int((err > 1e-11).sum()),
          "above 1e-12:", int((err > 1e-12).sum()))
    for k in order[:12]:
        i = int(idx[k])
        print("  flat", i, "recur", fast[i], "direct", ref[i], "rel", err[k], "" if cls is None else "class %g log p_min %.6f" % (cls[i], lp[i]))
    if cls is not None:
        for c in range(5):
            sel = fin & (cls == c)
            if sel.any():
                e = np.abs(fast[sel] - ref[sel]) / np.abs(ref[sel])
                print("class", c, "points", int((cls == c).sum()), "finite", int(sel.sum()), "worst rel err", float(e.max()))


if __name__ == "__main__":
    main()
