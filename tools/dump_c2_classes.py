#!/usr/bin/env python3
"""GPU box, DIAGNOSTIC library (python -m covest_amd.build --out tools/bin/libcovest_amd_diag.so -DCOVEST_DIAG, loaded
through COVEST_AMD_LIB): which route every point of the C2 grid takes through K-basic's closed form
(ll_basic.hip: closed form / -inf by its bound / the key-by-key walk), and how close to the class boundaries it is.
Writes gpurun_out/c2_classes.json: per class the flat indices where the REFERENCE is to be asked
(tests/golden/make_golden.py section `c2classes` turns them into tests/golden/c2_classes.json) --
  walk_self     lanes whose own log p_j lies between the -inf bound and the clamp (they send their wave to the walk)
  walk_dragged  lanes of such a wave that would have taken the closed form or the bound
  closed_edge   closed-form lanes closest to the clamp (smallest margin of log p_min over it)
  closed_far    closed-form lanes, seeded
  neginf_edge   -inf-by-bound lanes closest to the bound (largest log p_min below -746.5)
  neginf_far    -inf-by-bound lanes, seeded
  flush_edge    walking lanes whose smallest p_j is closest to HALF A GRID STEP of the doubles (2^-1075): where a product
                rounded once flushes to 0 and the reference's term-by-term roundings may not (direct_point.h kZeroSteps;
                flat index 826002 is the point that was -inf here and finite in the reference until round 4)
Only indices (and the class counts) leave this script: every value of the fixture is the reference's."""
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
PER_CLASS = int(os.environ.get("COVEST_C2_PER_CLASS", "72"))


def main():
    from conftest import load_hist
    from covest_amd import BasicModel, DenseGrid
    m = BasicModel(21, 100, load_hist("H10k_basic"), 0, max_error=8)
    cs = np.linspace(2000.0, 6000.0, 1000)
    es = np.linspace(0.001, 0.1, 1000)
    grid = DenseGrid(m, [cs, es])
    os.environ["COVEST_DIAG_BASIC_CLASS"] = "1"
    grid.evaluate(kernel="recur")
    cls = grid.loglikelihoods().copy()
    os.environ["COVEST_DIAG_BASIC_CLASS"] = "2"
    grid.evaluate(kernel="recur")
    lp = grid.loglikelihoods().copy()
    del os.environ["COVEST_DIAG_BASIC_CLASS"]
    grid.evaluate(kernel="recur")
    ll = grid.loglikelihoods().copy()
    assert set(np.unique(cls).tolist()) <= {0.0, 1.0, 2.0, 3.0, 4.0}, "not the diagnostic library?"
    counts = {int(c): int((cls == c).sum()) for c in range(5)}
    print("classes", counts, "finite LL", int(np.isfinite(ll).sum()))
    rng = np.random.default_rng(20241004)

    def pick(mask, key=None, n=PER_CLASS):
        idx = np.flatnonzero(mask)
        if len(idx) <= n:
            return idx.tolist()
        if key is None:
            return np.sort(rng.choice(idx, size=n, replace=False)).tolist()
        return np.sort(idx[np.argsort(key[idx], kind="stable")[:n]]).tolist()

    out = {"what": "K-basic's routes over the C2 grid (indices only)", "class_counts": counts,
           "walk_self": pick(cls == 4.0), "walk_dragged": pick(cls == 3.0),
           "closed_edge": pick(cls == 1.0, lp), "closed_far": pick(cls == 1.0),
           "neginf_edge": pick(cls == 2.0, -lp), "neginf_far": pick(cls == 2.0),
           "flush_edge": pick(cls == 4.0, np.abs(lp + 1075.0 * np.log(2.0)))}
    for k, v in out.items():
        if isinstance(v, list):
            print(k, len(v), "log p_min", float(np.nanmin(lp[v])) if v else None, float(np.nanmax(lp[v])) if v else None,
                  "finite", int(np.isfinite(ll[v]).sum()))
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    with open(os.path.join(REPO, "gpurun_out", "c2_classes.json"), "w") as f:
        json.dump(out, f)


if __name__ == "__main__":
    main()
