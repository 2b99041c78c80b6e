#!/usr/bin/env python3
"""GPU box: evaluate the C3 grid -- and the C2 grid --  and write the arg-min candidates that the
reference itself has to confirm (parity procedure of 8(d)): the GPU's top-64 points by -LL
plus the 2 P axis neighbours of its arg-min.  Output: gpurun_out/c3_candidates.json, c2_candidates.json -- flat
indices and the GPU's values only; tests/golden/make_golden.py section `c3argmin` turns them
into a fixture by running the reference on exactly these points (build container)."""
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))


def main():
    from conftest import load_hist
    from covest_amd import DenseGrid, RepeatsModel
    out = {}
    for tail in (0, 1000):
        m = RepeatsModel(21, 100, load_hist("H10k_rep"), tail, max_error=8)
        axes = [np.linspace(15.0, 30.0, 32), np.linspace(0.005, 0.08, 32), np.linspace(0.3, 0.95, 16),
                [0.5], np.linspace(0.05, 0.95, 16)]
        grid = DenseGrid(m, axes)
        grid.evaluate(kernel="factored")
        ll = grid.loglikelihoods()
        val, arg = grid.argmin()
        negll = np.where(np.isnan(ll), np.inf, -ll)
        cand = set(np.argsort(negll, kind="stable")[:64].tolist())
        idx = np.unravel_index(arg, grid.shape)
        for d in range(len(grid.shape)):
            for step in (-1, 1):
                j = list(idx)
                j[d] += step
                if 0 <= j[d] < grid.shape[d]:
                    cand.add(int(np.ravel_multi_index(j, grid.shape)))
        cand = sorted(cand)
        out["tail%d" % tail] = {"argmin_flat": int(arg), "min_negll": float(val), "candidates": cand,
                                "gpu_ll": [float(ll[i]) for i in cand],
                                "n_finite": int(np.isfinite(ll).sum()), "n_neg_inf": int(np.isneginf(ll).sum())}
        print("tail", tail, "arg-min", arg, val, "candidates", len(cand))
        grid.close()
        m.close()
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    with open(os.path.join(REPO, "gpurun_out", "c3_candidates.json"), "w") as f:
        json.dump(out, f)
    # the same for config 2 (BasicModel, H10k_basic, 1000 x 1000, tail 0): tests/golden/make_golden.py section c2argmin
    from covest_amd import BasicModel
    m = BasicModel(21, 100, load_hist("H10k_basic"), 0, max_error=8)
    grid = DenseGrid(m, [np.linspace(2000.0, 6000.0, 1000), np.linspace(0.001, 0.1, 1000)])
    grid.evaluate(kernel="recur")
    ll = grid.loglikelihoods()
    val, arg = grid.argmin()
    negll = np.where(np.isnan(ll), np.inf, -ll)
    cand = set(np.argsort(negll, kind="stable")[:64].tolist())
    idx = np.unravel_index(arg, grid.shape)
    for d in range(2):
        for step in (-1, 1):
            j = list(idx)
            j[d] += step
            if 0 <= j[d] < grid.shape[d]:
                cand.add(int(np.ravel_multi_index(j, grid.shape)))
    cand = sorted(cand)
    c2 = {"tail0": {"argmin_flat": int(arg), "min_negll": float(val), "candidates": cand, "gpu_ll": [float(ll[i]) for i in cand],
                    "n_finite": int(np.isfinite(ll).sum()), "n_neg_inf": int(np.isneginf(ll).sum())}}
    print("C2 tail 0 arg-min", arg, val, "candidates", len(cand))
    with open(os.path.join(REPO, "gpurun_out", "c2_candidates.json"), "w") as f:
        json.dump(c2, f)


if __name__ == "__main__":
    main()
