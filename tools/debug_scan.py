import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import load_hist, workload
from covest_amd import DenseGrid, RepeatsModel, _capi
import ctypes
kind, hname, axes = workload("c3", 1)
m = RepeatsModel(21, 100, load_hist(hname), 0, max_error=8)
res = {}
for k in ("direct", "factored", "scan"):
    g = DenseGrid(m, axes); g.evaluate(kernel=k); res[k] = g.loglikelihoods().copy()
d = res["direct"]
for k in ("factored", "scan"):
    with np.errstate(invalid="ignore", divide="ignore"):
        rel = np.abs(res[k] - d) / np.abs(d)
    rel[~np.isfinite(rel)] = 0
    bad_inf = np.sum(np.isinf(res[k]) != np.isinf(d))
    print(k, "worst rel", rel.max(), "inf mismatch", bad_inf)
    idx = np.argsort(rel)[-12:][::-1]
    q3 = np.array([(a, b, c) for a in axes[2] for b in axes[3] for c in axes[4]])
    T = np.zeros(len(q3), dtype=np.int32)
    L = _capi.lib()
    L.covest_threshold_o(len(q3), np.ascontiguousarray(q3).ctypes.data_as(ctypes.POINTER(ctypes.c_double)), ctypes.c_double(1e-8), 1, 10000, T.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
    for i in idx:
        ce, q = divmod(int(i), 256)
        print("  flat", i, "ce", ce, "q1idx", q // 16, "qidx", q % 16, "T", T[q], "rel", rel[i], res[k][i], d[i])
    # error by q index
    r2 = rel.reshape(-1, 16, 16)
    print("  max rel by qidx:", np.array2string(r2.max(axis=(0, 1)), precision=2))
    print("  max rel by q1idx:", np.array2string(r2.max(axis=(0, 2)), precision=2))
g = DenseGrid(m, axes); g.evaluate(kernel="scan"); again = g.loglikelihoods().copy()
print("scan deterministic:", np.array_equal(again, res["scan"], equal_nan=True))
with np.errstate(invalid="ignore", divide="ignore"):
    rel = np.abs(res["scan"] - d) / np.abs(d)
rel[~np.isfinite(rel)] = 0
bad = (rel > 1e-11).reshape(-1, 16, 16)
print("bad count total", bad.sum(), "of", bad.size)
print("bad per (q1idx rows, qidx cols):")
print(bad.sum(axis=0))
cnt = bad.reshape(32, 32, 256).sum(axis=2)
print("bad per (c rows, e cols):")
print(cnt)
