#!/usr/bin/env python3
"""Where along the walk the waves of K-factored wait (DIAGNOSTIC build, COVEST_FACTORED_DIAG=2): mean s_memtime ticks
a wave spends at the interval barrier, by eighths of the walk (position 0 = the last key tile, then tiles 0 .. n-2).
    COVEST_AMD_LIB=$PWD/tools/bin/libcovest_amd_diag.so python tools/factored_diag_walk.py"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["COVEST_FACTORED_DIAG"] = os.environ.get("COVEST_DIAG_WALK_MODE", "2")  # 3: the first seven intervals one by one
from bench import load_hist, workload  # noqa: E402
from covest_amd import DenseGrid, RepeatsModel, _capi  # noqa: E402

kind, hname, axes = workload("c3", 1)
m = RepeatsModel(21, 100, load_hist(hname), 0, max_error=8)
g = DenseGrid(m, axes)
g.evaluate(kernel="factored")
g.argmin()
L = _capi.lib()
n = L.covest_grid_diag(g._handle, None, 0)
buf = np.zeros(n, dtype=np.int64)
L.covest_grid_diag(g._handle, buf.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), n)
d = buf.reshape(-1, 8, 8).astype(np.float64)  # [wg][wave][eighth of the walk]
print("workgroups", d.shape[0], " barrier wait per wave and eighth of the walk (mean ticks), last column: sum")
for w in range(8):
    row = d[:, w, :].mean(axis=0)
    print("wave %d: " % w + " ".join("%7.0f" % v for v in row) + "   %8.0f" % row.sum())
