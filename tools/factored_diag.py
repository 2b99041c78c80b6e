#!/usr/bin/env python3
"""Per-wave phase cycles of K-factored on C3 (in-kernel s_memtime stamps).
Run on the GPU box:  COVEST_FACTORED_DIAG=1 python tools/factored_diag.py"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("COVEST_FACTORED_DIAG", "1")
from bench import load_hist, workload, workload_tail  # noqa: E402
from covest_amd import DenseGrid, RepeatsModel, _capi  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"  # c3, or c3t: the trimmed histogram with its tail
kind, hname, axes = workload(wl, 1)
m = RepeatsModel(21, 100, load_hist(hname), float(os.environ.get("COVEST_DIAG_TAIL", str(workload_tail(wl)))), max_error=8)  # COVEST_DIAG_TAIL=1000 with c3: all 10 000 keys
g = DenseGrid(m, axes)
g.evaluate(kernel="factored")
g.argmin()
L = _capi.lib()
n = L.covest_grid_diag(g._handle, None, 0)
buf = np.zeros(n, dtype=np.int64)
L.covest_grid_diag(g._handle, buf.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), n)
if os.environ.get("COVEST_FACTORED_DIAG") == "4":  # the stages OUTSIDE the walk, per wave (ll_factored.hip dgx_*)
    st = buf.reshape(-1, 8, 8).astype(np.float64)
    print("workgroups", st.shape[0])
    print("mean cycles per wave: start -> first barrier | streams' constants | units' tables, first weights | the walk | "
          "per-q results | whole workgroup")
    for w in range(8):
        print("wave %d: %8.0f %8.0f %8.0f %9.0f %8.0f %9.0f" % ((w,) + tuple(st[:, w, k].mean() for k in range(6))))
    t0, t1 = buf.reshape(-1, 8, 8)[:, 0, 6], buf.reshape(-1, 8, 8)[:, 0, 7]
    # a CU's turnover: the stamps are one clock per XCD and workgroup i runs on XCD i mod 8 -- within an XCD (32 CUs,
    # one workgroup each) the k-th start past the first 32 against the k-th end
    wg = np.arange(len(t0))
    for x in range(8):
        sel = wg % 8 == x
        a, b = np.sort(t0[sel]), np.sort(t1[sel])
        n_cu = 32
        if len(a) <= n_cu:
            continue
        gap = a[n_cu:] - b[:len(a) - n_cu]
        print("XCD %d: %d workgroups, first start to last end %.0f cycles, lengths mean %.0f; start of the (32 + k)-th minus the "
              "k-th end: mean %.0f, median %.0f, max %.0f; first 32 starts spread over %.0f" % (
                  x, int(sel.sum()), float(b[-1] - a[0]), float((t1[sel] - t0[sel]).mean()), float(gap.mean()),
                  float(np.median(gap)), float(gap.max()), float(a[n_cu - 1] - a[0])))
    sys.exit(0)
d = buf.reshape(-1, 8, 8)[:, :, :4].astype(np.float64)  # [wg][wave][build, contract, log, barrier]
extra = buf.reshape(-1, 8, 8)
print("workgroups", d.shape[0])
print("key tiles (of %d) whose 64 G columns of a builder wave were all zero: " % m.bins_evaluated +
      "  ".join("wave %d: %.1f" % (w, extra[:, w, 5].mean()) for w in range(5)))
tot = d.sum(axis=2)
print("mean cycles per wave (s_memtime ticks): total %.0f" % tot.mean())
for w in range(8):
    a, b, c, bar = d[:, w, :].mean(axis=0)
    b0 = extra[:, w, 4].mean()  # shared steps + first MFMA step
    ent, fst = extra[:, w, 6].mean(), extra[:, w, 7].mean()
    print("wave %d: enter %7.0f  walk %7.0f  wait-first %7.0f  shared+first %7.0f  contract %7.0f  log %7.0f  barrier %7.0f   total %8.0f" % (
        w, ent, a, fst, b0, b, c, bar, ent + a + fst + b0 + b + c + bar))
