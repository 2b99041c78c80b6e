"""Where the FIRST search of a process spends its time (time-to-argmin, cold): HIP context up (torch), library not
yet touched.  One line per step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.cuda.init(); torch.cuda.synchronize()
from bench import load_hist, workload
t = [time.perf_counter()]
from covest_amd import DenseGrid, RepeatsModel, _capi
_capi.lib(); t.append(time.perf_counter())
kind, hname, axes = workload("c3", 1)
hist = load_hist(hname); t.append(time.perf_counter())
m = RepeatsModel(21, 100, hist, 0, max_error=8); m.handle; t.append(time.perf_counter())
g = DenseGrid(m, axes); t.append(time.perf_counter())
g.evaluate(); t.append(time.perf_counter())
g.argmin(); t.append(time.perf_counter())
names = ["import + dlopen", "read .hist", "model handle", "grid handle + plan", "launch (module load)", "wait + read-back"]
for n, a, b in zip(names, t[:-1], t[1:]):
    print("%-24s %8.2f ms" % (n, 1e3 * (b - a)))
g.evaluate(); t0 = time.perf_counter(); g.argmin(); print("second evaluate wait %.2f ms" % (1e3 * (time.perf_counter() - t0)))
