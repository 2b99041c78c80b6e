import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init(); torch.cuda.synchronize()
from covest_amd import BasicModel, _capi
L = _capi.lib()
m = BasicModel(21, 100, {1: 5, 2: 3}, 0, max_error=8)
t0 = time.perf_counter(); desc, keep = m._desc(); t1 = time.perf_counter()
h = ctypes.c_void_p()
rc = L.covest_model_create(ctypes.byref(desc), ctypes.byref(h)); t2 = time.perf_counter()
print("_desc %.2f ms, covest_model_create %.2f ms (rc %d)" % (1e3 * (t1 - t0), 1e3 * (t2 - t1), rc))
t0 = time.perf_counter(); rc = L.covest_model_create(ctypes.byref(desc), ctypes.byref(h)); t1 = time.perf_counter()
print("second covest_model_create %.2f ms" % (1e3 * (t1 - t0)))
