import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.cuda.init(); torch.cuda.synchronize()
x = torch.zeros(1024, device="cuda"); torch.cuda.synchronize()
hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
def T(label, f):
    t0 = time.perf_counter(); r = f(); print("%-44s %8.2f ms (rc %s)" % (label, 1e3 * (time.perf_counter() - t0), r))
p = ctypes.c_void_p()
T("hipMalloc 1 MB", lambda: hip.hipMalloc(ctypes.byref(p), 1 << 20))
buf = np.zeros(1 << 17)
T("hipMemcpy H2D 1 MB pageable", lambda: hip.hipMemcpy(p, buf.ctypes.data_as(ctypes.c_void_p), 1 << 20, 1))
T("hipMemcpy H2D again", lambda: hip.hipMemcpy(p, buf.ctypes.data_as(ctypes.c_void_p), 1 << 20, 1))
T("hipMemset", lambda: hip.hipMemset(p, 0, 1 << 20))
T("hipDeviceSynchronize", lambda: hip.hipDeviceSynchronize())
from covest_amd import BasicModel, _capi
L = _capi.lib()
t0 = time.perf_counter(); m = BasicModel(21, 100, {1: 5, 2: 3}, 0, max_error=8); m.handle; t1 = time.perf_counter()
print("tiny basic model handle: %.2f ms" % (1e3 * (t1 - t0)))
t0 = time.perf_counter(); v = m.compute_loglikelihood(10.0, 0.05); t1 = time.perf_counter()
print("first basic evaluation: %.2f ms" % (1e3 * (t1 - t0)))
