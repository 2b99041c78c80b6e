import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init(); torch.cuda.synchronize()
x = torch.zeros(1024, device="cuda"); torch.cuda.synchronize()
from bench import load_hist
from covest_amd import BasicModel, RepeatsModel, DenseGrid, _capi
L = _capi.lib()
t0 = time.perf_counter(); n = L.covest_device_count(); t1 = time.perf_counter()
print("covest_device_count %d: %.2f ms" % (n, 1e3 * (t1 - t0)))
t0 = time.perf_counter(); m = BasicModel(21, 100, {1: 5, 2: 3}, 0, max_error=8); m.handle; t1 = time.perf_counter()
print("tiny basic model handle: %.2f ms" % (1e3 * (t1 - t0)))
t0 = time.perf_counter(); m2 = BasicModel(21, 100, {1: 5, 2: 3, 3: 1}, 0, max_error=8); m2.handle; t1 = time.perf_counter()
print("second tiny model handle: %.2f ms" % (1e3 * (t1 - t0)))
hist = load_hist("H10k_rep")
t0 = time.perf_counter(); r = RepeatsModel(21, 100, hist, 0, max_error=8); r.handle; t1 = time.perf_counter()
print("10k-key repeats model handle: %.2f ms" % (1e3 * (t1 - t0)))
t0 = time.perf_counter(); v = m.compute_loglikelihood(10.0, 0.05); t1 = time.perf_counter()
print("first basic evaluation: %.2f ms" % (1e3 * (t1 - t0)))
import subprocess
print(subprocess.run("grep -c amdhip64 /proc/%d/maps; grep amdhip64 /proc/%d/maps | awk '{print $6}' | sort -u" % (os.getpid(), os.getpid()), shell=True, capture_output=True, text=True).stdout)
