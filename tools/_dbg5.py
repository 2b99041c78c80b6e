import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from conftest import load_hist
from covest_amd import RepeatsModel
m = RepeatsModel(21, 100, load_hist("sim_c10_e0.05"), 0, max_error=8)
print(m.compute_loglikelihood(10.0, 0.05, 0.65, 0.5, 0.5))
