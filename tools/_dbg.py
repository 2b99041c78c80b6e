import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from conftest import load_golden, load_hist
from covest_amd import BasicModel
g = load_golden("c2_sample.json")
m = BasicModel(21, 100, load_hist("H10k_basic"), 0, max_error=8)
pts = np.array(g["points"])[[113, 161]]
os.environ["COVEST_DEBUG_WORDS"] = "1"
a = m.loglikelihood_points(pts, kernel="recur")
os.environ["COVEST_DEBUG_NOFIX"] = "1"
r = m.loglikelihood_points(pts, kernel="recur")
b = m.loglikelihood_points(pts, kernel="direct")
print(a, r, b)
keys = list(m.hist.keys()); ev = [k for k in keys if m.hist[k]]
print(len(ev), ev[-30:])
