"""Where the reference's own C extension has been built (oracle/_ref, build
container only), the oracle's truncated_poisson must agree with it bit for bit."""
import glob
import importlib.util
import math
import os
import random

import pytest

from conftest import REPO


def _load_ref():
    hits = glob.glob(os.path.join(REPO, "oracle", "_ref", "covest_poisson*.so"))
    if not hits:
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    spec = importlib.util.spec_from_file_location("covest_poisson", hits[0])
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_truncated_poisson_bitwise(oracle):
    ref = _load_ref()
    rnd = random.Random(11)
    for _ in range(3000):
        l = math.exp(rnd.uniform(math.log(1e-10), math.log(2e4)))
        j = int(math.exp(rnd.uniform(0, math.log(12000))))
        a, b = oracle.truncated_poisson(l, j), ref.truncated_poisson(l, j)
        assert a == b or (a != a and b != b), (l, j, a, b)
    for l in (199.99999999, 200.0, 200.00000001, 400.0, 400.000000001, 1e-8, 1.0000001e-8):
        for j in (1, 5, 300):
            assert oracle.truncated_poisson(l, j) == ref.truncated_poisson(l, j)
