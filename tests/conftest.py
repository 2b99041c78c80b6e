import json
import math
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(REPO, "tests", "golden")
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def load_hist(name):
    """`.hist` text format of covest/data.py:22-41 ("j count" lines, '#' comments)."""
    hist = {}
    with open(os.path.join(GOLDEN, name + ".hist")) as f:
        for line in f:
            if not line.strip() or line[0] == "#":
                continue
            a, b = line.split()[:2]
            hist[int(a)] = int(b)
    return hist


def rel_err(got, want):
    """Relative error with IEEE specials required to match exactly."""
    if want != want:
        return 0.0 if got != got else math.inf
    if math.isinf(want) or math.isinf(got):
        return 0.0 if got == want else math.inf
    if want == 0.0:
        return abs(got)
    return abs(got - want) / abs(want)


@pytest.fixture(scope="session")
def oracle():
    from oracle import covest_oracle
    covest_oracle.build()
    return covest_oracle


@pytest.fixture(scope="session")
def hip_lib():
    """The built HIP library (built here on CPU by hipcc; must already exist on the GPU box)."""
    from covest_amd import _capi, build
    build.build()
    return _capi.lib()
