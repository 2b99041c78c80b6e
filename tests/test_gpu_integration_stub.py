"""INTEGRATION.md is code a maintainer is meant to paste into the reference: run it.  The ctypes stub of
section 2 is extracted from the document, executed against the built library, and fed stand-ins that carry
exactly the attributes of the reference's model classes (covest/models.py:17-31,173-184)."""
import os
import re
import types

import numpy as np
import pytest

from conftest import REPO, load_hist, rel_err

pytestmark = pytest.mark.gpu


def _stub_namespace():
    text = open(os.path.join(REPO, "INTEGRATION.md")).read()
    section = text[text.index("## 2. `covest/hip_backend.py`"):text.index("## 3. The three edits")]
    code = re.search(r"```python\n(.*?)```", section, re.S).group(1)
    from covest_amd import build
    os.environ["COVEST_HIP_LIB"] = build.build()
    ns = {}
    exec(compile(code, "INTEGRATION.md:hip_backend", "exec"), ns)
    return ns


def _reference_like(kind, k, r, hist, tail, max_error=8, threshold=1e-8):
    """The attribute surface the stub reads, with the reference's own formulas."""
    from scipy.special import comb
    m = types.SimpleNamespace()
    m.repeats = kind == "repeats"
    m.k, m.r, m.hist, m.tail = k, r, hist, tail
    m.comb = [comb(k, s) * (3 ** s) for s in range(k + 1)]
    m.max_error = min(k + 1, max_error)
    m.bounds = ((0.01, None), (0, 0.5))
    if m.repeats:
        m.bounds = m.bounds + ((0.3, 1), (0, 1), (0, 1))
        m.threshold = threshold
    m.param_count = 5 if m.repeats else 2
    return m


def test_the_documented_stub_runs_and_agrees(hip_lib, oracle):
    ns = _stub_namespace()
    full = load_hist("sim_c10_e0.05")
    # with a tail, keep only the first 8 keys: on the whole histogram 1 - sp_j is ~1e-9 and the reference's own
    # tail term is rounding noise (tests/test_gpu_parity.py::_tail_noise)
    head = {j: h for j, h in list(full.items())[:8]}
    for kind, points in (("basic", [(10.0, 0.05), (8.5, 0.01), (12.0, 0.08)]),
                         ("repeats", [(10.0, 0.05, 0.8, 0.5, 0.3), (9.0, 0.04, 0.4, 0.1, 0.9)])):
        for hist, tail in ((full, 0), (head, 321)):
            like = ns["HipLikelihood"](_reference_like(kind, 21, 100, hist, tail))
            got = like.loglikelihoods(points)
            om = oracle.OracleModel(kind, 21, 100, hist, tail, max_error=8)
            want = om.compute_loglikelihood_many(np.array(points), n_threads=2)
            for a, b in zip(got, want):
                assert rel_err(float(a), float(b)) <= 1e-9, (kind, tail, a, b)
            ns["_lib"].covest_model_destroy(like.handle)
