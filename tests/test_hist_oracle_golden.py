"""The CPU restatement of the histogram steps (oracle/hist_oracle.py) against the vectors the reference
produced (tests/golden/hist_steps.json)."""
import math

import pytest

from conftest import load_golden, load_hist, rel_err

from oracle import hist_oracle as ho

G = load_golden("hist_steps.json")


def _close(a, b, tol):
    return a == b or rel_err(a, b) <= tol


def test_poisson_dist_faithful_including_the_defect():
    for case in G["poisson_dist"]:
        got = ho.poisson_dist(case["l"], case["max_j"], faithful=True)
        assert len(got) == len(case["p"])
        for j, (a, b) in enumerate(zip(got, case["p"]), start=1):
            assert _close(a, b, 1e-13), (case["l"], j, a, b)


def test_poisson_dist_exact_agrees_below_200_and_differs_above():
    for case in G["poisson_dist"]:
        if case["l"] == 0:
            continue
        exact = ho.poisson_dist(case["l"], case["max_j"], faithful=False)
        errs = [rel_err(a, b) for a, b in zip(exact, case["p"]) if max(a, b) > 1e-300]
        if case["l"] <= 200:
            assert errs and max(errs) <= 1e-12, (case["l"], max(errs))
        else:  # the reference's own values are wrong there: orders of magnitude apart
            bulk = [(a, b) for a, b in zip(exact, case["p"]) if a > 1e-12]  # where the probability mass is
            assert bulk and all(not (0.5 * a < b < 2 * a) for a, b in bulk), case["l"]
            if case["max_j"] >= 1.5 * case["l"]:
                assert abs(sum(exact) - 1.0) < 1e-9  # the pmf the GPU path computes sums to 1


def test_binom_pmf():
    for case in G["binom_pmf"]:
        p = 1.0 / case["factor"]
        for j, want in enumerate(case["p"], start=1):
            assert _close(ho.binom_pmf(case["i"], p, j), want, 1e-12)


@pytest.mark.parametrize("name", sorted(G["cases"]))
def test_histogram_steps(name):
    case = G["cases"][name]
    hist = load_hist(name if name != "H10k_rep_le640" else "H10k_rep")
    if name == "H10k_rep_le640":
        hist = {k: v for k, v in hist.items() if k <= 640}
    c, e = ho.compute_coverage_apx(hist, case["k"], case["r"])
    assert _close(c, case["coverage_apx"][0], 1e-12) and _close(e, case["coverage_apx"][1], 1e-12)
    assert ho.get_trim(hist) == case["get_trim"]
    assert ho.get_trim(hist, ignore_last=True) == case["get_trim_ignore_last"]
    th, tail = ho.trim_hist(hist, case["trim_hist"]["threshold"])
    assert [[k, v] for k, v in th.items()] == case["trim_hist"]["hist"] and tail == case["trim_hist"]["tail"]
    for s in case["sample"]:
        exp = ho.sample_expected(hist, factor=s["factor"], trim=s["trim"], faithful=True)
        want = {k: v for k, v in s["expected"]}
        got = {k: v for k, v in exp.items() if v > 0}
        assert list(got) == list(want)
        for k in want:
            assert _close(got[k], want[k], 1e-12), (name, s["factor"], k, got[k], want[k])
        rounded = ho.round_sampled(exp, s["uniforms"])
        assert [[k, v] for k, v in rounded.items()] == s["rounded"]


def test_c_twin_of_the_thinning_loop():
    """oracle_thin_expected (C, used for timing and full sizes) == the pinned Python restatement."""
    import numpy as np
    hist = {k: v for k, v in load_hist("H10k_rep").items() if k <= 640}
    for faithful in (True, False):
        want = ho.sample_expected(hist, factor=2, trim=700, faithful=faithful)
        kept = {k: v for k, v in hist.items() if k < 640}  # what sample_expected keeps with this trim
        got = ho.thin_expected_c(list(kept), [float(v) for v in kept.values()], 2, max(kept), faithful=faithful)
        for j, v in want.items():
            assert _close(float(got[j - 1]), v, 1e-13 if faithful else 1e-11), (faithful, j, got[j - 1], v)
