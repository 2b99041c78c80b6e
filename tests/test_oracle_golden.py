"""Pin the oracle (oracle/covest_oracle.c) against the golden vectors generated
from the reference itself (tests/golden/make_golden.py).  CPU only.

The oracle repeats the reference's operations in the reference's order with the
same libm, so agreement is expected to the last bit; the asserted bound is
1e-15 relative with IEEE specials required to match exactly.
"""
import math
import os

import numpy as np
import pytest

from conftest import load_golden, load_hist, rel_err

TIGHT = 1e-15


def _model(oracle, kind, case, hist=None):
    hist = load_hist(case["hist"]) if hist is None else hist
    kw = dict(max_error=case["max_error"])
    if kind == "basic":
        kw["max_cov"] = case.get("max_cov")
    else:
        kw["threshold"] = case.get("threshold", 1e-8)
        kw["min_single_copy_ratio"] = case.get("min_single_copy_ratio", 0.3)
    return oracle.OracleModel(kind, case["k"], case["r"], hist, case["tail"], **kw)


def test_truncated_poisson_table(oracle):
    rows = load_golden("tp_table.json")["rows"]
    assert len(rows) > 400
    for l, j, want in rows:
        got = oracle.truncated_poisson(l, j)
        assert rel_err(got, want) == 0.0, (l, j, got, want)


def test_truncated_poisson_specials(oracle):
    # l == 0 / NaN: intent of c_src/covest_poissonmodule.c:15-17 is 0.0
    assert oracle.truncated_poisson(0.0, 5) == 0.0
    assert oracle.truncated_poisson(float("nan"), 5) == 0.0
    # overflow of the long-double product: SURVEY 8(a) A1 (ii)
    assert oracle.truncated_poisson(11500.0, 10000) == math.inf
    assert 0 < oracle.truncated_poisson(11400.0, 10000) < 1e-40


def test_threshold_table(oracle):
    g = load_golden("threshold_o.json")
    for hist_max, q1, q2, q, want in g["rows"]:
        assert oracle.threshold_o(q1, q2, q, 1e-8, hist_max) == want, (hist_max, q1, q2, q)


@pytest.mark.parametrize("kind,fname", [("basic", "basic_ll.json"), ("repeats", "repeats_ll.json")])
def test_loglikelihood_cases(oracle, kind, fname):
    g = load_golden(fname)
    worst = 0.0
    n = 0
    for case in g["cases"]:
        m = _model(oracle, kind, case)
        got = m.compute_loglikelihood_many(np.array(case["points"]), n_threads=4)
        for p, a, b in zip(case["points"], got, case["ll"]):
            e = rel_err(float(a), b)
            assert e <= TIGHT, (case["hist"], case["tail"], case["max_error"], p, a, b)
            worst = max(worst, e)
            n += 1
        for d in case["detail"]:
            probs = m.compute_probabilities(*d["point"])
            for j, want in d["p_j"]:
                assert rel_err(probs[j], want) <= TIGHT, (d["point"], j)
    assert n > 300
    print(kind, "points", n, "worst rel err", worst)


def test_config1_full_grid(oracle):
    g = load_golden("c1_grid.json")
    m = _model(oracle, "basic", g)
    pts = np.array([(c, e) for c in g["c_axis"] for e in g["e_axis"]])
    got = m.compute_loglikelihood_many(pts, n_threads=8)
    for a, b in zip(got, g["ll"]):
        assert rel_err(float(a), b) <= TIGHT
    arg, best = oracle.first_min(-got)
    assert arg == g["argmin_flat"] == 24 * 50 + 9
    assert best == g["min_negll"]


def test_config2_sample(oracle):
    g = load_golden("c2_sample.json")
    m = _model(oracle, "basic", g)
    sel = list(range(0, len(g["points"]), 8))  # 32 of the 256 points: O(B^2) each on the CPU
    got = m.compute_loglikelihood_many(np.array([g["points"][i] for i in sel]), n_threads=8)
    for i, a in zip(sel, got):
        assert rel_err(float(a), g["ll"][i]) <= TIGHT, (g["points"][i], a, g["ll"][i])


def test_config3_sample_cheapest(oracle):
    g = load_golden("c3_sample.json")
    m = _model(oracle, "repeats", g)
    order = sorted(range(len(g["points"])), key=lambda i: g["cpu_seconds_per_point"][i])[:8]
    got = m.compute_loglikelihood_many(np.array([g["points"][i] for i in order]), n_threads=8)
    for i, a in zip(order, got):
        assert rel_err(float(a), g["ll"][i]) <= TIGHT, (g["points"][i], a, g["ll"][i])


def test_config3_trimmed_fixture(oracle):
    """tests/golden/c3_trim.json (the reference on H10k_rep trimmed by its own trim_hist, tail = the trimmed mass):
    the trimmed histogram is what this repository's trim_hist makes of H10k_rep, and the oracle reproduces the
    reference's LL and sp_j on every 24th sample point and on the arg-min candidates' winner -- bit-tight, tail
    term included (the same fsum, the same log)."""
    from covest_amd import hist_steps
    g = load_golden("c3_trim.json")
    src = load_hist(g["source_hist"])
    trim = hist_steps.get_trim(src, ignore_last=True)
    thist, tail = hist_steps.trim_hist(src, trim)
    assert trim == g["trim"] and tail == g["tail"] and thist == load_hist(g["hist"])
    m = oracle.OracleModel("repeats", g["k"], g["r"], thist, tail, max_error=g["max_error"])
    q1s, qs = g["axes"][2], g["axes"][3]

    def point(i):
        return (g["axes"][0][i // 8192], g["axes"][1][(i // 256) % 32], q1s[(i // 16) % 16], g["q2"], qs[i % 16])

    sel = list(range(0, len(g["flat_index"]), 24))
    got = m.compute_loglikelihood_many(np.array([point(g["flat_index"][k]) for k in sel]), n_threads=8)
    for k, a in zip(sel, got):
        assert rel_err(float(a), g["ll"][k]) <= TIGHT, (k, a, g["ll"][k])
    for k in sel[::8]:
        if math.isfinite(g["ll"][k]):
            sp = math.fsum(m.compute_probabilities(*point(g["flat_index"][k])).values())
            assert abs(sp - g["sp"][k]) <= 4e-16, (k, sp, g["sp"][k])
    c = g["candidates"]
    w = c["flat_index"].index(c["reference_argmin_flat"])
    assert rel_err(m.compute_loglikelihood(*point(c["reference_argmin_flat"])), c["ll"][w]) <= TIGHT
    assert -c["ll"][w] == c["reference_min_negll"] == min(-v for v in c["ll"] if v == v)
    # well conditioned where it matters: at the best points 1 - sp_j ~ tail / N = 1e-4 -- or sp_j > 1 by as much
    # (the 200-chunk normaliser of the reference makes some pmfs too large; the term is then 0)
    assert all(abs(1.0 - sp) > 1e-6 for sp in c["sp"])


def test_overflow_domain(oracle):
    """Where the reference's long-double pmf product overflows (c_src/covest_poissonmodule.c:19-24) its likelihood is
    +inf or NaN (tests/golden/overflow.json, make_golden.py section overflow): the oracle's faithful mode returns the
    same specials in the same places, and the finite values beside them."""
    g = load_golden("overflow.json")
    seen = set()
    for c in g["cases"]:
        hist = {int(j): int(v) for j, v in c["hist_items"]}
        om = oracle.OracleModel(c["model"], c["k"], c["r"], hist, c["tail"], max_error=c["max_error"])
        got = om.compute_loglikelihood_many(np.array(c["points"]), n_threads=4)
        for a, b, p in zip(got, c["ll"], c["points"]):
            assert rel_err(float(a), b) <= 1e-13, (c["model"], c["hist"], c["tail"], p, float(a), b)
            seen.add("nan" if b != b else "+inf" if b == math.inf else "-inf" if b == -math.inf else "finite")
    assert seen == {"nan", "+inf", "-inf", "finite"}


def test_synthetic_histograms_match_survey():
    # H256 has 73 non-zero bins and sum 10 000 003; H10k_basic 367 non-zero bins
    h = load_hist("H256")
    assert len(h) == 256 and sum(1 for v in h.values() if v) == 73 and sum(h.values()) == 10000003
    h = load_hist("H10k_basic")
    assert len(h) == 10000 and sum(1 for v in h.values() if v) == 367
    h = load_hist("H10k_rep")
    assert len(h) == 10000 and sum(1 for v in h.values() if v) > 500


def test_first_min_rules(oracle):
    inf, nan = math.inf, math.nan
    assert oracle.first_min([3.0, 1.0, 1.0, 2.0]) == (1, 1.0)         # first wins ties
    assert oracle.first_min([nan, 5.0, nan, 4.0]) == (3, 4.0)         # NaN never wins
    assert oracle.first_min([inf, inf]) == (-1, inf)                  # +inf never wins
    assert oracle.first_min([1.0, -inf, -inf]) == (1, -inf)           # -inf does
    assert oracle.first_min([]) == (-1, inf)
    assert oracle.first_min([5.0, 6.0], start=4.0) == (-1, 4.0)


def test_fast_mode_agrees(oracle):
    """The log-domain CPU mode (reported as a second baseline) against the faithful one."""
    for kind, fname in (("basic", "basic_ll.json"), ("repeats", "repeats_ll.json")):
        g = load_golden(fname)
        for case in g["cases"]:
            if case["tail"]:
                continue  # the tail term amplifies the 1e-12 term error near sp_j = 1
            m = _model(oracle, kind, case)
            pts = np.array(case["points"])
            fast = m.compute_loglikelihood_many_fast(pts, n_threads=4)
            for a, b in zip(fast, case["ll"]):
                assert rel_err(float(a), b) <= 1e-10
    g = load_golden("c3_sample.json")
    m = _model(oracle, "repeats", g)
    fast = m.compute_loglikelihood_many_fast(np.array(g["points"]), n_threads=8)
    for a, b in zip(fast, g["ll"]):
        assert rel_err(float(a), b) <= 1e-10
