"""Parity of the HIP path (through the C ABI) against the golden vectors generated
from the reference and against the oracle.  Needs a real MI355X: `-m gpu`.

Tolerance: log-likelihood values <= 1e-9 relative (BASELINE.json north_star),
IEEE specials (-inf / NaN) identical, arg-min index identical.
"""
import json
import math
import os

import numpy as np
import pytest

from conftest import load_golden, load_hist, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-9


def _gpu_model(kind, case, hist=None):
    from covest_amd import BasicModel, RepeatsModel
    hist = load_hist(case["hist"]) if hist is None else hist
    if kind == "basic":
        return BasicModel(case["k"], case["r"], hist, case["tail"], max_error=case["max_error"],
                          max_cov=case.get("max_cov"))
    return RepeatsModel(case["k"], case["r"], hist, case["tail"], max_error=case["max_error"],
                        threshold=case.get("threshold", 1e-8),
                        min_single_copy_ratio=case.get("min_single_copy_ratio", 0.3))


def _oracle_model(oracle, kind, case, hist=None):
    hist = load_hist(case["hist"]) if hist is None else hist
    kw = dict(max_error=case["max_error"])
    if kind == "basic":
        kw["max_cov"] = case.get("max_cov")
    else:
        kw["threshold"] = case.get("threshold", 1e-8)
        kw["min_single_copy_ratio"] = case.get("min_single_copy_ratio", 0.3)
    return oracle.OracleModel(kind, case["k"], case["r"], hist, case["tail"], **kw)


class _Slack(list):
    """Per point the absolute difference tolerated in the tail term (a list of floats) plus `classes`: per point
    None (the plain 1e-9 decides), "graded" (the conditioning-proportional slack is wider than 1e-10 |LL|) or
    "flip" (the reference's own term is a coin toss); `unit`: |tail| eps / (1 - sp_j) per point, the first-order
    price of ONE eps of error in sp_j (0 where there is no tail term to speak of; for reports)."""
    classes = ()
    unit = ()


def _check(got, want, what, tol=TOL, slack=None):
    """slack[i] > 0: the absolute difference tolerated at point i when the relative one exceeds `tol` (see
    _tail_slack).  Returns the worst relative error among the points that met `tol`."""
    worst, used, worst_use = 0.0, 0, 0.0
    for i, (a, b) in enumerate(zip(got, want)):
        e = rel_err(float(a), float(b))
        if e > tol and slack is not None and slack[i] > 0 and abs(float(a) - float(b)) <= slack[i]:
            used += 1
            worst_use = max(worst_use, abs(float(a) - float(b)) / slack[i])
            continue
        assert e <= tol, "%s[%d]: got %r want %r (rel %.3g%s)" % (
            what, i, float(a), float(b), e,
            "" if slack is None or not slack[i] else ", |diff| %.3g > tail slack %.3g" % (abs(float(a) - float(b)), slack[i]))
        worst = max(worst, e)
    if used:
        print("%s: %d of %d points beyond 1e-9 but inside the tail slack (largest share of it used: %.2g)" % (
            what, used, len(want), worst_use))
    return worst


K_TAIL = 8.0  # rounding errors of K eps per key are granted to the GPU's sp_j (first-order propagation)


def _tail_slack(tail, ll, sp, n_keys):
    """What may separate a correct implementation from the reference in the tail term tail * log(1 - sp_j)
    (covest/models.py:103-104), given the sp_j = fsum(p_j) the REFERENCE saw.  The p_j of two correct
    implementations differ by rounding, so their sp_j differ by up to delta = K eps n_keys and the term by
    |tail| |log(1 - delta / (1 - sp_j))| ~ |tail| delta / (1 - sp_j): a slack GRADED by the conditioning, e.g.
    7e-5 absolute at the optimum of the trimmed C3 histogram (1 - sp_j = 1e-4, 380 keys, tail 11 192) against
    1e-9 |LL| = 0.1 -- there the term is simply checked.  Only where |1 - sp_j| <= delta -- the reference's term
    itself hangs on the last bits of an fsum: it flips between 0 (sp_j rounds to >= 1) and tail * log(k 2^-53) -- is
    the old absolute allowance of 40 |tail| (|log 2^-53| = 36.7) kept: the FLIP class.  sp_j > 1 + delta (the
    reference's 200-chunk normaliser makes some pmfs too large) is no coin toss: the term is 0 on
    both sides.  Returns (slack, class, unit) -- see _Slack."""
    if not tail or not math.isfinite(ll):
        return 0.0, None, 0.0
    eps = 2.0 ** -52
    delta = K_TAIL * eps * n_keys
    if sp - 1.0 > delta:
        return 0.0, None, 0.0
    gap = 1.0 - sp
    if gap <= delta:
        return 40.0 * abs(tail), "flip", 0.0
    slack = min(40.0, -math.log1p(-delta / gap)) * abs(tail)
    return slack, ("graded" if slack > 1e-10 * abs(ll) else None), abs(tail) * eps / gap


def _slack_of(tail, lls, sps, n_keys):
    out, classes, unit = _Slack(), [], []
    for ll, sp in zip(lls, sps):
        v, c, u = _tail_slack(tail, ll, sp, n_keys)
        out.append(v)
        classes.append(c)
        unit.append(u)
    out.classes, out.unit = classes, unit
    return out


def _tail_noise(om, points, lls, tail):
    """_tail_slack for every point of a case whose reference values come from the oracle: sp_j = fsum of the
    oracle's p_j (bit-equal to the reference's, tests/test_oracle_golden.py)."""
    if not tail:
        out = _Slack([0.0] * len(lls))
        out.classes, out.unit = [None] * len(lls), [0.0] * len(lls)
        return out
    sps = [math.fsum(om.compute_probabilities(*p).values()) if math.isfinite(ll) else 1.0 for p, ll in zip(points, lls)]
    return _slack_of(tail, lls, sps, len(om.hist))


_BOUNDS_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tail_noise_bounds.json")
_RECORD = os.environ.get("COVEST_RECORD_TAIL_BOUNDS")  # a path: write the counts there instead of checking them
_recorded = {}


def _slack_budget(name, slack):
    """The slack must stay the exception: how many points of case `name` fall into the FLIP class and how many into
    the GRADED one (a property of the reference's / the oracle's numbers alone, so the same on every box) may not
    exceed the counts committed in tests/golden/tail_noise_bounds.json.  A change that widened the criterion until
    every tail point fell under it would fail here.  COVEST_RECORD_TAIL_BOUNDS=<path> records instead (new cases)."""
    n_flip = sum(1 for c in slack.classes if c == "flip")
    n_graded = sum(1 for c in slack.classes if c == "graded")
    if _RECORD:
        _recorded[name] = {"flip": n_flip, "graded": n_graded, "points": len(slack)}
        with open(_RECORD, "w") as f:
            json.dump(_recorded, f, indent=0, sort_keys=True)
        return n_flip + n_graded
    with open(_BOUNDS_PATH) as f:
        bounds = json.load(f)
    assert name in bounds, "no committed tail-slack bound for case %r" % name
    b = bounds[name]
    assert n_flip <= b["flip"] and n_graded <= b["graded"], (
        "case %r: %d flip / %d graded points under the tail slack, committed bounds %d / %d of %d" % (
            name, n_flip, n_graded, b["flip"], b["graded"], b["points"]))
    return n_flip + n_graded


def test_device_present(hip_lib):
    assert hip_lib.covest_device_count() >= 1


@pytest.mark.parametrize("kind,fname", [("basic", "basic_ll.json"), ("repeats", "repeats_ll.json")])
def test_golden_loglikelihood(hip_lib, oracle, kind, fname):
    g = load_golden(fname)
    worst = 0.0
    n_slack = 0
    for case in g["cases"]:
        m = _gpu_model(kind, case)
        got = m.loglikelihood_points(np.array(case["points"]), kernel="direct")
        slack = _tail_noise(_oracle_model(oracle, kind, case), case["points"], case["ll"], case["tail"])
        n_slack += _slack_budget("golden %s #%d" % (kind, g["cases"].index(case)), slack)
        worst = max(worst, _check(got, case["ll"], "%s %s tail=%s S=%s" % (
            kind, case["hist"], case["tail"], case["max_error"]), slack=slack))
        if kind == "basic":  # the recurrence kernel (K-basic) on the same points: 8, 22 (= k + 1) and 5 error classes
            fast = m.loglikelihood_points(np.array(case["points"]), kernel="recur")
            worst = max(worst, _check(fast, case["ll"], "recur %s tail=%s" % (case["hist"], case["tail"]),
                                      slack=slack))
            assert np.array_equal(fast, m.loglikelihood_points(np.array(case["points"]), kernel="auto"),
                                  equal_nan=True)
        # scalar API == batch API, and the dict form of compute_loglikelihood_multi
        auto = m.loglikelihood_points(np.array(case["points"][:3]))
        p0 = case["points"][0]
        assert m.compute_loglikelihood(*p0) == auto[0]
        multi = m.compute_loglikelihood_multi([tuple(p) for p in case["points"][:3]])
        assert list(multi.values()) == [float(v) for v in auto]
        for d in case["detail"]:
            probs = m.compute_probabilities(*d["point"], clamp=True)
            _check([probs[j] for j, _ in d["p_j"]], [v for _, v in d["p_j"]], "p_j %r" % d["point"])
        m.close()
    print(kind, "worst rel err", worst, "points whose reference tail term is rounding noise:", n_slack)


@pytest.mark.parametrize("kernel", ["direct", "recur"])
def test_config1_full_grid_and_argmin(hip_lib, kernel):
    from covest_amd import DenseGrid
    g = load_golden("c1_grid.json")
    m = _gpu_model("basic", g)
    grid = DenseGrid(m, [g["c_axis"], g["e_axis"]])
    grid.evaluate(kernel=kernel)
    ll = grid.loglikelihoods()
    worst = _check(ll, g["ll"], "C1")
    val, arg = grid.argmin()
    assert arg == g["argmin_flat"]
    assert rel_err(val, g["min_negll"]) <= TOL
    assert grid.point(arg) == (g["c_axis"][24], g["e_axis"][9])
    # list API agrees with grid API bit for bit (same kernel, same point values)
    pts = np.array([(c, e) for c in g["c_axis"] for e in g["e_axis"]])
    assert np.array_equal(m.loglikelihood_points(pts, kernel=kernel), ll, equal_nan=True)
    print("C1", kernel, "worst rel err", worst)


@pytest.mark.parametrize("kernel", ["direct", "recur"])
def test_config2_sample_and_argmin(hip_lib, oracle, kernel):
    from covest_amd import DenseGrid
    g = load_golden("c2_sample.json")
    m = _gpu_model("basic", g)
    assert m.bins_evaluated == 367  # tail == 0: only the non-zero bins are evaluated
    got = m.loglikelihood_points(np.array(g["points"]), kernel=kernel)
    worst = _check(got, g["ll"], "C2 sample")
    assert sum(1 for v in g["ll"] if v == -math.inf) > 50  # the fixture does exercise -inf
    print("C2 sample", kernel, "worst rel err", worst)
    # full 1000 x 1000 grid: sampled points equal the list API, arg-min verified by the oracle
    cs = np.linspace(2000.0, 6000.0, 1000)
    es = np.linspace(0.001, 0.1, 1000)
    grid = DenseGrid(m, [cs, es])
    grid.evaluate(kernel=kernel)
    ll = grid.loglikelihoods()
    val, arg = grid.argmin()
    assert val == -ll[arg] and arg == int(np.nanargmin(np.where(np.isnan(ll), np.inf, -ll)))
    # ... and the arg-min judged by the REFERENCE (round 5; the oracle until then): tests/golden/c2_argmin.json holds its
    # values at the GPU's top 64 and the arg-min's axis neighbours (make_golden.py section c2argmin)
    worst = _verify_argmin_with_reference(load_golden("c2_argmin.json")["tail0"], grid, ll, arg)
    print("C2", kernel, "arg-min", arg, val, "confirmed by the reference; candidates' worst rel err", worst)


def _c2_grid(m):
    from covest_amd import DenseGrid
    return DenseGrid(m, [np.linspace(2000.0, 6000.0, 1000), np.linspace(0.001, 0.1, 1000)])


def test_config2_whole_grid_recur_against_direct(hip_lib):
    """Every one of C2's 10^6 points: K-basic (closed form / -inf by its bound / key-by-key walk, ll_basic.hip)
    against K-direct (one exp per term, itself pinned to the reference's values by the fixtures) at 1e-11, IEEE
    specials in the same places, the same arg-min.  What test_config3_* has had for K-factored since round 2."""
    g = load_golden("c2_sample.json")
    m = _gpu_model("basic", g)
    grid = _c2_grid(m)
    grid.evaluate(kernel="recur")
    assert grid.work()[2] == "ll_basic"
    fast = grid.loglikelihoods()
    best = grid.argmin()
    grid.evaluate(kernel="direct")
    assert grid.work()[2] == "ll_direct"
    ref = grid.loglikelihoods()
    assert np.array_equal(np.isneginf(fast), np.isneginf(ref)) and np.array_equal(np.isnan(fast), np.isnan(ref))
    fin = np.isfinite(ref)
    assert 400000 < int(fin.sum()) < 700000  # the grid does exercise both outcomes
    err = np.abs(fast[fin] - ref[fin]) / np.abs(ref[fin])
    worst = float(err.max())
    assert worst <= 1e-11, (worst, int(np.flatnonzero(fin)[int(err.argmax())]))
    assert grid.argmin()[1] == best[1]
    print("C2 whole grid, K-basic vs K-direct: worst rel err %.3g over %d finite points, %d -inf" % (
        worst, int(fin.sum()), int(np.isneginf(ref).sum())))
    grid.close()
    m.close()


def test_config2_class_boundaries_against_reference(hip_lib):
    """The reference's values where K-basic's closed form DECIDES (tests/golden/c2_classes.json: flat indices named
    per route by tools/dump_c2_classes.py on a GPU box with the diagnostic library -- >= 64 lanes each that walk key
    by key themselves, that are dragged into the walk by their wave, closed-form lanes closest to the clamp,
    -inf-by-bound lanes closest to the bound, and seeded lanes of both far from the boundaries)."""
    g = load_golden("c2_classes.json")
    m = _gpu_model("basic", g)
    grid = _c2_grid(m)
    grid.evaluate(kernel="recur")
    ll = grid.loglikelihoods()
    at = {i: k for k, i in enumerate(g["flat_index"])}
    for i, p in zip(g["flat_index"], g["points"]):
        assert np.allclose(grid.point(i), p, rtol=1e-15, atol=0)
    worst = {}
    for name, idx in g["classes"].items():
        assert len(idx) >= 64, (name, len(idx))
        want = [g["ll"][at[i]] for i in idx]
        worst[name] = (_check(ll[idx], want, "C2 class " + name), sum(1 for v in want if math.isfinite(v)))
    # the classes are what they say: the walk and the closed form end finite (mostly), the bound ends at -inf
    assert all(g["ll"][at[i]] == -math.inf for i in g["classes"]["neginf_far"])
    assert all(math.isfinite(g["ll"][at[i]]) for i in g["classes"]["closed_far"])
    # the list API (one lane per point of ANOTHER wave composition) agrees with the reference as well
    pts = np.array(g["points"])
    _check(m.loglikelihood_points(pts, kernel="recur"), g["ll"], "C2 classes, point list")
    _check(m.loglikelihood_points(pts, kernel="direct"), g["ll"], "C2 classes, K-direct")
    print("C2 class boundaries: worst rel err (finite values) per class", worst)
    grid.close()
    m.close()


def test_config2_trimmed_histogram_with_its_tail(hip_lib):
    """Config 2 on the histogram the reference's pipeline hands the model: H10k_basic trimmed by the reference's own
    get_trim / trim_hist (covest/histogram.py:105-134) -- K-basic WITH a tail at C2's scale (the closed form is off,
    every key enters sp_j).  Seeded grid points with the reference's LL and sp_j (graded slack, _tail_slack), the
    whole grid against K-direct on a seeded subset, the arg-min confirmed by the reference among the best 96 points
    and the axis neighbours of the best (tests/golden/c2_trim.json)."""
    from covest_amd import BasicModel
    g = load_golden("c2_trim.json")
    hist = load_hist(g["hist"])
    assert len(hist) == g["n_keys"] and g["tail"] > 0
    m = BasicModel(g["k"], g["r"], hist, g["tail"], max_error=g["max_error"])
    assert m.bins_evaluated == g["n_keys"]
    grid = _c2_grid(m)
    grid.evaluate(kernel="recur")
    assert grid.work()[2] == "ll_basic" and grid.total == 10 ** 6
    ll = grid.loglikelihoods()
    idx = np.array(g["flat_index"])
    want = np.array(g["ll"])
    assert int(np.isfinite(want).sum()) >= 256
    slack = _slack_of(g["tail"], g["ll"], g["sp"], g["n_keys"])
    n_flip = sum(1 for c in slack.classes if c == "flip")
    assert n_flip <= 0.05 * len(idx), n_flip
    _slack_budget("C2 trimmed sample", slack)
    worst = _check(ll[idx], want, "C2 trimmed (recur)", slack=slack)
    pts = np.array([grid.point(int(i)) for i in idx])
    worst = max(worst, _check(m.loglikelihood_points(pts, kernel="direct"), want, "C2 trimmed (direct)", slack=slack))
    val, arg = grid.argmin()
    assert val == -ll[arg] and arg == int(np.argmin(np.where(np.isnan(ll), np.inf, -ll)))
    fix = g["candidates"]
    cand = _argmin_candidates(grid, ll, arg, top=64)
    assert set(cand) <= set(fix["flat_index"]), "arg-min candidates outside the fixture: regenerate tests/golden/c2_trim.json"
    cslack = _slack_of(g["tail"], fix["ll"], fix["sp"], g["n_keys"])
    _slack_budget("C2 trimmed arg-min candidates", cslack)
    _check(ll[fix["flat_index"]], fix["ll"], "C2 trimmed arg-min candidates", slack=cslack)
    assert fix["reference_argmin_flat"] == arg
    assert rel_err(val, fix["reference_min_negll"]) <= TOL
    print("C2 trimmed: worst rel err", worst, "flip class", n_flip, "of", len(idx), "arg-min", arg, val)
    grid.close()
    m.close()


def _c3_axes():
    return [np.linspace(15.0, 30.0, 32), np.linspace(0.005, 0.08, 32), np.linspace(0.3, 0.95, 16),
            [0.5], np.linspace(0.05, 0.95, 16)]


def _argmin_candidates(grid, ll, arg, top=64):
    """the GPU's `top` best points and the 2 P axis neighbours of its arg-min (flat indices)."""
    negll = np.where(np.isnan(ll), np.inf, -ll)
    cand = set(np.argsort(negll, kind="stable")[:top].tolist())
    idx = np.unravel_index(arg, grid.shape)
    for d in range(len(grid.shape)):
        for step in (-1, 1):
            j = list(idx)
            j[d] += step
            if 0 <= j[d] < grid.shape[d]:
                cand.add(int(np.ravel_multi_index(j, grid.shape)))
    return sorted(cand)


def _verify_argmin_with_reference(fix, grid, ll, arg, slack=None):
    """The same procedure with the REFERENCE as the judge: tests/golden/c3_argmin.json holds the reference's own
    log-likelihoods at the candidates (make_golden.py section c3argmin, run in the build container on the
    indices tools/dump_c3_candidates.py wrote on a GPU box).  The candidate set of THIS run must be the
    fixture's, every value must agree to 1e-9, and the reference's winner under the scan of
    covest/grid.py:65-70 (strict <, first index) must be the GPU's arg-min."""
    cand = _argmin_candidates(grid, ll, arg)
    assert cand == fix["flat_index"], "arg-min candidates changed: regenerate tests/golden/c3_argmin.json"
    for i, p in zip(cand, fix["points"]):
        assert np.allclose(grid.point(i), p, rtol=1e-15, atol=0)
    worst = _check(ll[cand], fix["ll"], "arg-min candidates vs reference", slack=slack)
    best, winner = math.inf, -1
    for i, v in zip(cand, fix["ll"]):
        if -v < best:  # NaN never wins
            best, winner = -v, i
    assert winner == fix["reference_argmin_flat"] == arg
    assert rel_err(-float(ll[arg]), best) <= TOL
    return worst


@pytest.mark.parametrize("kernel", ["direct", "factored"])
def test_config3_sample_and_argmin(hip_lib, oracle, kernel):
    """Config 3 (the headline grid): >= 256 seeded points against the reference's values, and the arg-min
    confirmed by the reference itself."""
    from covest_amd import DenseGrid
    g = load_golden("c3_sample.json")
    assert len(g["points"]) >= 256
    m = _gpu_model("repeats", g)
    got = m.loglikelihood_points(np.array(g["points"]), kernel="direct")
    worst = _check(got, g["ll"], "C3 sample")
    print("C3 sample worst rel err", worst, "finite values", sum(1 for v in g["ll"] if math.isfinite(v)))
    grid = DenseGrid(m, _c3_axes())
    grid.evaluate(kernel=kernel)
    assert grid.work()[2] == "ll_" + kernel
    ll = grid.loglikelihoods()
    # the fixture's flat indices address the same points
    for i, p, want in zip(g["flat_index"], g["points"], g["ll"]):
        assert np.allclose(grid.point(i), p, rtol=1e-15, atol=0)
        assert rel_err(float(ll[i]), want) <= TOL
    val, arg = grid.argmin()
    assert val == -ll[arg] and arg == int(np.argmin(np.where(np.isnan(ll), np.inf, -ll)))
    worst = _verify_argmin_with_reference(load_golden("c3_argmin.json")["tail0"], grid, ll, arg)
    print("C3", kernel, "arg-min", arg, val, "confirmed by the reference; candidates' worst rel err", worst)
    if kernel == "factored":  # whole grid against the direct kernel (itself pinned to the fixture above)
        ref = DenseGrid(m, _c3_axes())
        ref.evaluate(kernel="direct")
        worst = _check(ll, ref.loglikelihoods(), "C3 factored vs direct", tol=1e-11)
        assert ref.argmin() == (val, arg) or ref.argmin()[1] == arg
        print("C3 factored vs direct worst rel err", worst)


def test_config3_with_tail(hip_lib, oracle):
    """Config 3 with tail = 1000 -- what a trimmed real histogram has (covest/histogram.py:125-134): all 10 000
    keys enter sp_j (covest/models.py:103-104).  48 seeded points and the arg-min candidates against the
    reference's values, K-direct on the points and K-factored on the whole grid."""
    from covest_amd import DenseGrid
    g = load_golden("c3_tail_sample.json")
    assert g["tail"] == 1000
    m = _gpu_model("repeats", g)
    assert m.bins_evaluated == 10000
    # where the reference's own tail term is rounding noise (1 - sp_j of a few ulp) the comparison carries the
    # explicit slack, decided from the sp_j the REFERENCE saw (recorded in the fixture)
    slack = _slack_of(1000, g["ll"], [1.0 if sp is None else sp for sp in g["sp"]], 10000)
    _slack_budget("C3 tail sample", slack)
    got = m.loglikelihood_points(np.array(g["points"]), kernel="direct")
    worst = _check(got, g["ll"], "C3 tail sample (direct)", slack=slack)
    grid = DenseGrid(m, _c3_axes())
    grid.evaluate(kernel="factored")
    assert grid.work()[2] == "ll_factored"
    ll = grid.loglikelihoods()
    worst = max(worst, _check(ll[g["flat_index"]], g["ll"], "C3 tail sample (factored)", slack=slack))
    val, arg = grid.argmin()
    assert val == -ll[arg] and arg == int(np.argmin(np.where(np.isnan(ll), np.inf, -ll)))
    fix = load_golden("c3_argmin.json")["tail1000"]
    aslack = _slack_of(1000, fix["ll"], [1.0 if sp is None else sp for sp in fix["sp"]], 10000)
    _slack_budget("C3 tail arg-min candidates", aslack)
    _verify_argmin_with_reference(fix, grid, ll, arg, slack=aslack)
    print("C3 tail=1000 worst rel err", worst, "arg-min", arg, val)


def test_config3_trimmed_histogram_with_its_tail(hip_lib):
    """Config 3 on the histogram the reference's pipeline hands the model: H10k_rep trimmed by the reference's own
    get_trim / trim_hist (covest/histogram.py:105-134) -- 380 keys, tail = 11 192, so that at the optimum
    1 - sp_j ~ 1e-4 and the tail term tail * log(1 - sp_j) (covest/models.py:103-104) is WELL CONDITIONED.  4 096
    seeded grid points with the reference's LL and sp_j: K-factored on the whole grid and K-direct on every eighth
    point, graded slack (_tail_slack), at most 5 % of the points in the flip class; the arg-min confirmed by the
    reference among the best 96 points of the grid and the axis neighbours of the best (tests/golden/c3_trim.json)."""
    from covest_amd import DenseGrid, RepeatsModel
    g = load_golden("c3_trim.json")
    hist = load_hist(g["hist"])
    assert len(hist) == g["n_keys"] == 380 and g["tail"] == 11192
    m = RepeatsModel(g["k"], g["r"], hist, g["tail"], max_error=g["max_error"])
    assert m.bins_evaluated == 380
    axes = [g["axes"][0], g["axes"][1], g["axes"][2], [g["q2"]], g["axes"][3]]
    grid = DenseGrid(m, axes)
    grid.evaluate(kernel="factored")
    assert grid.work()[2] == "ll_factored" and grid.total == 262144
    ll = grid.loglikelihoods()
    idx = np.array(g["flat_index"])
    want = np.array(g["ll"])
    assert int(np.isfinite(want).sum()) >= 4000
    slack = _slack_of(g["tail"], g["ll"], g["sp"], g["n_keys"])
    n_flip = sum(1 for c in slack.classes if c == "flip")
    assert n_flip <= 0.05 * len(idx), n_flip
    _slack_budget("C3 trimmed sample", slack)
    worst = _check(ll[idx], want, "C3 trimmed (factored)", slack=slack)
    sub = slice(0, None, 8)
    pts = np.array([grid.point(int(i)) for i in idx[sub]])
    dslack = _Slack(slack[sub])
    dslack.classes, dslack.unit = slack.classes[sub], slack.unit[sub]
    worst = max(worst, _check(m.loglikelihood_points(pts, kernel="direct"), want[sub], "C3 trimmed (direct)", slack=dslack))
    # the arg-min: the GPU's candidates must be among those the reference was asked about, agree there, and the
    # reference's winner under the scan of covest/grid.py:65-70 must be the GPU's
    val, arg = grid.argmin()
    assert val == -ll[arg] and arg == int(np.argmin(np.where(np.isnan(ll), np.inf, -ll)))
    fix = g["candidates"]
    known = {i: k for k, i in enumerate(fix["flat_index"])}
    cand = _argmin_candidates(grid, ll, arg, top=64)
    assert set(cand) <= set(known), "arg-min candidates outside the fixture: regenerate tests/golden/c3_trim.json"
    cslack = _slack_of(g["tail"], fix["ll"], fix["sp"], g["n_keys"])
    _slack_budget("C3 trimmed arg-min candidates", cslack)
    _check(ll[fix["flat_index"]], fix["ll"], "C3 trimmed arg-min candidates", slack=cslack)
    assert fix["reference_argmin_flat"] == arg
    assert rel_err(val, fix["reference_min_negll"]) <= TOL
    print("C3 trimmed: worst rel err", worst, "flip class", n_flip, "of", len(idx), "arg-min", arg, val)
    grid.close()
    m.close()


def test_config4_c3_in_eight_blocks(hip_lib):
    """Config 4's workload on one device: the C3 grid cut into 8 contiguous flat-index blocks balanced by
    sum(T - 1), each evaluated on its own; the values are those of the whole grid bit for bit and
    the winner of the ranks' (min, index) pairs under distributed_argmin's scan is the whole grid's."""
    from covest_amd import DenseGrid, RepeatsModel
    from covest_amd.grid import partition_flat_range, repeats_cost_weights, scan_min_pairs
    m = RepeatsModel(21, 100, load_hist("H10k_rep"), 0, max_error=8)
    axes = _c3_axes()
    whole = DenseGrid(m, axes)
    whole.evaluate()
    assert whole.work()[2] == "ll_factored"
    ll = whole.loglikelihoods()
    best = whole.argmin()
    w = repeats_cost_weights(m, axes)
    bounds = partition_flat_range(whole.total, 8, w)
    cost = [float(np.sum(np.tile(w, whole.total // len(w))[a:b])) for a, b in zip(bounds[:-1], bounds[1:])]
    assert max(cost) <= 1.02 * (sum(cost) / 8)  # balanced by work, not by point count
    parts, pairs = [], []
    for r in range(8):
        g = DenseGrid(m, axes, (bounds[r], bounds[r + 1]))
        g.evaluate()
        assert g.work()[2] == "ll_factored"
        parts.append(g.loglikelihoods())
        pairs.append(g.argmin())
        g.close()
    assert np.array_equal(np.concatenate(parts), ll, equal_nan=True)
    assert scan_min_pairs(pairs) == best
    assert best[1] == 165489


def test_strong_scaling_grid_sampled_against_direct(hip_lib, oracle):
    """The fixed grid of `bench.py --scaling strong` (c128 x e128 x 16 x 1 x 16 = 4.2 M points): K-factored on the
    whole grid against K-direct on 4 000 seeded points and against the oracle on 6 (its faithful O(j) product
    manages half a point a second on this histogram); cut into 8 blocks balanced by
    sum(T - 1), every block reproduces its slice bit for bit and the scan of the blocks' pairs is the whole grid's
    arg-min (what N ranks compute, on one device)."""
    import bench
    from covest_amd import DenseGrid, RepeatsModel
    from covest_amd.grid import partition_flat_range, repeats_cost_weights, scan_min_pairs
    kind, hname, axes = bench.workload("c3", 1, "strong")
    hist = load_hist(hname)
    m = RepeatsModel(21, 100, hist, 0, max_error=8)
    whole = DenseGrid(m, axes)
    whole.evaluate()
    assert whole.work()[2] == "ll_factored" and whole.total == 128 * 128 * 256
    ll = whole.loglikelihoods()
    best = whole.argmin()
    assert best[1] == int(np.nanargmin(-ll)) and best[0] == float(-ll[best[1]])
    rng = np.random.default_rng(20240905)
    sel = np.sort(rng.choice(whole.total, size=4000, replace=False))
    pts = np.array([whole.point(i) for i in sel])
    _check(ll[sel], m.loglikelihood_points(pts, kernel="direct"), "strong grid vs K-direct", tol=1e-10)
    om = oracle.OracleModel("repeats", 21, 100, hist, 0, max_error=8)
    few = sel[rng.choice(len(sel), size=6, replace=False)]
    _check(ll[few], om.compute_loglikelihood_many(np.array([whole.point(i) for i in few]), n_threads=16),
           "strong grid vs oracle")
    bounds = partition_flat_range(whole.total, 8, repeats_cost_weights(m, axes))
    pairs = []
    for r in range(8):
        g = DenseGrid(m, axes, (bounds[r], bounds[r + 1]))
        g.evaluate()
        assert np.array_equal(g.loglikelihoods(), ll[bounds[r]:bounds[r + 1]], equal_nan=True), r
        pairs.append(g.argmin())
        g.close()
    assert scan_min_pairs(pairs) == best
    whole.close()
    m.close()


@pytest.mark.parametrize("tail", [0, 1000])
@pytest.mark.parametrize("hname", ["sim_c10_e0.05", "sim_c10_e0.05_sparse", "sim_c10_e0"])
def test_factored_small_histograms(hip_lib, oracle, hname, tail):
    """K-factored on the reference's own test histograms: every point of a dense
    5-D grid against the oracle, whole and in ragged flat-index blocks."""
    from covest_amd import DenseGrid, RepeatsModel
    hist = load_hist(hname)
    m = RepeatsModel(21, 100, hist, tail, max_error=8)
    om = oracle.OracleModel("repeats", 21, 100, hist, tail, max_error=8)
    axes = [np.array([6.0, 10.0, 14.5]), np.array([0.0, 0.02, 0.05, 0.3]), np.array([0.2, 0.5, 0.8, 1.0]),
            np.array([0.0, 0.4, 1.0]), np.array([0.0, 0.05, 0.3, 0.7, 1.0])]
    grid = DenseGrid(m, axes)
    grid.evaluate(kernel="factored")
    ll = grid.loglikelihoods()
    pts = np.array([grid.point(i) for i in range(grid.total)])
    ref = om.compute_loglikelihood_many(pts, n_threads=16)
    slack = _tail_noise(om, pts, ref, tail)
    _slack_budget("factored small %s tail=%d" % (hname, tail), slack)
    worst = _check(ll, ref, "factored %s tail=%d" % (hname, tail), slack=slack)
    k, best = oracle.first_min(-ref)
    val, arg = grid.argmin()
    assert arg == k or ll[arg] == ll[k]
    # ragged blocks: cuts inside a (c, e) row
    cuts = [0, 7, 61, 200, grid.total - 1, grid.total]
    parts = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        blk = DenseGrid(m, axes, (a, b))
        blk.evaluate(kernel="factored")
        parts.append(blk.loglikelihoods())
    assert np.array_equal(np.concatenate(parts), ll, equal_nan=True)
    print("factored", hname, tail, "worst rel err", worst)


def test_factored_plan_shapes(hip_lib, oracle):
    """The launch shapes of K-factored that the benchmark grid does not reach: several
    workgroups per (c, e) (more weight vectors than one workgroup carries), threshold_o
    above 320 (single-buffered G) and the 256-thread variant -- each against K-direct on
    the whole grid and against the oracle on a sample."""
    from covest_amd import DenseGrid, RepeatsModel
    rng = np.random.default_rng(7)
    hist = {j: int(v) for j, v in zip(range(1, 451), rng.integers(0, 2000, size=450))}
    hist[700] = 3
    cases = [
        # many weight vectors: 7*6*16 = 672 -> 42 q-tiles -> 84 units > 48 per workgroup
        ("q-blocks", 0, [np.array([20.0, 31.0]), np.array([0.01, 0.04]), np.linspace(0.3, 0.9, 7),
                         np.linspace(0.0, 1.0, 6), np.linspace(0.08, 0.9, 16)]),
        # threshold_o up to ~450: LD > 320, one LDS buffer
        ("single buffer", 5, [np.array([12.0]), np.array([0.02, 0.1]), np.linspace(0.3, 0.9, 4),
                              np.array([0.2, 0.7]), np.array([0.03, 0.04, 0.2, 0.5, 0.9])]),
        # few vectors, small T: the 256-thread workgroup
        ("256 threads", 0, [np.array([5.0, 9.0]), np.array([0.03]), np.linspace(0.4, 0.9, 4),
                            np.array([0.5]), np.linspace(0.3, 0.9, 8)]),
        # what optimize_grid walks into (covest/grid.py:23-26: q down to 0.01): threshold_o up to ~1500, far beyond a
        # workgroup's 512 lanes -- the long weight vectors go chunk by chunk of copy numbers, the short ones
        # (q = 0.2, 0.6) stay on the one-launch path, in the same grid
        ("chunks of copy numbers", 3, [np.array([0.8, 1.5]), np.array([0.02]), np.array([0.5, 0.9]),
                                       np.array([0.3]), np.array([0.0105, 0.012, 0.02, 0.05, 0.2, 0.6])]),
        # ... and a grid of long vectors only
        ("long vectors only", 0, [np.array([1.1]), np.array([0.01, 0.04]), np.array([0.6]), np.array([0.5]),
                                  np.array([0.0103, 0.0107, 0.011])]),
        # max_error = k + 1 = 22, the default of a model built directly (covest/models.py:28-31): a copy number's
        # error classes are dealt to three lanes
        ("22 error classes", 7, [np.array([8.0, 20.0]), np.array([0.02, 0.2, 0.45]), np.linspace(0.3, 0.9, 3),
                                 np.array([0.2, 0.7]), np.array([0.06, 0.2, 0.5, 0.9]), 22]),
        # ... with long vectors on top (chunks of 168 copy numbers), and 12 classes (two lanes)
        ("22 classes, chunks", 0, [np.array([1.2]), np.array([0.05]), np.array([0.7]), np.array([0.5]),
                                   np.array([0.02, 0.03, 0.3]), 22]),
        ("12 error classes", 0, [np.array([6.0]), np.array([0.1, 0.4]), np.array([0.5, 0.8]), np.array([0.5]),
                                 np.linspace(0.1, 0.9, 5), 12]),
        ("5 error classes", 2, [np.array([6.0, 30.0]), np.array([0.1]), np.array([0.5, 0.8]), np.array([0.5]),
                                np.linspace(0.1, 0.9, 5), 5]),
    ]
    for case in cases:
        name, tail, axes = case[0], case[1], case[2]
        S = 8
        if not isinstance(axes[-1], np.ndarray):
            S, axes = axes[-1], axes[:-1]
        m = RepeatsModel(21, 100, hist, tail, max_error=S)
        om = oracle.OracleModel("repeats", 21, 100, hist, tail, max_error=S)
        fac = DenseGrid(m, axes)
        fac.evaluate(kernel="factored")
        assert fac.work()[2] == "ll_factored"
        ll = fac.loglikelihoods()
        ref = DenseGrid(m, axes)
        ref.evaluate(kernel="direct")
        worst = _check(ll, ref.loglikelihoods(), name + " vs direct", tol=1e-10)
        assert fac.argmin()[1] == ref.argmin()[1]
        sel = rng.choice(fac.total, size=min(24, fac.total), replace=False)
        pts = np.array([fac.point(i) for i in sel])
        want = om.compute_loglikelihood_many(pts, n_threads=16)
        slack = _tail_noise(om, pts, want, tail)
        _slack_budget("plan shape " + name, slack)
        _check(ll[sel], want, name + " vs oracle", slack=slack)
        print("factored plan shape:", name, "points", fac.total, "worst rel err vs direct", worst)


def test_shared_steps_layouts(hip_lib, oracle):
    """K-factored's shared steps (tiles.h): the weight vectors of one q fill whole q-tiles of 16 -- exactly (16), with
    padding columns inside the order (12 -> 16, 27 -> 32, 36 -> 48), or not at all (9: too much padding, the plain
    order).  Each layout against K-direct on the whole grid and against the oracle on a sample; the grid cut into
    blocks gives bit-identical values.  (The switch that turned the sharing off for A/B runs exists in diagnostic
    builds only: the shipped library has no knob that changes values, tests/test_capi_symbols.py.)"""
    from covest_amd import DenseGrid, RepeatsModel
    rng = np.random.default_rng(11)
    hist = {j: int(v) for j, v in zip(range(1, 301), rng.integers(1, 3000, size=300))}
    q_short = np.array([0.04, 0.07, 0.15, 0.33, 0.6, 0.97])  # threshold_o from ~350 down to 8
    # ... and with weight vectors beyond a workgroup's 512 lanes in the same grid (threshold_o ~1500 at q = 0.0105):
    # their q-tiles go chunk by chunk without shared steps, the others share
    q_long = np.array([0.0105, 0.02, 0.08, 0.5])
    for n1, n2, tail, q_axis in ((4, 4, 0, q_short), (4, 3, 0, q_short), (9, 3, 7, q_short), (6, 6, 0, q_short),
                                 (3, 3, 0, q_short), (4, 4, 0, q_long), (4, 3, 5, q_long)):
        axes = [np.array([14.0, 27.0]) if q_axis is q_short else np.array([1.2, 2.5]), np.array([0.01, 0.06]),
                np.linspace(0.3, 0.95, n1), np.linspace(0.05, 0.9, n2) if n2 > 1 else np.array([0.5]), q_axis]
        m = RepeatsModel(21, 100, hist, tail, max_error=8)
        om = oracle.OracleModel("repeats", 21, 100, hist, tail, max_error=8)
        name = "shared steps %dx%d tail %d%s" % (n1, n2, tail, "" if q_axis is q_short else " long")
        fac = DenseGrid(m, axes)
        fac.evaluate(kernel="factored")
        ll = fac.loglikelihoods()
        ref = DenseGrid(m, axes)
        ref.evaluate(kernel="direct")
        _check(ll, ref.loglikelihoods(), name + " vs direct", tol=1e-10)
        assert fac.argmin()[1] == ref.argmin()[1]
        # two blocks of the flat range: the same plan, the same bits
        half = fac.total // 2 + 3
        parts = []
        for lo, hi in ((0, half), (half, fac.total)):
            blk = DenseGrid(m, axes, flat_range=(lo, hi))
            blk.evaluate(kernel="factored")
            parts.append(blk.loglikelihoods())
            blk.close()
        assert np.array_equal(np.concatenate(parts), ll, equal_nan=True), name
        sel = rng.choice(fac.total, size=16, replace=False)
        pts = np.array([fac.point(i) for i in sel])
        want = om.compute_loglikelihood_many(pts, n_threads=16)
        slack = _tail_noise(om, pts, want, tail)
        _slack_budget(name, slack)
        _check(ll[sel], want, name + " vs oracle", slack=slack)
        for g in (fac, ref):
            g.close()
        m.close()


def test_threshold_fixture_through_capi(hip_lib):
    from covest_amd import RepeatsModel
    g = load_golden("threshold_o.json")
    rows = np.array(g["rows"])
    for hist_max in (15, 256, 10000):
        sel = rows[rows[:, 0] == hist_max]
        m = RepeatsModel(21, 100, {hist_max: 1, 1: 1}, 0, max_error=8)
        got = m.get_hist_threshold_values(sel[:, 1:4])
        assert np.array_equal(got, sel[:, 4].astype(np.int32))


def test_edge_cases(hip_lib, oracle):
    from covest_amd import BasicModel, RepeatsModel
    # empty histogram: LL = 0 (covest/models.py:103-107 on an empty dict)
    m = BasicModel(21, 100, {}, 0, max_error=8)
    assert m.compute_loglikelihood(10.0, 0.05) == 0.0
    assert m.compute_loglikelihood_multi([]) == {}
    # all-zero counts with a tail: only the tail term
    hist = {j: 0 for j in range(1, 40)}
    m = BasicModel(21, 100, hist, 500, max_error=8)
    om = oracle.OracleModel("basic", 21, 100, hist, 500, max_error=8)
    assert rel_err(m.compute_loglikelihood(10.0, 0.05), om.compute_loglikelihood(10.0, 0.05)) <= TOL
    # ragged / unordered / sparse keys, counts above 2^31, S not dividing 64
    hist = {40: 3, 2: 5_000_000_000, 7: 0, 1000: 1, 1: 12}
    for S in (1, 5, 22):
        m = BasicModel(21, 100, hist, 7, max_error=S)
        om = oracle.OracleModel("basic", 21, 100, hist, 7, max_error=S)
        pts = np.array([(10.0, 0.05), (300.0, 0.2), (0.01, 0.5), (40.0, 0.0)])
        _check(m.loglikelihood_points(pts), om.compute_loglikelihood_many(pts), "ragged S=%d" % S)
        r = RepeatsModel(21, 100, hist, 7, max_error=S)
        orr = oracle.OracleModel("repeats", 21, 100, hist, 7, max_error=S)
        pts5 = np.array([(10.0, 0.05, 0.7, 0.5, 0.5), (30.0, 0.1, 0.3, 0.0, 0.9),
                         (5.0, 0.0, 1.0, 0.5, 0.5), (8.0, 0.3, 0.5, 1.0, 0.01)])
        _check(r.loglikelihood_points(pts5), orr.compute_loglikelihood_many(pts5), "ragged rep S=%d" % S)
    # the recurrence kernel on ragged keys: gaps bridged (2..7), gaps re-anchored (40 -> 1000), tail and no tail
    for tail in (0, 7):
        m = BasicModel(21, 100, hist, tail, max_error=8)
        om = oracle.OracleModel("basic", 21, 100, hist, tail, max_error=8)
        pts = np.array([(10.0, 0.05), (300.0, 0.2), (0.01, 0.5), (40.0, 0.0), (1200.0, 0.01), (900.0, 0.3)])
        ref = om.compute_loglikelihood_many(pts)
        slack = _tail_noise(om, pts, ref, tail)
        _slack_budget("ragged recur tail=%d" % tail, slack)
        _check(m.loglikelihood_points(pts, kernel="recur"), ref, "ragged recur tail=%d" % tail, slack=slack)
    # NaN parameters poison the result, as in the reference
    m = BasicModel(21, 100, {1: 5, 2: 3}, 0, max_error=8)
    assert math.isnan(m.compute_loglikelihood(float("nan"), 0.05))
    assert math.isnan(m.loglikelihood_points([[float("nan"), 0.05]], kernel="direct")[0])
    # a non-zero bin with p_j == 0 -> -inf
    m = BasicModel(21, 100, {5000: 1, 1: 10}, 0, max_error=8)
    assert m.compute_loglikelihood(1.0, 0.01) == -math.inf


def test_streams_that_start_below_the_window(hip_lib, oracle):
    """Regression (found by the C3 fixture with a tail): a stream whose rate o * lambda_s is about 1083 ... 1119 starts
    a run of keys with a term of e^-1100 -- outside the recurrence's window.  Round 1 anchored it there all the same,
    as a SUBNORMAL double (a handful of significant bits) that every later key inherited: p_j wrong by up to a per
    cent exactly where such a stream has its mass.  A histogram counted around key 1100 and rates swept through that
    band, basic model (the stream IS the likelihood) and repeats model, against the oracle."""
    from covest_amd import BasicModel, DenseGrid, RepeatsModel
    rng = np.random.default_rng(11)
    hist = {j: int(v) for j, v in zip(range(1, 1301), np.zeros(1300))}
    for j in range(950, 1251):
        hist[j] = int(1000 * math.exp(-0.5 * ((j - 1100) / 33.0) ** 2) * 40) + int(rng.integers(0, 3))
    hist = {j: v for j, v in hist.items() if v or j % 7 == 0}  # (some zero-count keys stay: they matter with a tail)
    # lambda_0 = c * 0.8 * (1 - e)^21: c such that it sweeps 1070 ... 1130 at e = 0.01
    cs = np.linspace(1070.0, 1130.0, 41) / (0.8 * 0.99 ** 21)
    for tail in (0, 500):
        m = BasicModel(21, 100, hist, tail, max_error=8)
        om = oracle.OracleModel("basic", 21, 100, hist, tail, max_error=8)
        pts = np.array([(c, 0.01) for c in cs])
        ref = om.compute_loglikelihood_many(pts, n_threads=16)
        assert np.all(np.isfinite(ref))
        slack = _tail_noise(om, pts, ref, tail)
        _slack_budget("window start basic tail=%d" % tail, slack)
        for kernel in ("direct", "recur"):
            _check(m.loglikelihood_points(pts, kernel=kernel), ref, "window start basic %s tail=%d" % (kernel, tail),
                   slack=slack)
        # repeats: copy numbers 1 .. 8 of a rate eight times smaller put o * lambda_0 = 1100 at o = 8
        rm = RepeatsModel(21, 100, hist, tail, max_error=8)
        orm = oracle.OracleModel("repeats", 21, 100, hist, tail, max_error=8)
        axes = [cs[::4] / 8.0, [0.01], [0.4], [0.3], [0.05, 0.3]]
        grid = DenseGrid(rm, axes)
        gp = np.array([grid.point(i) for i in range(grid.total)])
        gref = orm.compute_loglikelihood_many(gp, n_threads=16)
        gslack = _tail_noise(orm, gp, gref, tail)
        _slack_budget("window start repeats tail=%d" % tail, gslack)
        for kernel in ("direct", "factored"):
            grid.evaluate(kernel=kernel)
            _check(grid.loglikelihoods(), gref, "window start repeats %s tail=%d" % (kernel, tail), slack=gslack)


def test_subnormal_pj_goes_to_the_strict_kernel(hip_lib, oracle):
    """A key that was observed 6000 times and to which the model gives a SUBNORMAL probability (1e-323 ... 1e-309:
    one or a few steps of the 4.9e-324 grid): the reference's value there hangs on the rounding of every single
    term (c_src/covest_poissonmodule.c:32, covest/models.py:93).  The recurrence kernels detect it and hand the
    point to the term-by-term kernel -- in the arg-min pass for grids, in covest_eval_points for lists -- so
    every kernel name must deliver the plain 1e-9 (round 1 excused this case with a slack)."""
    from covest_amd import BasicModel, DenseGrid, RepeatsModel
    hist = {1: 1000, 2: 500, 150: 6000}
    cs = np.linspace(0.55, 0.80, 51)
    om = oracle.OracleModel("basic", 21, 100, hist, 0, max_error=8)
    pts = np.array([(c, 0.01) for c in cs])
    ref = om.compute_loglikelihood_many(pts, n_threads=16)
    sub = [0.0 < om.compute_probabilities(*p)[150] < 2.2250738585072014e-308 for p in pts]
    assert sum(sub) >= 20 and sum(1 for v in ref if v == -math.inf) >= 5 and not all(sub)
    m = BasicModel(21, 100, hist, 0, max_error=8)
    for kernel in ("direct", "recur"):
        _check(m.loglikelihood_points(pts, kernel=kernel), ref, "subnormal basic list " + kernel)
        grid = DenseGrid(m, [cs, [0.01]])
        grid.evaluate(kernel=kernel)
        ll = grid.loglikelihoods()
        _check(ll, ref, "subnormal basic grid " + kernel)
        assert grid.argmin()[1] == oracle.first_min(-ref)[0]
        grid.close()
    # repeats model: the same key through K-factored (dense grid, point list) and K-direct
    rm = RepeatsModel(21, 100, hist, 0, max_error=8)
    orm = oracle.OracleModel("repeats", 21, 100, hist, 0, max_error=8)
    axes = [np.exp(np.linspace(np.log(0.03), np.log(0.8), 60)), [0.01], [0.7, 0.9], [0.5], [0.6, 0.95, 1.0]]
    grid = DenseGrid(rm, axes)
    gp = np.array([grid.point(i) for i in range(grid.total)])
    gref = orm.compute_loglikelihood_many(gp, n_threads=16)
    n_sub = sum(0.0 < orm.compute_probabilities(*p)[150] < 2.2250738585072014e-308 for p in gp)
    assert n_sub >= 20  # (27 of the 360 points when this was written)
    for kernel in ("direct", "factored"):
        grid.evaluate(kernel=kernel)
        _check(grid.loglikelihoods(), gref, "subnormal repeats grid " + kernel)
        assert grid.argmin()[1] == oracle.first_min(-gref)[0]
        _check(rm.loglikelihood_points(gp, kernel=kernel), gref, "subnormal repeats list " + kernel)
    print("subnormal p_j: %d basic points, >= %d repeats points re-evaluated by the strict kernel" % (sum(sub), n_sub))


def test_round4_shortcuts_on_awkward_shapes(hip_lib, oracle):
    """The work the round-4 kernels leave OUT must never change a value.  K-factored takes the LAST key tile first (a
    unit whose weight vectors all end at -inf dies in the first interval, ll_factored.hip `last_first`); K-basic lets a
    wave whose points are all -inf by a sufficient bound at the last counted key leave before its prologue
    (ll_basic.hip); the strict kernel of the basic model takes 64 / S points a wave (argmin.hip).  Histograms chosen
    for what those shortcuts could trip over -- one partial tile, a full and a partial one, a gap between two runs with
    the last tile a run of its own, a lone far key that dooms most of the grid, a subnormal key with 8, 16 and 24 error
    classes -- each on a grid that holds finite points, -inf points and the border between them: the recurrence kernel
    against K-direct (term by term, itself pinned to the reference by the fixtures) at 1e-11 with IEEE specials in the
    same places and the same arg-min, and against the oracle at 1e-9 on a sample."""
    from covest_amd import BasicModel, DenseGrid, RepeatsModel

    def falling(keys, top):
        return {int(j): max(1, int(top * math.exp(-0.07 * i))) for i, j in enumerate(keys)}

    hists = {
        "one partial tile": falling(range(1, 21), 5000),
        "a full and a partial tile": falling(range(1, 46), 20000),
        "two runs": {**falling(range(1, 41), 9000), **{j: 3 + (j % 4) for j in range(300, 331)}},
        "a far key": {**falling(range(1, 101), 50000), 2000: 3},
    }
    rng = np.random.default_rng(5)
    for name, hist in hists.items():
        top = max(hist)
        # ---- repeats model: K-factored against K-direct on the whole grid
        rm = RepeatsModel(21, 100, hist, 0, max_error=8)
        orm = oracle.OracleModel("repeats", 21, 100, hist, 0, max_error=8)
        axes = [np.exp(np.linspace(np.log(0.4), np.log(max(4.0, 0.9 * top)), 14)), [0.004, 0.03, 0.2],
                np.linspace(0.35, 0.95, 4), [0.5], [0.05, 0.3, 0.6, 0.95]]
        grid = DenseGrid(rm, axes)
        grid.evaluate(kernel="factored")
        assert grid.work()[2] == "ll_factored"
        fast, best = grid.loglikelihoods(), grid.argmin()
        grid.evaluate(kernel="direct")
        ref = grid.loglikelihoods()
        assert np.array_equal(np.isneginf(fast), np.isneginf(ref)) and np.array_equal(np.isnan(fast), np.isnan(ref)), name
        fin = np.isfinite(ref)
        assert fin.any(), name
        if name in ("two runs", "a far key"):
            assert (~fin).any(), name  # the grid does hold doomed points
        err = np.abs(fast[fin] - ref[fin]) / np.abs(ref[fin])
        assert float(err.max()) <= 1e-11, (name, float(err.max()), int(np.flatnonzero(fin)[int(err.argmax())]))
        assert grid.argmin()[1] == best[1], name
        some = rng.choice(grid.total, size=24, replace=False)
        pts = np.array([grid.point(int(i)) for i in some])
        _check(fast[some], orm.compute_loglikelihood_many(pts, n_threads=16), "round-4 shapes, repeats, " + name)
        grid.close()
        rm.close()
        # ---- basic model, 8 / 16 / 24 error classes: K-basic (+ the packed strict kernel) against K-direct
        for max_error in (8, 16, 24):
            m = BasicModel(21, 100, hist, 0, max_error=max_error)
            cs = np.exp(np.linspace(np.log(0.3), np.log(3.0 * top), 640))
            g = DenseGrid(m, [cs, [0.002, 0.05, 0.3, 0.5]])
            g.evaluate(kernel="recur")
            assert g.work()[2] == "ll_basic"
            fast, best = g.loglikelihoods(), g.argmin()
            g.evaluate(kernel="direct")
            ref = g.loglikelihoods()
            assert np.array_equal(np.isneginf(fast), np.isneginf(ref)) and np.array_equal(np.isnan(fast), np.isnan(ref)), (name, max_error)
            fin = np.isfinite(ref)
            assert fin.any() and ((~fin).any() or name == "one partial tile"), (name, max_error)
            err = np.abs(fast[fin] - ref[fin]) / np.abs(ref[fin])
            assert float(err.max()) <= 1e-11, (name, max_error, float(err.max()))
            assert g.argmin()[1] == best[1], (name, max_error)
            if max_error == 8:
                om = oracle.OracleModel("basic", 21, 100, hist, 0, max_error=8)
                some = rng.choice(g.total, size=48, replace=False)
                pts = np.array([g.point(int(i)) for i in some])
                _check(fast[some], om.compute_loglikelihood_many(pts, n_threads=16), "round-4 shapes, basic, " + name)
            g.close()
            m.close()
    # the strict kernel of the basic model with several points a wave: a key with a SUBNORMAL p_j (as in
    # test_subnormal_pj_goes_to_the_strict_kernel) under 8, 16, 24 and (k = 31) 32 error classes -- one instantiation of
    # the packed kernel each --, list and grid
    hist = {1: 1000, 2: 500, 150: 6000, 151: 40, 153: 7}
    cs = np.linspace(0.50, 0.85, 141)
    for k, max_error in ((21, 8), (21, 16), (21, 24), (31, 32)):
        m = BasicModel(k, 100, hist, 0, max_error=max_error)
        # (k = 31: the same error-free rates c (r - k + 1) / r (1 - e)^k as k = 21 has on `cs`)
        g = DenseGrid(m, [cs * (1.265 if k == 31 else 1.0), [0.01, 0.02]])
        g.evaluate(kernel="recur")
        fast = g.loglikelihoods()
        g.evaluate(kernel="direct")
        ref = g.loglikelihoods()
        assert np.array_equal(np.isneginf(fast), np.isneginf(ref))
        fin = np.isfinite(ref)
        assert 20 <= int(fin.sum()) < g.total
        err = np.abs(fast[fin] - ref[fin]) / np.abs(ref[fin])
        assert float(err.max()) <= 1e-11, (max_error, float(err.max()))
        pts = np.array([g.point(i) for i in range(g.total)])
        _check(m.loglikelihood_points(pts, kernel="recur"), ref, "packed strict kernel, list, max_error %d" % max_error, tol=1e-11)
        g.close()
        m.close()


def test_round5_tail_paths_on_awkward_shapes(hip_lib, oracle):
    """Round 5 re-routed how a histogram WITH A TAIL is evaluated -- the shape every real CovEst run hands the model
    (covest/histogram.py:105-134, covest/models.py:103-104).  K-factored sums sp_j per copy number in phase A and
    contracts it once at the end (ll_factored.hip SPO), walks count-less tiles without contracting them, takes the last
    COUNTED item first and skips dead units; K-basic takes the closed form for the logs and walks the tiles it covers for
    their sums alone (ll_basic.hip).  Histograms for what those routes could trip over: a single partial tile; keys with
    gaps that the recurrence bridges with filler keys (scale 0: they must add nothing to sp_j); long stretches of
    zero-count keys (sum items) before, between and BEHIND the counted ones; a far counted key that dooms most of the
    grid.  On every one: IEEE specials in the same places as K-direct over the whole grid, the same arg-min, and a
    seeded sample against the oracle at 1e-9 with the graded tail slack."""
    from covest_amd import BasicModel, DenseGrid, RepeatsModel

    def falling(keys, top):
        return {int(j): max(1, int(top * math.exp(-0.07 * i))) for i, j in enumerate(keys)}

    zeros = lambda keys: {int(j): 0 for j in keys}
    hists = {
        "one partial tile": falling(range(1, 21), 5000),
        "gaps bridged by filler keys": {j: c for j, c in falling(range(1, 90), 30000).items() if j % 7 != 3},
        "count-less stretches": {**falling(range(1, 41), 9000), **zeros(range(41, 150)), **falling(range(150, 171), 700),
                                 **zeros(range(171, 400))},
        "a far key behind zeros": {**falling(range(1, 70), 50000), **zeros(range(70, 300)), 300: 3},
    }
    rng = np.random.default_rng(55)
    for name, hist in hists.items():
        top = max(j for j, c in hist.items() if c)
        for tail in (777,):
            rm = RepeatsModel(21, 100, hist, tail, max_error=8)
            orm = oracle.OracleModel("repeats", 21, 100, hist, tail, max_error=8)
            axes = [np.exp(np.linspace(np.log(0.4), np.log(max(4.0, 0.9 * top)), 12)), [0.004, 0.03, 0.2],
                    np.linspace(0.35, 0.95, 4), [0.5], [0.05, 0.3, 0.6, 0.95]]
            grid = DenseGrid(rm, axes)
            grid.evaluate(kernel="factored")
            assert grid.work()[2] == "ll_factored"
            fast, best = grid.loglikelihoods(), grid.argmin()
            grid.evaluate(kernel="direct")
            ref = grid.loglikelihoods()
            assert np.array_equal(np.isneginf(fast), np.isneginf(ref)) and np.array_equal(np.isnan(fast), np.isnan(ref)), name
            assert np.isfinite(ref).any(), name
            some = rng.choice(grid.total, size=48, replace=False)
            pts = np.array([grid.point(int(i)) for i in some])
            want = orm.compute_loglikelihood_many(pts, n_threads=16)
            slack = _tail_noise(orm, pts, want, tail)
            _slack_budget("round-5 tail paths, repeats, " + name, slack)
            _check(fast[some], want, "round-5 tail paths, repeats, " + name, slack=slack)
            k_ref = int(np.argmin(np.where(np.isnan(ref), np.inf, -ref)))
            assert best[1] == k_ref or rel_err(float(fast[best[1]]), float(ref[k_ref])) <= TOL, name
            grid.close()
            rm.close()
            m = BasicModel(21, 100, hist, tail, max_error=8)
            om = oracle.OracleModel("basic", 21, 100, hist, tail, max_error=8)
            cs = np.exp(np.linspace(np.log(0.3), np.log(3.0 * top), 320))
            g = DenseGrid(m, [cs, [0.002, 0.05, 0.3, 0.5]])
            g.evaluate(kernel="recur")
            assert g.work()[2] == "ll_basic"
            fast = g.loglikelihoods()
            g.evaluate(kernel="direct")
            ref = g.loglikelihoods()
            assert np.array_equal(np.isneginf(fast), np.isneginf(ref)) and np.array_equal(np.isnan(fast), np.isnan(ref)), name
            some = rng.choice(g.total, size=64, replace=False)
            pts = np.array([g.point(int(i)) for i in some])
            want = om.compute_loglikelihood_many(pts, n_threads=16)
            slack = _tail_noise(om, pts, want, tail)
            _slack_budget("round-5 tail paths, basic, " + name, slack)
            _check(fast[some], want, "round-5 tail paths, basic, " + name, slack=slack)
            g.close()
            m.close()


def test_selection_scan_on_the_device(hip_lib):
    """covest_grid_eval_scan / covest_grid_scan (include/covest_amd.h): the records the device lists are exactly where
    the reference's selection loop (covest/grid.py:65-70) changes its state, for any starting minimum -- replayed, the
    loop's (min_val, arg, diff) come out bit for bit as first_wins_scan over the values read back.  Also: a NaN and
    +inf never pass, ties keep the first index, a list longer than the device keeps comes back as None."""
    from covest_amd import DenseGrid, RepeatsModel
    from covest_amd.grid import first_wins_scan, replay_records
    hist = {1: 1000, 2: 500, 150: 6000, 151: 40, 153: 7}  # (small coverages leave the far keys a p_j of 0: LL = -inf)
    m = RepeatsModel(21, 100, hist, 0, max_error=8)
    axes = [np.exp(np.linspace(np.log(0.05), np.log(60.0), 9)), np.linspace(0.001, 0.3, 6), np.linspace(0.3, 1.0, 6),
            [0.2, 0.5, float("nan")], np.linspace(0.05, 0.95, 8)]
    grid = DenseGrid(m, axes)
    grid.evaluate()
    vals = -grid.loglikelihoods()
    assert np.isnan(vals).any() and np.isposinf(vals).any() and np.isfinite(vals).any()
    finite = np.sort(vals[np.isfinite(vals)])
    starts = [math.inf, float(finite[-1]), float(np.median(finite)), float(finite[3]), float(finite[0]),
              float(finite[0]) - 1.0, float(finite[len(finite) // 3]) + 1e-9]
    for start in starts:
        grid.evaluate(scan_start=start)
        rec = grid.scan_records()
        assert rec is not None
        want = first_wins_scan(vals, start, 1)
        got = replay_records(rec[0], rec[1], start)
        assert got == want, (start, got, want)
        assert list(rec[0]) == sorted(rec[0]) and all(vals[i] == v for i, v in zip(rec[0], rec[1]))
        assert grid.argmin() == (float(np.nanmin(vals)), int(np.flatnonzero(vals == np.nanmin(vals))[0]))
    # a block of the grid: the records carry GLOBAL flat indices
    lo, hi = 1000, 5000
    part = DenseGrid(m, axes, (lo, hi))
    part.evaluate(scan_start=math.inf)
    rec = part.scan_records()
    want = first_wins_scan(vals[lo:hi], math.inf, 1)
    got = replay_records(rec[0], rec[1], math.inf)
    assert (got[0], got[1] - lo, got[2]) == want
    part.close()
    # more strict records than the device keeps (a steadily falling objective along the fastest axis): None
    bm_axes = [np.linspace(20.0, 120.0, 400), [0.05], [0.9], [0.5], [0.5]]  # (towards the optimum near c = 150)
    long_grid = DenseGrid(m, bm_axes)
    long_grid.evaluate(scan_start=math.inf)
    v2 = -long_grid.loglikelihoods()
    n_rec = len(np.flatnonzero(np.concatenate(([True], v2[1:] < np.minimum.accumulate(v2)[:-1]))))
    assert n_rec > 128, n_rec
    assert long_grid.scan_records() is None
    long_grid.evaluate()  # a plain evaluation leaves no list either
    assert long_grid.scan_records() is None
    long_grid.close()
    grid.close()
    m.close()


def test_estimator_fix_and_err_scale_on_gpu(hip_lib, oracle):
    """CoverageEstimator.likelihood_f (covest/covest.py:26-31) with `fix` and `err_scale != 1` -- the "4-D grid =
    q2 fixed" case of SURVEY discrepancy 1 -- on the GPU: the scalar objective, its batched form negll_grid, and
    optimize_grid over it, against the oracle evaluated at the model-space points the adapter must produce."""
    from covest_amd import CoverageEstimator, RepeatsModel, optimize_grid
    hist = load_hist("sim_c10_e0.05")
    m = RepeatsModel(21, 100, hist, 0, max_error=8)
    om = oracle.OracleModel("repeats", 21, 100, hist, 0, max_error=8)
    fix = [None, None, None, 0.5, None]
    est = CoverageEstimator(m, err_scale=10, fix=fix)
    assert est.bounds[1] == (0, 5.0)  # the error-rate bound lives in scaled space (covest/covest.py:22-24)
    x = [10.0, 0.5, 0.7, 0.123, 0.4]  # optimiser space: error rate x 10, q2 ignored (fixed)
    want = -om.compute_loglikelihood(10.0, 0.05, 0.7, 0.5, 0.4)
    assert rel_err(est.likelihood_f(x), want) <= TOL
    axes = [[9.0, 10.0, 11.0], [0.3, 0.5, 0.8], [0.6, 0.8], [0.123], [0.2, 0.5, 0.7]]
    got = est.negll_grid(axes)
    pts = np.array([(c, e / 10, q1, 0.5, q) for c in axes[0] for e in axes[1] for q1 in axes[2] for q in axes[4]])
    ref = -om.compute_loglikelihood_many(pts, n_threads=16)
    _check(got, ref, "negll_grid fix + err_scale")
    # the search itself: every iterate is a grid point, so the end point must reproduce under the oracle, q2 must
    # still be the fixed value, and no grid point the search saw may beat it
    guess = [10.0, 0.5, 0.65, 0.5, 0.5]
    trace = []
    res = optimize_grid(est.likelihood_f, guess, bounds=est.bounds, fix=fix, trace=trace)
    assert res[3] == 0.5
    final = trace[-1]
    model_pt = (res[0], res[1] / 10, res[2], 0.5, res[4])
    assert rel_err(final["value"], -om.compute_loglikelihood(*model_pt)) <= TOL
    assert all(t["grid_size"] <= 6 ** 4 for t in trace)  # a fixed dimension contributes one value
    assert final["value"] <= est.likelihood_f(guess)
    # the same search without err_scale ends at the same model-space point: the scaling is only a reparametrisation
    plain = CoverageEstimator(m, fix=fix)
    res1 = optimize_grid(plain.likelihood_f, [10.0, 0.05, 0.65, 0.5, 0.5], bounds=plain.bounds, fix=fix)
    assert rel_err(plain.likelihood_f(res1), final["value"]) <= 1e-6


def test_reference_overflow_is_flagged(hip_lib, oracle):
    """The documented divergence: beyond ln(l^i / i!) = 11356.5 the reference's long-double
    pmf product is inf before it is scaled (c_src/covest_poissonmodule.c:19-24) and its likelihood +inf or NaN; the
    kernels return the finite value.  covest_reference_overflow must mark exactly the points where the faithful
    oracle -- bit-equal to the reference's extension, tests/test_oracle_vs_ref.py -- stops being finite that way."""
    from covest_amd import BasicModel, RepeatsModel
    hist = {1: 5, 2: 3, 10000: 2}
    m = BasicModel(21, 100, hist, 0, max_error=8)
    om = oracle.OracleModel("basic", 21, 100, hist, 0, max_error=8)
    pts = np.array([(c, 0.001) for c in np.linspace(14300.0, 14900.0, 25)])
    ref = om.compute_loglikelihood_many(pts, n_threads=16)
    want = np.array([v == math.inf or v != v for v in ref])
    assert want.any() and not want.all()
    assert np.array_equal(m.reference_overflows(pts), want)
    got = m.loglikelihood_points(pts)
    assert np.all(np.isfinite(got[want]))  # the finite value the formula defines
    _check(got[~want], ref[~want], "below the overflow boundary")
    # repeats model: the copy number multiplies the rate, so the boundary moves with threshold_o
    rm = RepeatsModel(21, 100, hist, 0, max_error=8)
    orm = oracle.OracleModel("repeats", 21, 100, hist, 0, max_error=8)
    rp = np.array([(c, 0.001, 0.5, 0.5, q) for c in (300.0, 700.0, 1500.0) for q in (0.9, 0.5, 0.3, 0.2)])
    rref = orm.compute_loglikelihood_many(rp, n_threads=16)
    rwant = np.array([v == math.inf or v != v for v in rref])
    assert rwant.any() and not rwant.all()
    assert np.array_equal(rm.reference_overflows(rp), rwant)


_PAIR_SCRIPT = r'''
import os, sys
import torch                      # FIRST: one HIP runtime per process -- the one torch ships; the library binds to it
torch.cuda.init()
sys.path.insert(0, os.environ["COVEST_REPO"]); sys.path.insert(0, os.path.join(os.environ["COVEST_REPO"], "tests"))
import numpy as np
from conftest import load_hist
from covest_amd import DenseGrid, RepeatsModel
from covest_amd.grid import _scan_pairs_tensor, distributed_argmin
m = RepeatsModel(21, 100, load_hist("sim_c10_e0.05"), 0, max_error=8)
axes = [np.linspace(8, 12, 5), np.linspace(0.03, 0.07, 4), np.linspace(0.5, 1.0, 3), [0.5], np.linspace(0.05, 0.95, 6)]
whole = DenseGrid(m, axes)
pairs = []
for a, b in ((0, 100), (100, 360)):
    g = DenseGrid(m, axes, (a, b))
    g.evaluate(stream=torch.cuda.current_stream().cuda_stream)
    t = g.argmin_pair_tensor(torch.cuda.current_device())
    assert t.is_cuda and t.dtype == torch.float64 and tuple(t.shape) == (2,)
    v, i = g.argmin()
    assert t.cpu().tolist() == [v, float(i)] and a <= i < b, (t, v, i)
    assert distributed_argmin(None, None, pair=t) == (v, i)  # no process group: the identity
    pairs.append(t.clone())
    g.close()
whole.evaluate()
best = whole.argmin()
got = _scan_pairs_tensor(torch.stack(pairs)).cpu().tolist()
assert (got[0], int(got[1])) == best, (got, best)
# the same through a one-rank RCCL group: the all-gather consumes the HBM pair where the arg-min kernel left it
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
g = DenseGrid(m, axes)
g.evaluate(stream=torch.cuda.current_stream().cuda_stream)
t = g.argmin_pair_tensor(0)
out = torch.empty((1, 2), dtype=torch.float64, device="cuda")
dist.all_gather_into_tensor(out, t)
assert out.cpu().tolist()[0] == [best[0], float(best[1])]
dist.destroy_process_group()
print("pair ok", best)
'''


def test_argmin_pair_in_hbm_feeds_the_exchange(hip_lib):
    """Multi-GPU exchange without a host round trip: the arg-min kernel's (min, GLOBAL index) pair is
    adopted by torch where it lies in HBM (DenseGrid.argmin_pair_tensor), scanned with the tensor form of the
    selection rule, and fed to an RCCL all-gather (a one-rank group: the box has one GPU); it must be what
    covest_grid_argmin copies out, for a block that does not start at 0 too.  In a process of its own: torch has to
    initialise its HIP runtime before the library's first call (as bench.py does) -- a process has ONE runtime."""
    import subprocess
    import sys
    env = dict(os.environ, COVEST_REPO=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    proc = subprocess.run([sys.executable, "-c", _PAIR_SCRIPT], env=env, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0 and "pair ok" in proc.stdout, proc.stdout[-2000:] + proc.stderr[-4000:]


def test_block_partition_equals_whole_grid(hip_lib):
    """Multi-GPU block partition on one device: evaluating the
    blocks of a partition separately gives the same values and the same winner."""
    from covest_amd import DenseGrid, RepeatsModel
    from covest_amd.grid import partition_flat_range, repeats_cost_weights
    hist = load_hist("sim_c10_e0.05")
    m = RepeatsModel(21, 100, hist, 0, max_error=8)
    axes = [np.linspace(8, 12, 5), np.linspace(0.03, 0.07, 4), np.linspace(0.5, 1.0, 3),
            np.linspace(0.1, 0.9, 3), np.linspace(0.05, 0.95, 6)]
    whole = DenseGrid(m, axes)
    whole.evaluate()
    ll = whole.loglikelihoods()
    best = whole.argmin()
    bounds = partition_flat_range(whole.total, 3, repeats_cost_weights(m, axes))
    parts, winners = [], []
    for r in range(3):
        g = DenseGrid(m, axes, (bounds[r], bounds[r + 1]))
        g.evaluate()
        parts.append(g.loglikelihoods())
        winners.append(g.argmin())
    assert np.array_equal(np.concatenate(parts), ll, equal_nan=True)
    assert min(winners) == best


def test_lbfgsb_refinement_reaches_reference_optimum(hip_lib):
    """The default (non-grid) flow of covest.covest.main: scipy L-BFGS-B with finite-difference
    gradients over likelihood_f (covest/covest.py:33-39,55), every call a GPU evaluation.  The
    reference, on its own test histogram, ends at (10.019077633773197, 0.04999234428925103) with
    LL -3678682.5790824727 (measured by running the reference)."""
    from covest_amd import BasicModel, CoverageEstimator
    hist = load_hist("sim_c10_e0.05")
    m = BasicModel(21, 100, hist, 0, max_error=8)
    est = CoverageEstimator(m)
    res, ok = est.compute_coverage([10.0, 0.05], starting_points=1, use_grid_search=False)
    ll = m.compute_loglikelihood(*res)
    assert rel_err(ll, -3678682.5790824727) <= 1e-9
    # the optimum is flat to 1e-9 in LL over ~1e-3 in c: where L-BFGS-B stops depends on rounding noise
    assert abs(res[0] - 10.019077633773197) <= 2e-3 and abs(res[1] - 0.04999234428925103) <= 1e-5


def test_optimize_grid_trace(hip_lib):
    """covest.grid.optimize_grid on the reference's own test histogram: the batched
    GPU search must walk the same iterations to the same arguments."""
    from covest_amd import CoverageEstimator, optimize_grid
    g = load_golden("grid_trace.json")
    for tr in g["traces"]:
        m = _gpu_model(tr["model"], tr)
        est = CoverageEstimator(m)
        trace = []
        res = optimize_grid(est.likelihood_f, list(tr["initial_guess"]), bounds=est.bounds, trace=trace)
        sizes = [int(line.split("Grid size:")[1]) for line in tr["log"] if "Grid size" in line]
        assert [t["grid_size"] for t in trace] == sizes
        assert list(res) == tr["result"], (tr["model"], res, tr["result"])


def test_reference_overflow_mode(hip_lib):
    """The reference-faithful overflow mode (COVEST_KERNEL_DIRECT_REF, CoverageEstimator(reference_specials=True),
    optimize_grid(..., reference_specials=True)) against tests/golden/overflow.json: the reference's likelihoods where
    its long-double pmf product overflows (c_src/covest_poissonmodule.c:19-24) -- +inf, NaN (inf - inf), -inf and the
    finite values beside them, tail term dropped where sp_j is not < 1 -- and its own optimize_grid traces that walk
    into the band and stop at the first -(+inf) (covest/grid.py:65-70).  The default behaviour stays: finite values."""
    from covest_amd import BasicModel, CoverageEstimator, DenseGrid, RepeatsModel, optimize_grid
    g = load_golden("overflow.json")
    classes = {"nan": 0, "+inf": 0, "-inf": 0, "finite": 0}
    for c in g["cases"]:
        hist = {int(j): int(v) for j, v in c["hist_items"]}
        cls = BasicModel if c["model"] == "basic" else RepeatsModel
        m = cls(c["k"], c["r"], hist, c["tail"], max_error=c["max_error"])
        pts = np.array(c["points"])
        got = m.loglikelihood_points(pts, kernel="direct_ref")
        _check(got, c["ll"], "overflow %s %s tail=%s" % (c["model"], c["hist"], c["tail"]))
        for b in c["ll"]:
            classes["nan" if b != b else "+inf" if b == math.inf else "-inf" if b == -math.inf else "finite"] += 1
        # the estimator's substitution reaches the same values through the host's flags
        est = CoverageEstimator(m, reference_specials=True)
        _check(-est.negll_points([list(p) for p in c["points"]]), c["ll"], "overflow via the estimator")
        # every point the reference overflows at is flagged by the host test, and the DEFAULT kernels stay finite
        # (or -inf) there
        flags = m.reference_overflows(pts)
        want = np.array(c["ll"])
        assert flags[np.isposinf(want) | np.isnan(want)].all()
        plain = m.loglikelihood_points(pts)
        assert not np.isposinf(plain).any() and not np.isnan(plain).any()
        same = ~flags
        _check(plain[same], want[same], "outside the band the default kernels agree")
        m.close()
    assert min(classes.values()) > 0, classes
    for tr in g["traces"]:
        hist = {int(j): int(v) for j, v in tr["hist_items"]}
        m = BasicModel(tr["k"], tr["r"], hist, tr["tail"], max_error=tr["max_error"])
        est = CoverageEstimator(m)
        trace = []
        res = optimize_grid(est.likelihood_f, list(tr["initial_guess"]), bounds=est.bounds, trace=trace,
                            reference_specials=True)
        assert est.reference_specials is False  # (for the duration of the call only)
        sizes = [int(line.split("Grid size:")[1]) for line in tr["log"] if "Grid size" in line]
        assert [t["grid_size"] for t in trace] == sizes
        assert list(res) == tr["result"], (tr["hist"], res, tr["result"])
        assert trace[-1]["value"] == tr["result_negll"] == -math.inf
        # the same grid on the device: K-direct-ref + the arg-min kernel pick the first -(+inf) as the scan does
        plain = optimize_grid(est.likelihood_f, list(tr["initial_guess"]), bounds=est.bounds)
        assert math.isfinite(est.likelihood_f(plain)) and list(plain) != tr["result"]
        axes = [np.linspace(0.9 * tr["result"][0], 1.1 * tr["result"][0], 24), np.linspace(0.0, 0.004, 8)]
        grid = DenseGrid(m, axes)
        grid.evaluate(kernel="direct_ref")
        ll = grid.loglikelihoods()
        val, arg = grid.argmin()
        first = int(np.flatnonzero(np.isposinf(ll))[0])
        assert val == -math.inf and arg == first
        grid.close()
        m.close()
    print("overflow mode: classes checked", classes, "traces", len(g["traces"]))


@pytest.mark.parametrize("seed", list(range(1, 1 + int(os.environ.get("COVEST_FUZZ_SEEDS", "50")))))
def test_fuzz_random_histograms(hip_lib, oracle, seed):
    """Random histograms (gapped keys, zero counts, huge counts, tail or not) and random points incl.
    the bound corners: every kernel that accepts the request against the oracle."""
    from covest_amd import BasicModel, DenseGrid, RepeatsModel
    rng = np.random.default_rng(seed)
    n_keys = int(rng.integers(3, 120))
    keys = np.sort(rng.choice(np.arange(1, 400), size=n_keys, replace=False))
    counts = rng.integers(0, 10 ** int(rng.integers(1, 10)), size=n_keys)
    counts[rng.random(n_keys) < 0.2] = 0
    hist = {int(k): int(v) for k, v in zip(keys, counts)}
    tail = int(rng.choice([0, 0, 13, 100000]))
    k = int(rng.choice([15, 21, 31]))
    r = int(rng.choice([50, 100, 151]))
    # --- basic model: point list through direct and recur ---
    m = BasicModel(k, r, hist, tail, max_error=8)
    om = oracle.OracleModel("basic", k, r, hist, tail, max_error=8)
    pts = np.column_stack([np.exp(rng.uniform(np.log(0.005), np.log(500), 60)), rng.uniform(-0.05, 0.6, 60)])
    pts[:4] = [(0.01, 0.0), (0.01, 0.5), (400.0, 0.0), (400.0, 0.5)]
    ref = om.compute_loglikelihood_many(pts, n_threads=16)
    slack = _tail_noise(om, pts, ref, tail)
    _slack_budget("fuzz basic seed %d" % seed, slack)
    for kernel in ("direct", "recur"):
        _check(m.loglikelihood_points(pts, kernel=kernel), ref, "fuzz basic %s seed %d" % (kernel, seed), slack=slack)
    # --- repeats model: dense grid through direct and factored, point list through direct ---
    rm = RepeatsModel(k, r, hist, tail, max_error=8)
    orm = oracle.OracleModel("repeats", k, r, hist, tail, max_error=8)
    axes = [np.exp(rng.uniform(np.log(0.5), np.log(60), 3)), rng.uniform(0.0, 0.5, 3), rng.uniform(0.3, 1.0, 4),
            rng.uniform(0.0, 1.0, 3), np.concatenate([rng.uniform(0.02, 1.0, 4), [0.0, 1.0]])]
    grid = DenseGrid(rm, axes)
    gp = np.array([grid.point(i) for i in range(grid.total)])
    gref = orm.compute_loglikelihood_many(gp, n_threads=16)
    gslack = _tail_noise(orm, gp, gref, tail)
    _slack_budget("fuzz repeats seed %d" % seed, gslack)
    for kernel in ("direct", "factored"):
        grid.evaluate(kernel=kernel)
        ll = grid.loglikelihoods()
        _check(ll, gref, "fuzz repeats %s seed %d" % (kernel, seed), slack=gslack)
        k_ref, _ = oracle.first_min(-gref)
        val, arg = grid.argmin()
        assert arg == k_ref or rel_err(float(ll[arg]), float(gref[k_ref])) <= TOL
    sub = _Slack(gslack[::7])
    sub.classes, sub.unit = gslack.classes[::7], gslack.unit[::7]
    _check(rm.loglikelihood_points(gp[::7]), gref[::7], "fuzz repeats list seed %d" % seed, slack=sub)
