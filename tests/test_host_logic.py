"""Host-side logic of the grid search, on the CPU: partitioning, the first-wins
scan, threshold_o through the C ABI (pure host code), and optimize_grid driven by
the oracle as an opaque callable against the reference's own trace."""
import math

import numpy as np
import pytest

from conftest import load_golden, load_hist


def test_partition_uniform():
    from covest_amd.grid import partition_flat_range
    for total in (0, 1, 7, 100, 262144):
        for world in (1, 2, 3, 8):
            b = partition_flat_range(total, world)
            assert b[0] == 0 and b[-1] == total and len(b) == world + 1
            sizes = np.diff(b)
            assert (sizes >= 0).all() and sizes.max() - sizes.min() <= 1


def test_partition_weighted_balances_cost():
    from covest_amd.grid import partition_flat_range
    rng = np.random.default_rng(0)
    w = rng.integers(1, 300, size=256).astype(float)
    total = 256 * 1024
    for world in (2, 3, 8):
        b = partition_flat_range(total, world, w)
        assert b[0] == 0 and b[-1] == total and all(x <= y for x, y in zip(b, b[1:]))
        full = np.tile(w, 1024)
        costs = [full[b[r]:b[r + 1]].sum() for r in range(world)]
        assert max(costs) - min(costs) <= 2 * w.max()
    assert partition_flat_range(10, 4, np.zeros(5)) == [0, 2, 5, 7, 10]


def test_first_wins_scan_matches_oracle(oracle):
    from covest_amd.grid import first_wins_scan
    rng = np.random.default_rng(1)
    inf, nan = math.inf, math.nan
    cases = [[3.0, 1.0, 1.0, 2.0], [nan, 5.0, nan, 4.0], [inf, inf], [1.0, -inf, -inf], [],
             [5.0, 6.0]]
    for _ in range(200):
        v = rng.normal(size=rng.integers(1, 60))
        v[rng.random(len(v)) < 0.1] = nan
        v[rng.random(len(v)) < 0.05] = inf
        v = np.round(v, 1)  # many ties
        cases.append(v.tolist())
    for v in cases:
        for start in (inf, 0.3):
            mv, arg, diff = first_wins_scan(v, start)
            k, best = oracle.first_min(v, start)
            assert (arg, mv) == (k, best)
            # diff = the sequential accumulation of covest/grid.py:68
            d, cur = 0.0, start
            for x in v:
                if x < cur:
                    d += cur - x
                    cur = x
            assert diff == d or (math.isnan(diff) and math.isnan(d))


def _device_scan_model(vals, start, threads=1024, cap=120):
    """The algorithm of argmin.hip argmin_scan_small in numpy: a contiguous run of the values per thread, an exclusive
    prefix minimum across the threads seeded with `start`, then every thread lists the strict running-minimum records
    of its run -- (indices, values, truncated)."""
    v = np.asarray(vals, dtype=np.float64)
    n = len(v)
    chunk = (n + threads - 1) // threads if n else 0
    idx, out = [], []
    run0 = start
    for t in range(threads):
        lo, hi = min(n, t * chunk), min(n, t * chunk + chunk)
        run = run0
        for i in range(lo, hi):
            if v[i] < run:
                run = v[i]
                idx.append(i)
                out.append(float(v[i]))
        if hi > lo:
            with np.errstate(invalid="ignore"):
                m = np.fmin.reduce(np.where(np.isnan(v[lo:hi]), np.inf, v[lo:hi]))
            run0 = min(run0, float(m))
    return idx[:cap], out[:cap], len(idx) > cap


def test_replaying_the_device_records_is_the_selection_loop():
    """optimize_grid's selection loop (covest/grid.py:65-70) over the records the device lists
    (covest_grid_eval_scan: the strict running-minimum records below the starting minimum, in index order) ends in the
    same (min_val, arg, diff), bit for bit, as the loop over every value -- NaN, +inf, ties and all."""
    from covest_amd.grid import first_wins_scan, replay_records
    rng = np.random.default_rng(7)
    inf, nan = math.inf, math.nan
    cases = [[3.0, 1.0, 1.0, 2.0], [nan, 5.0, nan, 4.0], [inf, inf], [1.0, -inf, -inf], [], [5.0, 6.0]]
    for _ in range(120):
        v = rng.normal(size=rng.integers(1, 3000)) * 10.0 ** rng.integers(0, 9)
        v[rng.random(len(v)) < 0.1] = nan
        v[rng.random(len(v)) < 0.05] = inf
        if rng.random() < 0.5:
            v = np.round(v, 1)  # many ties
        cases.append(v.tolist())
    cases.append(np.linspace(1e6, 1.0, 200).tolist())  # steadily falling: more records than the device keeps
    for v in cases:
        finite = [x for x in v if x == x and abs(x) != inf]
        starts = [inf, 0.3] + ([float(np.median(finite)), min(finite), min(finite) - 1.0] if finite else [])
        for start in starts:
            idx, vals, truncated = _device_scan_model(v, start)
            if truncated:
                assert len(v) == 200
                continue
            assert idx == sorted(idx)
            got = replay_records(idx, vals, start)
            want = first_wins_scan(v, start)
            assert got[:2] == want[:2] and (got[2] == want[2] or (math.isnan(got[2]) and math.isnan(want[2]))), (start, got, want)


def test_threshold_through_capi_on_host(hip_lib):
    from covest_amd import RepeatsModel
    g = load_golden("threshold_o.json")
    rows = np.array(g["rows"])
    for hist_max in (15, 256, 10000):
        sel = rows[rows[:, 0] == hist_max]
        m = RepeatsModel(21, 100, {hist_max: 1, 1: 1}, 0, max_error=8)
        assert np.array_equal(m.get_hist_threshold_values(sel[:, 1:4]), sel[:, 4].astype(np.int32))
    m = RepeatsModel(21, 100, {50: 1}, 0, max_error=8, threshold=None)
    assert m.get_hist_threshold_values([[0.5, 0.5, 0.5]])[0] == 50


class _OracleNegLL:
    def __init__(self, om):
        self.om = om

    def __call__(self, x):
        return -self.om.compute_loglikelihood(*list(x))


@pytest.mark.parametrize("which", [0, 1])
def test_optimize_grid_reproduces_reference_trace(oracle, which):
    """covest_amd.grid.optimize_grid with an opaque callable (the oracle) must walk
    exactly the iterations the reference's optimize_grid logged."""
    from covest_amd.grid import optimize_grid
    tr = load_golden("grid_trace.json")["traces"][which]
    hist = load_hist(tr["hist"])
    om = oracle.OracleModel(tr["model"], tr["k"], tr["r"], hist, tr["tail"], max_error=tr["max_error"])
    trace = []
    res = optimize_grid(_OracleNegLL(om), list(tr["initial_guess"]),
                        bounds=[tuple(b) for b in tr["bounds"]], trace=trace)
    sizes = [int(line.split("Grid size:")[1]) for line in tr["log"] if "Grid size" in line]
    news = [line for line in tr["log"] if line.startswith("New args")]
    assert [t["grid_size"] for t in trace] == sizes
    assert list(res) == tr["result"]
    last = trace[-1]
    assert "ll: %r" % last["value"] in news[-1] or "ll: %s" % last["value"] in news[-1]


def test_initial_grid_shape():
    from covest_amd.grid import initial_grid
    pts = initial_grid([10.0, 0.05], count=5, bounds=[(0.01, None), (0, 0.5)])
    assert len(pts) == 5 and pts[0] == [10.0, 0.05]
    for c, e in pts[1:]:
        assert 10.0 / 3 <= c <= 30.0 and 0.05 / 3 <= e <= 0.15
    assert initial_grid([1.0, 0.1], count=0) == []
    pts = initial_grid([10.0, 0.05], count=3, fix=[None, 0.07])
    assert all(p[1] == 0.07 for p in pts[1:])


def test_comb_table_is_scipys():
    """covest_amd.models builds comb(k, s) * 3 ** s (covest/models.py:25) without importing scipy.special (its import
    is most of a first search's time); the numbers must be scipy's, bit for bit."""
    from scipy.special import comb
    from covest_amd.models import _comb_float, _comb_table
    for n in list(range(0, 70)) + [100, 150, 199]:
        for k in range(0, n + 1):
            assert _comb_float(n, k) == float(comb(n, k)), (n, k)
    assert _comb_table(21) == [comb(21, s) * (3 ** s) for s in range(22)]
