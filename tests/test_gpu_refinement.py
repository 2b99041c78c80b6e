"""Row F2 on the GPU: scalar refinement with the gradient's P + 1 evaluations in one launch, and the
multi-start in lock step -- identical, iterate for iterate, to the reference's call pattern driven by
the same kernels, and ending where the reference ends."""
import random

import numpy as np
import pytest

from conftest import load_hist, rel_err

pytestmark = pytest.mark.gpu


def _same(a, b):
    return np.array_equal(a.x, b.x) and a.fun == b.fun and a.nit == b.nit and a.success == b.success


def test_batched_refinement_is_the_same_optimisation(hip_lib):
    from covest_amd import BasicModel, CoverageEstimator, RepeatsModel
    hist = load_hist("sim_c10_e0.05")
    for model, start in ((BasicModel(21, 100, hist, 0, max_error=8), [10.0, 0.05]),
                         (BasicModel(21, 100, hist, 250, max_error=8), [7.0, 0.1]),
                         (RepeatsModel(21, 100, hist, 0, max_error=8), [10.0, 0.05, 0.8, 0.5, 0.3])):
        a = CoverageEstimator(model, batched=False)._optimize(start)
        b = CoverageEstimator(model, batched=True)._optimize(start)
        assert _same(a, b), (type(model).__name__, a.x, b.x, a.fun, b.fun)


def test_reference_optimum_through_the_batched_path(hip_lib):
    """covest.covest.main's default flow on the reference's own test histogram ends at
    (10.019077633773197, 0.04999234428925103), LL -3678682.5790824727."""
    from covest_amd import BasicModel, CoverageEstimator
    m = BasicModel(21, 100, load_hist("sim_c10_e0.05"), 0, max_error=8)
    res, ok = CoverageEstimator(m).compute_coverage([10.0, 0.05], starting_points=1, use_grid_search=False)
    assert rel_err(m.compute_loglikelihood(*res), -3678682.5790824727) <= 1e-9
    assert abs(res[0] - 10.019077633773197) <= 2e-3 and abs(res[1] - 0.04999234428925103) <= 1e-5


def test_lock_step_multi_start(hip_lib):
    from covest_amd import CoverageEstimator, RepeatsModel, initial_grid
    from covest_amd.estimator import _LockStep
    m = RepeatsModel(21, 100, load_hist("sim_c10_e0.05"), 0, max_error=8)
    est = CoverageEstimator(m)
    random.seed(3)
    starts = initial_grid([10.0, 0.05, 0.8, 0.5, 0.3], count=6, bounds=est.bounds)
    seq = [CoverageEstimator(m, batched=False)._optimize(s) for s in starts]
    lock = _LockStep(est.negll_points, len(starts))
    par = lock.map(est._optimize, starts)
    assert all(_same(a, b) for a, b in zip(seq, par))
    assert lock.rounds == max(r.nfev for r in par)
    # compute_coverage picks the first strictly smallest objective, as covest/covest.py:60-69
    want = min(seq, key=lambda r: r.fun)
    for flag in (False, True):
        random.seed(3)
        got, ok = CoverageEstimator(m, lock_step=flag).compute_coverage([10.0, 0.05, 0.8, 0.5, 0.3], starting_points=6)
        assert list(got) == list(want.x) and ok == want.success


def test_point_lists_of_any_threshold(hip_lib, oracle):
    """A repeats-model point list goes to K-factored in list mode: one workgroup per point, and a point whose
    threshold_o exceeds the workgroup's 512 lanes in chunks of 512 copy numbers.  Against K-direct (pinned to
    the reference elsewhere) on the 10 000-key histogram, against the oracle where it finishes in seconds,
    and independent of what else shares the call."""
    from covest_amd import RepeatsModel
    hist = load_hist("H10k_rep")
    m = RepeatsModel(21, 100, hist, 0, max_error=8)
    pts = np.array([
        [25.0, 0.02, 0.6, 0.5, 0.1],        # T ~ 140
        [25.0, 0.02, 0.6, 0.5, 0.03],       # T ~ 480: one workgroup, nearly full
        [25.0, 0.02, 0.6, 0.5, 0.0254],     # T = 514: two chunks, the second holds one copy number
        [25.0, 0.02, 0.6, 0.5, 0.02],       # two chunks
        [18.0, 0.05, 0.3, 0.2, 0.01],       # T ~ 1500
        [30.0, 0.01, 0.9, 0.9, 0.002],      # T ~ 6000
        [22.0, 0.03, 0.5, 0.5, 1e-6],       # b_o never reaches the threshold: T = max(hist) = 10000
        [25.0, 0.02, 1.0, 0.0, 0.5],        # q1 = 1: T = 2
        [25.0, 0.02, 0.6, 0.5, 1.0],        # q = 1
    ])
    T = m.get_hist_threshold_values(pts[:, 2:5])
    assert T.max() == 10000 and (T > 513).sum() >= 4 and (T <= 513).sum() >= 3 and ((T > 513) & (T < 540)).any(), T
    auto = m.loglikelihood_points(pts)
    direct = m.loglikelihood_points(pts, kernel="direct")
    for a, b, t in zip(auto, direct, T):
        assert rel_err(float(a), float(b)) <= 1e-11, (t, a, b)
    # batch composition does not matter: singly, reversed, duplicated
    assert np.array_equal(auto, np.array([m.loglikelihood_points(pts[i:i + 1])[0] for i in range(len(pts))]))
    assert np.array_equal(auto[::-1], m.loglikelihood_points(pts[::-1].copy()))
    # with a tail, on a small histogram the oracle can afford (T up to max(hist) = 700)
    rng = np.random.default_rng(5)
    small = {j: int(v) for j, v in zip(range(1, 120), rng.integers(1, 5000, size=119))}
    small[700] = 2
    for tail in (0, 77):
        ms = RepeatsModel(21, 100, small, tail, max_error=8)
        om = oracle.OracleModel("repeats", 21, 100, small, tail, max_error=8)
        ps = np.array([[8.0, 0.03, 0.5, 0.4, 0.02], [5.0, 0.01, 0.7, 0.3, 0.004], [9.0, 0.02, 0.4, 0.6, 1e-5]])
        assert ms.get_hist_threshold_values(ps[:, 2:5]).max() == 700
        got = ms.loglikelihood_points(ps)
        want = om.compute_loglikelihood_many(ps, n_threads=8)
        for a, b in zip(got, want):
            assert rel_err(float(a), float(b)) <= 1e-9, (tail, a, b)


def test_long_point_lists_take_the_throughput_kernel(hip_lib):
    """More than 4096 points in one call are throughput work: AUTO hands them to K-direct (same values to 1e-11),
    an explicit request for the factored kernel is refused."""
    from covest_amd import RepeatsModel
    from covest_amd._capi import CovestHipError
    m = RepeatsModel(21, 100, load_hist("sim_c10_e0.05"), 0, max_error=8)
    rng = np.random.default_rng(0)
    pts = np.column_stack([rng.uniform(5, 15, 5000), rng.uniform(0.01, 0.1, 5000), rng.uniform(0.3, 1, 5000),
                           rng.uniform(0, 1, 5000), rng.uniform(0.05, 1, 5000)])
    long_auto = m.loglikelihood_points(pts)
    short_auto = m.loglikelihood_points(pts[:4096])
    direct = m.loglikelihood_points(pts, kernel="direct")
    assert np.array_equal(long_auto, direct)
    assert all(rel_err(float(a), float(b)) <= 1e-11 for a, b in zip(short_auto, direct[:4096]))
    with pytest.raises(CovestHipError):
        m.loglikelihood_points(pts, kernel="factored")
