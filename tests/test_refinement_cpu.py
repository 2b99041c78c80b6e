"""Row F2 host logic without a GPU: the batched finite-difference gradient and the lock-step multi-start of
covest_amd.estimator against the reference's call pattern (scipy differentiating a scalar objective itself),
on a stand-in model whose likelihood is an ordinary Python function."""
import math
import random

import numpy as np

from covest_amd.estimator import CoverageEstimator, _LockStep
from covest_amd.grid import initial_grid


class _StubModel:
    """Smooth, bounded, with the API the estimator touches."""
    bounds = ((0.01, None), (0, 0.5), (0.3, 1), (0, 1), (0, 1))
    params = ('coverage', 'error_rate', 'q1', 'q2', 'q')

    def __init__(self):
        self.calls = 0
        self.points = 0

    def compute_loglikelihood(self, c, e, q1, q2, q):
        self.calls += 1
        self.points += 1
        return -(1e6 * ((math.log(c) - math.log(12.0)) ** 2 + 40 * (e - 0.03) ** 2 + (q1 - 0.9) ** 2
                        + 0.5 * (q2 - 0.4) ** 2 * (1 + c / 50) + (q - 0.2) ** 4) + 3.25e7)

    def loglikelihood_points(self, pts, kernel="auto"):
        self.calls += 1
        self.points += len(pts)
        calls, points = self.calls, self.points
        out = np.array([self.compute_loglikelihood(*p) for p in pts])
        self.calls, self.points = calls, points
        return out


def _same(a, b):
    return (np.array_equal(a.x, b.x) and a.fun == b.fun and a.nit == b.nit and a.nfev * 1 >= 1
            and a.success == b.success)


def test_batched_gradient_reproduces_scipys_own_differencing():
    for start in ([10.0, 0.05, 0.8, 0.5, 0.3], [30.0, 0.5, 1.0, 0.0, 1.0], [0.01, 0.0, 0.3, 1.0, 0.0]):
        plain, fast = _StubModel(), _StubModel()
        a = CoverageEstimator(plain, batched=False)._optimize(start)
        b = CoverageEstimator(fast, batched=True)._optimize(start)
        assert _same(a, b), (start, a.x, b.x)
        # one launch per gradient instead of P + 1 evaluations
        assert fast.points == plain.points and fast.calls * 6 == plain.calls


def test_fixed_parameters_and_error_scale_ride_along():
    fix = [None, None, 0.7, None, 0.25]
    plain, fast = _StubModel(), _StubModel()
    a = CoverageEstimator(plain, err_scale=10, fix=fix, batched=False).compute_coverage([9.0, 0.04, 0.5, 0.5, 0.5])
    b = CoverageEstimator(fast, err_scale=10, fix=fix, batched=True).compute_coverage([9.0, 0.04, 0.5, 0.5, 0.5])
    assert list(a[0]) == list(b[0]) and a[1] == b[1]


def test_lock_step_multi_start_equals_sequential():
    random.seed(11)
    est_seq = CoverageEstimator(_StubModel(), batched=False)
    starts = initial_grid([11.0, 0.04, 0.8, 0.5, 0.3], count=7, bounds=est_seq.bounds)
    seq = [est_seq._optimize(s) for s in starts]
    model = _StubModel()
    est = CoverageEstimator(model, batched=True)
    lock = _LockStep(est.negll_points, len(starts))
    par = lock.map(est._optimize, starts)
    assert all(_same(a, b) for a, b in zip(seq, par))
    # rounds: as many launches as the LONGEST refinement needs, not the sum
    longest = max(r.nfev for r in par)
    assert lock.rounds == longest == model.calls and lock.points == sum(6 * r.nfev for r in par)
    want = min(seq, key=lambda r: r.fun)
    for flag in (False, True):  # the schedule does not change the answer
        best = CoverageEstimator(_StubModel(), lock_step=flag)._best_of(starts)
        assert np.array_equal(best.x, want.x)


def test_a_failing_evaluation_reaches_the_caller():
    class Boom(RuntimeError):
        pass

    def bad(points):
        raise Boom("device lost")

    lock = _LockStep(bad, 3)
    est = CoverageEstimator(_StubModel())
    try:
        lock.map(est._optimize, [[10.0, 0.05, 0.8, 0.5, 0.3]] * 3)
    except Boom:
        return
    raise AssertionError("the failure was swallowed")


def test_restated_difference_scheme_is_scipys():
    """The finite-difference steps and quotient of estimator._fd_steps against the installed scipy's approx_derivative:
    the probe the estimator runs before it trusts them, then random points inside, on and within a step of the bounds,
    and a whole refinement through either path."""
    from scipy.optimize._numdiff import approx_derivative
    from covest_amd import estimator as est_mod
    lo = np.array([0.01, 0.0, 0.3, 0.0, 0.0])
    hi = np.array([np.inf, 0.5, 1.0, 1.0, 1.0])
    assert est_mod._fd_fast_matches_scipy(lo, hi, 1e-8)
    rng = np.random.default_rng(7)
    for trial in range(200):
        x = np.where(np.isfinite(hi), lo + (hi - lo) * rng.random(5), lo + 100 * rng.random(5))
        snap = rng.integers(0, 5, size=5)  # 1: on the lower bound, 2: on the upper, 3 / 4: half a step inside it
        x = np.where(snap == 1, lo, x)
        x = np.where((snap == 2) & np.isfinite(hi), hi, x)
        x = np.where(snap == 3, lo + 0.5e-8, x)
        x = np.where((snap == 4) & np.isfinite(hi), hi - 0.5e-8, x)
        seen = []
        approx_derivative(lambda z: seen.append(np.array(z)) or 0.0, x, method='2-point', abs_step=1e-8, f0=0.0, bounds=(lo, hi))
        steps = est_mod._fd_steps(x, lo, hi, 1e-8)
        for i, z in enumerate(seen):
            want = x.copy()
            want[i] = x[i] + steps[i]
            assert np.array_equal(z, want), (trial, i, x, z, want)
    for start in ([10.0, 0.05, 0.8, 0.5, 0.3], [0.01, 0.0, 0.3, 1.0, 0.0]):
        fast, slow = CoverageEstimator(_StubModel()), CoverageEstimator(_StubModel())
        slow._fd_fast = False  # scipy's approx_derivative twice per gradient, as before round 3
        a, b = fast._optimize(start), slow._optimize(start)
        assert fast._fd_fast is True and _same(a, b), (start, a.x, b.x)
