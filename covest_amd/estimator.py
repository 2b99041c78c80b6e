"""CoverageEstimator over a GPU-backed model.

The reference's estimator (covest/covest.py:18-96) is an adapter between the optimiser's
parameter space and the model's: optimiser vectors carry the error rate multiplied by
`err_scale`, and parameters named in `fix` are pinned.  Its contract, kept here:

  likelihood_f(x)      -> -LL(model-space(x))                       covest/covest.py:26-31
  bounds               -> model bounds with the error-rate upper bound scaled      :22-24
  compute_coverage(guess, starting_points, use_grid_search, n_threads)
                       -> (estimate in model space, success flag)   :41-96

What differs is how the likelihood gets evaluated: every call lands on the HIP kernels, and the
grid search asks for a whole grid at once (`negll_grid`) instead of mapping pickled calls over a
process pool.

Refinement.  The reference hands scipy's L-BFGS-B a scalar objective, so
scipy differentiates it numerically: P + 1 separate evaluations per gradient, and `-sp N` starts run
in N worker processes.  Here one gradient is ONE launch -- the point and its P finite-difference
neighbours go to the GPU together (`negll_points`) -- and the N starts can advance in lock step, their
requests merged into one launch per round (`_LockStep`, opt-in: see `_best_of`).  Which neighbours scipy would visit (step
1e-8, flipped at an upper bound, ...) is not re-derived: scipy's own `approx_derivative` is run twice,
first against a recorder to learn the points, then against the batch's values, so the gradient -- and
with it every iterate -- is bit-identical to what `minimize(..., jac=None)` computes from the same
likelihood values.  The kernels are deterministic per point, whatever else shares the launch.
"""
import threading
import itertools
import time

import numpy as np

from . import constants
from .grid import DenseGrid, initial_grid, optimize_grid


def _fd_steps(x, lo, hi, h0):
    """The steps scipy's 2-point scheme takes from x (scipy/optimize/_numdiff.py: approx_derivative with abs_step,
    then _adjust_scheme_to_bounds(..., 1, '1-sided', lb, ub)): forward by h0 where that stays inside the bounds,
    backward where only that does, else the whole distance to the farther bound."""
    h = np.full_like(x, h0)
    stuck = ((x + h) - x) == 0  # (a step lost in x's last bit: scipy falls back to a relative one)
    if stuck.any():
        sign = (x >= 0).astype(np.float64) * 2 - 1
        h = np.where(stuck, np.finfo(np.float64).eps ** 0.5 * sign * np.maximum(1.0, np.abs(x)), h)
    if np.all((lo == -np.inf) & (hi == np.inf)):
        return h
    lower, upper = x - lo, hi - x
    moved = x + h
    violated = (moved < lo) | (moved > hi)
    fitting = np.abs(h) <= np.maximum(lower, upper)
    h = h.copy()
    h[violated & fitting] *= -1
    forward = (upper >= lower) & ~fitting
    h[forward] = upper[forward]
    backward = (upper < lower) & ~fitting
    h[backward] = -lower[backward]
    return h


def _fd_fast_matches_scipy(lo, hi, h0):
    """True iff _fd_steps and the quotient built on it give, bit for bit, the points and the gradient the installed
    scipy's approx_derivative gives -- probed inside the bounds, on them and within a step of them."""
    try:
        from scipy.optimize._numdiff import approx_derivative
    except ImportError:
        return False
    rng = np.random.default_rng(12345)
    finite_lo = np.where(np.isfinite(lo), lo, -50.0)
    finite_hi = np.where(np.isfinite(hi), hi, 50.0)
    probes = [finite_lo + (finite_hi - finite_lo) * rng.random(len(lo)) for _ in range(4)]
    probes += [finite_lo.copy(), finite_hi.copy(), finite_lo + 0.5 * h0, finite_hi - 0.5 * h0,
               np.where(np.arange(len(lo)) % 2 == 0, finite_lo, finite_hi)]
    fun = lambda z: float(np.sum(np.sin(z) * np.arange(1, len(z) + 1)))
    for x in probes:
        x = np.clip(np.asarray(x, dtype=np.float64), lo, hi)
        seen = []

        def record(z):
            seen.append(np.array(z, dtype=np.float64))
            return fun(z)

        want = approx_derivative(record, x, method='2-point', abs_step=h0, f0=fun(x), bounds=(lo, hi))
        steps = _fd_steps(x, lo, hi, h0)
        pts = np.repeat(x[None, :], len(x), axis=0)
        idx = np.arange(len(x))
        pts[idx, idx] = x + steps
        if len(seen) != len(x) or any(not np.array_equal(a, b) for a, b in zip(seen, pts)):
            return False
        got = (np.array([fun(p) for p in pts]) - fun(x)) / (pts[idx, idx] - x)
        if not np.array_equal(np.asarray(want, dtype=np.float64), got):
            return False
    return True


class CoverageEstimator:
    ERROR_RATE = 1  # index of the parameter that err_scale applies to

    def __init__(self, model, err_scale=1, fix=None, batched=True, lock_step=False, reference_specials=False):
        self.model = model
        # True: where the reference's long-double pmf product overflows (c_src/covest_poissonmodule.c:19-24) the
        # objective is what the REFERENCE returns there -- -(+inf) or NaN -- instead of the finite value the formula
        # defines, so that a search ends where the reference's ends (covest/grid.py:65-70 selects -inf).  Off by
        # default: the overflow is a defect of the reference, not a feature of the model.
        self.reference_specials = reference_specials
        self.fix = fix
        self.err_scale = err_scale
        self.batched = batched      # value and gradient from one launch (else scipy differences a scalar objective)
        self.lock_step = lock_step  # multi-start: all starts advance together, one launch per round (see _best_of)
        self._grid = None           # the grid handle negll_grid keeps (see there)
        self._fd_bounds = None      # (lo, hi) arrays of the finite-difference scheme
        self._fd_fast = None        # the restated scheme reproduces the installed scipy's (checked on first use)
        self.timings = None         # a list: negll_grid appends {points, create_s, eval_s, readback_s, kernel} per call
        bounds = [tuple(b) for b in model.bounds]
        lo, hi = bounds[self.ERROR_RATE]
        bounds[self.ERROR_RATE] = (lo, hi * err_scale)
        self.bounds = bounds

    def __getstate__(self):
        # (an estimator's bound methods get pickled into Pool workers by the reference's callers: plain data only)
        state = dict(self.__dict__)
        state['_grid'] = None
        return state

    # ------------------------------------------------------------------ space mapping
    def _pinned(self, values):
        """Optimiser-space values with the fixed parameters substituted."""
        if self.fix is None:
            return list(values)
        return [v if f is None else f for v, f in zip(values, self.fix)]

    def _model_args(self, x):
        args = self._pinned(x)
        args[self.ERROR_RATE] = args[self.ERROR_RATE] / self.err_scale
        return args

    # ------------------------------------------------------------------ objective
    def likelihood_f(self, x):
        """The scalar objective handed to scipy and to optimize_grid: -LL."""
        if self.reference_specials:
            return float(self.negll_points([x])[0])
        return -self.model.compute_loglikelihood(*self._model_args(x))

    def _model_axes(self, axes):
        axes = [list(a) for a in axes]
        if self.fix is not None:
            axes = [a if f is None else [f] * len(a) for a, f in zip(axes, self.fix)]
        axes[self.ERROR_RATE] = [v / self.err_scale for v in axes[self.ERROR_RATE]]
        return axes

    def _grid_for(self, axes):
        """One grid handle per estimator, re-configured for every grid (covest_grid_reset): its device memory stays."""
        if self._grid is not None and self._grid._handle is not None and self._grid.model is self.model:
            return self._grid.reset(axes)
        self._grid = DenseGrid(self.model, axes)
        return self._grid

    def negll_grid(self, axes, kernel="auto", reference_specials=None):
        """likelihood_f over itertools.product(*axes) -- the map of covest/grid.py:59-64 -- as ONE
        dense-grid evaluation.  Returns an ndarray in product order.  reference_specials: None = the estimator's
        setting; a search that wants the other one passes it here (optimize_grid) instead of changing the estimator
        under the feet of whoever else is using it."""
        specials = self.reference_specials if reference_specials is None else bool(reference_specials)
        axes = self._model_axes(axes)
        t = self.timings
        t0 = time.perf_counter() if t is not None else 0.0
        grid = self._grid_for(axes)
        t1 = time.perf_counter() if t is not None else 0.0
        grid.evaluate(kernel=kernel)
        if t is not None:
            grid.argmin()  # (wait for the kernels: the split below is only meaningful with a sync here)
        t2 = time.perf_counter() if t is not None else 0.0
        out = -grid.loglikelihoods()
        if specials:
            mesh = np.meshgrid(*[np.asarray(a, dtype=np.float64) for a in axes], indexing="ij")  # (product order)
            pts = np.stack([m.reshape(-1) for m in mesh], axis=1)
            out = self._with_reference_specials(pts, out)
        if t is not None:
            t3 = time.perf_counter()
            t.append({"points": len(grid), "create_s": t1 - t0, "eval_s": t2 - t1, "readback_s": t3 - t2,
                      "kernel": grid.work()[2]})
        return out

    def negll_grid_scan(self, axes, min_val, kernel="auto", reference_specials=None):
        """negll_grid for optimize_grid's selection loop (covest/grid.py:65-70), with the loop's scan done where the
        values are: returns (records, values) -- records = (flat indices, -LL) of the strict running-minimum records
        below `min_val` in product order (DenseGrid.scan_records) and values = None, or records = None and the values
        as negll_grid returns them where the device's list does not apply (reference_specials, a list cut short)."""
        if self.reference_specials if reference_specials is None else reference_specials:
            return None, self.negll_grid(axes, kernel=kernel, reference_specials=True)
        axes = self._model_axes(axes)
        t = self.timings
        t0 = time.perf_counter() if t is not None else 0.0
        grid = self._grid_for(axes)
        t1 = time.perf_counter() if t is not None else 0.0
        grid.evaluate(kernel=kernel, scan_start=min_val)
        if t is not None:
            grid.argmin()
        t2 = time.perf_counter() if t is not None else 0.0
        records = grid.scan_records()
        values = None if records is not None else -grid.loglikelihoods()
        if t is not None:
            t3 = time.perf_counter()
            t.append({"points": len(grid), "create_s": t1 - t0, "eval_s": t2 - t1, "readback_s": t3 - t2,
                      "kernel": grid.work()[2]})
        return records, values

    def negll_points(self, xs, reference_specials=None):
        """likelihood_f of several optimiser-space vectors in one launch: ndarray."""
        pts = np.array([self._model_args(x) for x in xs], dtype=np.float64)
        out = -self.model.loglikelihood_points(pts)
        specials = self.reference_specials if reference_specials is None else bool(reference_specials)
        return self._with_reference_specials(pts, out) if specials else out

    def _with_reference_specials(self, pts, negll):
        """-LL with the reference's own result substituted where its pmf product overflows: the points the host
        test flags (model.reference_overflows: the largest rate against the largest key) go through
        COVEST_KERNEL_DIRECT_REF, which returns +inf / NaN / the value without the tail term exactly where the
        reference does (direct_point.h REF_OVF)."""
        flagged = np.flatnonzero(self.model.reference_overflows(pts))
        if len(flagged):
            negll = np.array(negll, dtype=np.float64)
            negll[flagged] = -self.model.loglikelihood_points(pts[flagged], kernel="direct_ref")
        return negll

    # ------------------------------------------------------------------ refinement
    FD_STEP = 1e-8  # scipy's default `eps` of L-BFGS-B, what the reference runs with

    def _value_and_gradient(self, x, evaluate):
        """(f, grad f) as scipy's 2-point scheme defines them, all P + 1 evaluations through one call of
        `evaluate(list of points)`."""
        x = np.asarray(x, dtype=np.float64)
        if self._fd_bounds is None:
            self._fd_bounds = (np.array([-np.inf if b[0] is None else b[0] for b in self.bounds]),
                               np.array([np.inf if b[1] is None else b[1] for b in self.bounds]))
        lo, hi = self._fd_bounds
        if self._fd_fast is None:
            self._fd_fast = _fd_fast_matches_scipy(lo, hi, self.FD_STEP)
        if self._fd_fast:
            # scipy's own arithmetic, restated (and checked against the installed scipy when this estimator first
            # needed it): two passes through approx_derivative cost 130 us of Python per gradient, the launch 85
            steps = _fd_steps(x, lo, hi, self.FD_STEP)
            pts = np.repeat(x[None, :], len(x), axis=0)
            idx = np.arange(len(x))
            pts[idx, idx] = x + steps
            values = np.asarray(evaluate([x] + list(pts)), dtype=np.float64)
            return float(values[0]), (values[1:] - values[0]) / (pts[idx, idx] - x)
        from scipy.optimize._numdiff import approx_derivative
        visited = []

        def record(z):
            visited.append(np.array(z, dtype=np.float64))
            return 0.0

        approx_derivative(record, x, method='2-point', abs_step=self.FD_STEP, f0=0.0, bounds=(lo, hi))
        values = evaluate([x] + visited)
        replay = iter(values[1:])
        grad = approx_derivative(lambda z: next(replay), x, method='2-point', abs_step=self.FD_STEP,
                                 f0=values[0], bounds=(lo, hi))
        return float(values[0]), np.asarray(grad, dtype=np.float64)

    def _optimize(self, start, evaluate=None):
        """One bounded quasi-Newton refinement, the reference's choice (method and options of
        covest/covest.py:33-39).  batched=False is the reference's call pattern (scipy differentiates
        a scalar objective itself); the default feeds scipy value and gradient from one launch."""
        from scipy.optimize import minimize
        if not self.batched:
            return minimize(self.likelihood_f, start, method=constants.OPTIMIZATION_METHOD,
                            bounds=self.bounds, options={'disp': False})
        evaluate = evaluate or self.negll_points
        return minimize(lambda x: self._value_and_gradient(x, evaluate), start, jac=True,
                        method=constants.OPTIMIZATION_METHOD, bounds=self.bounds, options={'disp': False})

    def _best_of(self, starts):
        """Refine every start; keep the first result with the strictly smallest objective
        (covest/covest.py:60-69).  The results do not depend on how the starts are scheduled.  Default: one
        after the other, each gradient one launch.  lock_step=True runs them as threads whose requests are
        merged into one launch per round -- fewer, fuller launches, but measured SLOWER on one GPU once a
        single evaluation costs < 0.1 ms (bench.py --workload f2: the threads' hand-offs through the GIL cost
        more than the launches they save); it pays when an evaluation is expensive (huge threshold_o)."""
        starts = list(starts)
        if self.batched and self.lock_step and len(starts) > 1:
            results = _LockStep(self.negll_points, len(starts)).map(self._optimize, starts)
        else:
            results = [self._optimize(s) for s in starts]
        best = None
        for res in results:
            if best is None or best.fun > res.fun:
                best = res
        return best

    def compute_coverage(self, guess, starting_points=1, use_grid_search=False,
                         n_threads=constants.DEFAULT_THREAD_COUNT):
        x = list(guess)
        x[self.ERROR_RATE] *= self.err_scale
        success = True
        try:
            if starting_points >= 1:
                starts = [x] if starting_points == 1 else initial_grid(
                    x, count=starting_points, bounds=self.bounds, fix=self.fix)
                best = self._best_of(starts)
                x, success = best.x, best.success
            if use_grid_search or (use_grid_search is None and not success):
                x = optimize_grid(self.likelihood_f, x, bounds=self.bounds, fix=self.fix, n_threads=n_threads)
        except KeyboardInterrupt:
            pass  # return the best so far, as the reference does
        x = list(x)
        x[self.ERROR_RATE] /= self.err_scale
        return x, success


class _LockStep:
    """Merge the evaluation requests of N concurrently running refinements into one launch per round.

    Every client thread calls `evaluate(points)` and blocks; when all clients still running have a
    request pending, the last one to arrive evaluates the concatenation and hands each its slice.  A
    client that finishes leaves the round (and, if everybody else is already waiting, fires it)."""

    def __init__(self, evaluate_batch, n_clients):
        self._evaluate_batch = evaluate_batch
        self._active = n_clients
        self._cv = threading.Condition()
        self._pending = {}   # client id -> list of points
        self._answers = {}   # client id -> ndarray
        self._failure = None
        self.rounds = 0
        self.points = 0

    def _fire(self):
        """Called with the lock held and every active client pending."""
        order = sorted(self._pending)
        merged = [p for cid in order for p in self._pending[cid]]
        try:
            values = np.asarray(self._evaluate_batch(merged))
        except BaseException as exc:  # hand the failure to every waiter
            self._failure = exc
            values = np.full(len(merged), np.nan)
        self.rounds += 1
        self.points += len(merged)
        at = 0
        for cid in order:
            n = len(self._pending[cid])
            self._answers[cid] = values[at:at + n]
            at += n
        self._pending.clear()
        self._cv.notify_all()

    def _evaluate(self, cid, points):
        with self._cv:
            self._pending[cid] = list(points)
            if len(self._pending) == self._active:
                self._fire()
            while cid not in self._answers:
                self._cv.wait()
            if self._failure is not None:
                raise self._failure
            return self._answers.pop(cid)

    def _leave(self):
        with self._cv:
            self._active -= 1
            if self._active > 0 and len(self._pending) == self._active:
                self._fire()

    def map(self, refine, starts):
        results = [None] * len(starts)
        errors = []

        def client(cid):
            try:
                results[cid] = refine(starts[cid], lambda pts: self._evaluate(cid, pts))
            except BaseException as exc:
                errors.append(exc)
            finally:
                self._leave()

        threads = [threading.Thread(target=client, args=(cid,), daemon=True) for cid in range(len(starts))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise errors[0]
        return results
