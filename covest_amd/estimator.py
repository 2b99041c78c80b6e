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
process pool.  Multi-start refinements run one after the other in-process (each is a stream of
single-point GPU evaluations; a pool would only add pickling).
"""
import numpy as np

from . import constants
from .grid import DenseGrid, initial_grid, optimize_grid


class CoverageEstimator:
    ERROR_RATE = 1  # index of the parameter that err_scale applies to

    def __init__(self, model, err_scale=1, fix=None):
        self.model = model
        self.fix = fix
        self.err_scale = err_scale
        bounds = [tuple(b) for b in model.bounds]
        lo, hi = bounds[self.ERROR_RATE]
        bounds[self.ERROR_RATE] = (lo, hi * err_scale)
        self.bounds = bounds

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
        return -self.model.compute_loglikelihood(*self._model_args(x))

    def negll_grid(self, axes, kernel="auto"):
        """likelihood_f over itertools.product(*axes) -- the map of covest/grid.py:59-64 -- as ONE
        dense-grid evaluation.  Returns an ndarray in product order."""
        axes = [list(a) for a in axes]
        if self.fix is not None:
            axes = [a if f is None else [f] * len(a) for a, f in zip(axes, self.fix)]
        axes[self.ERROR_RATE] = [v / self.err_scale for v in axes[self.ERROR_RATE]]
        grid = DenseGrid(self.model, axes)
        try:
            grid.evaluate(kernel=kernel)
            return -grid.loglikelihoods()
        finally:
            grid.close()

    # ------------------------------------------------------------------ refinement
    def _optimize(self, start):
        """One bounded quasi-Newton refinement with finite-difference gradients, the reference's
        choice (method and options of covest/covest.py:33-39)."""
        from scipy.optimize import minimize
        return minimize(self.likelihood_f, start, method=constants.OPTIMIZATION_METHOD,
                        bounds=self.bounds, options={'disp': False})

    def _best_of(self, starts):
        """Refine every start; keep the first result with the strictly smallest objective."""
        best = None
        for res in map(self._optimize, starts):
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
