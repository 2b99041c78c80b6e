"""CoverageEstimator: the adapter between optimiser space and model space
(covest/covest.py:18-96), over a GPU-backed model.

`likelihood_f` keeps the reference contract (apply `fix`, divide x[1] by
`err_scale`, return -LL) so scipy's L-BFGS-B and `optimize_grid` see the same
function; `negll_grid` is its batched form used by covest_amd.grid.optimize_grid.
"""
import numpy as np

from . import constants
from .grid import DenseGrid, initial_grid, optimize_grid


class CoverageEstimator:
    def __init__(self, model, err_scale=1, fix=None):
        # covest/covest.py:19-24
        self.model = model
        self.fix = fix
        self.err_scale = err_scale
        self.bounds = list(self.model.bounds)
        self.bounds[1] = self.bounds[1][0], self.bounds[1][1] * self.err_scale

    def likelihood_f(self, x):
        # covest/covest.py:26-31
        args = list(x)
        if self.fix is not None:
            args = [j if self.fix[i] is None else self.fix[i] for i, j in enumerate(args)]
        args[1] /= self.err_scale
        return -self.model.compute_loglikelihood(*args)

    def negll_grid(self, axes, kernel="auto"):
        """-LL over itertools.product(*axes) (optimiser space) in ONE launch:
        likelihood_f mapped over the grid of covest/grid.py:59-64."""
        axes = [list(a) for a in axes]
        if self.fix is not None:
            axes = [a if self.fix[i] is None else [self.fix[i]] * len(a) for i, a in enumerate(axes)]
        axes[1] = [v / self.err_scale for v in axes[1]]
        grid = DenseGrid(self.model, axes)
        try:
            grid.evaluate(kernel=kernel)
            return -grid.loglikelihoods()
        finally:
            grid.close()

    def _optimize(self, r):
        # covest/covest.py:33-39 (scalar refinement)
        from scipy.optimize import minimize
        return minimize(
            self.likelihood_f, r,
            method=constants.OPTIMIZATION_METHOD,
            bounds=self.bounds,
            options={'disp': False}
        )

    def compute_coverage(self, guess, starting_points=1, use_grid_search=False,
                         n_threads=constants.DEFAULT_THREAD_COUNT):
        # covest/covest.py:41-96; multi-start runs sequentially in-process (each
        # likelihood call is a GPU launch; a Pool would only add pickling).
        r = list(guess)
        r[1] *= self.err_scale
        success = True
        try:
            if starting_points == 1:
                res = self._optimize(r)
                success = res.success
                r = res.x
            elif starting_points > 1:
                params = initial_grid(r, count=starting_points, bounds=self.bounds, fix=self.fix)
                min_r = None
                for res in [self._optimize(p) for p in params]:
                    if min_r is None or min_r > res.fun:
                        min_r = res.fun
                        success = res.success
                        r = res.x
            if use_grid_search is None and not success:
                use_grid_search = True
            if use_grid_search:
                r = list(optimize_grid(self.likelihood_f, r, bounds=self.bounds, fix=self.fix,
                                       n_threads=n_threads))
        except KeyboardInterrupt:
            pass
        r = list(r)
        r[1] /= self.err_scale
        return r, success
