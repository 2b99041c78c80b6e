"""Host-side mirror of covest/models.py for the MI355X likelihood path.

`BasicModel` and `RepeatsModel` take the reference's constructor arguments and
expose the attribute surface `covest.covest.main`, `CoverageEstimator` and
`covest.data.print_output` touch (params, param_count, bounds, defaults, hist,
tail, k, r, repeats, correct_c, short_name, fit_to_bounds, compute_probabilities,
compute_loglikelihood, compute_loglikelihood_multi), but every likelihood value
is computed by the gfx950 kernels behind include/covest_amd.h.  There is no CPU
fallback: without the HIP library or a GPU the compute methods raise.

Reference: covest/models.py (BasicModel :17-170, RepeatsModel :173-242,
registry :245-259).  plot_probs (:119-170, matplotlib UI) is out of scope.
"""
import ctypes
import inspect
import math
import sys

import numpy as np

from . import _capi, constants

MODEL_CLASS_SUFFIX = 'Model'
_DP = ctypes.POINTER(ctypes.c_double)


_COMB_TABLES = {}


def _comb_float(n, k):
    """scipy.special.comb(n, k) (exact=False) -- what the reference's `comb` returns today (covest/models.py:10,25).
    For integer arguments with min(k, n - k) < 20, which is every entry a model with k <= 39 has and the first 20 of
    any model, scipy's routine (xsf binom) is a short product loop in double precision; restated here so that a model
    does not import scipy.special -- 0.2 to 0.7 s, most of a first search's time-to-argmin.  Bit-equal to scipy for
    every n < 200 (checked when this was written; tests/test_host_logic.py keeps checking).  Other arguments go to
    scipy itself."""
    kx = k
    if n > 0 and kx > n // 2:
        kx = n - kx
    if 0 <= kx < 20:
        num, den = 1.0, 1.0
        for i in range(1, kx + 1):
            num *= i + n - kx
            den *= i
            if abs(num) > 1e50:
                num /= den
                den = 1.0
        return num / den
    from scipy.special import comb
    return float(comb(n, k))


def _comb_table(k):
    # covest/models.py:25 -- comb(k, s) * 3 ** s for s = 0 .. k; one table per k and process
    if k not in _COMB_TABLES:
        _COMB_TABLES[k] = tuple(_comb_float(k, s) * (3 ** s) for s in range(k + 1))
    return list(_COMB_TABLES[k])


def _as_dp(a):
    return a.ctypes.data_as(_DP)


class BasicModel:
    """covest/models.py:17-117 -- (coverage, error_rate) truncated-Poisson mixture."""
    params = ('coverage', 'error_rate')
    _kind = _capi.MODEL_BASIC

    def __init__(self, k, r, hist, tail, max_error=None, max_cov=None, *args, **kwargs):
        self.repeats = False
        self.k = k
        self.r = r
        self.bounds = ((0.01, max_cov), (0, 0.5))
        self.defaults = (1, self._default_param(1))
        self.comb = _comb_table(k)
        # `hist`: the reference's dict {j: count} (covest/models.py:26) -- or, for callers that hold the histogram as
        # arrays already (a 10 000-key dict costs 0.3 ms to walk, a third of a warm C3 search), a pair
        # (keys, counts) of equal-length sequences in dict order; `self.hist` then builds the dict on first use
        if isinstance(hist, tuple) and len(hist) == 2 and not isinstance(hist[0], (int, float)):
            self._keys = np.ascontiguousarray(hist[0], dtype=np.int32)
            self._counts = np.ascontiguousarray(hist[1], dtype=np.float64)
            if self._keys.shape != self._counts.shape or self._keys.ndim != 1:
                raise ValueError('hist as arrays: (keys, counts) of one length')
            self._hist = None
        else:
            self._keys = self._counts = None
            self._hist = hist
        self.tail = tail
        if max_error is None:
            self.max_error = self.k + 1
        else:
            self.max_error = min(self.k + 1, max_error)
        self.device = kwargs.get('device', -1)  # HIP ordinal; -1 = current device
        self._handle = None

    # ------------------------------------------------------------------ surface
    @property
    def hist(self):
        if self._hist is None:
            self._hist = {int(j): (int(v) if float(v).is_integer() else float(v))
                          for j, v in zip(self._keys.tolist(), self._counts.tolist())}
        return self._hist

    @hist.setter
    def hist(self, value):
        self._hist = value
        self._keys = self._counts = None

    @classmethod
    def short_name(cls):
        name = cls.__name__
        if name.endswith(MODEL_CLASS_SUFFIX):
            name = name[:-len(MODEL_CLASS_SUFFIX)]
        return name.lower()

    @property
    def param_count(self):
        return len(self.params)

    def _default_param(self, i, default=None):
        lo, hi = self.bounds[i]
        if lo is None or hi is None:
            return default
        return (lo + hi) / 2

    def check_bounds(self, args):
        """covest/models.py:50-58 (dead code upstream, kept for surface parity)."""
        for arg, (lo, hi) in zip(args, self.bounds):
            if arg is None:
                continue
            if (lo is not None and arg < lo) or (hi is not None and arg > hi):
                return False
        return True

    def fit_to_bounds(self, args):
        """covest/models.py:60-69.  The kernels apply the same clamp on the device."""
        args = list(args)
        for i, (arg, (lo, hi)) in enumerate(zip(args, self.bounds)):
            if arg is None:
                continue
            if lo is not None and arg < lo:
                args[i] = lo
            elif hi is not None and arg > hi:
                args[i] = hi
        return args

    def correct_c(self, c):
        """covest/models.py:71-72."""
        return c * (self.r - self.k + 1) / self.r

    # ------------------------------------------------------------------ handle
    def _threshold(self):
        return None

    def _desc(self):
        if self._keys is not None:
            keys, counts = self._keys, self._counts
        else:
            n = len(self._hist)
            keys = np.fromiter(self._hist.keys(), dtype=np.int32, count=n)  # (dict order: covest/models.py:92-97 walks it)
            counts = np.fromiter(self._hist.values(), dtype=np.float64, count=n)
        comb = np.asarray(self.comb[:self.max_error], dtype=np.float64)
        d = _capi.ModelDesc()
        d.kind = self._kind
        d.k, d.r, d.n_err = int(self.k), int(self.r), int(self.max_error)
        d.comb = _as_dp(comb)
        d.n_keys = len(keys)
        d.keys = keys.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
        d.counts = _as_dp(counts)
        d.tail = float(self.tail)
        for i in range(_capi.MAX_PARAMS):
            lo, hi = self.bounds[i] if i < len(self.bounds) else (None, None)
            d.lo[i] = math.nan if lo is None else float(lo)
            d.hi[i] = math.nan if hi is None else float(hi)
        thr = self._threshold()
        d.threshold = 0.0 if thr is None else float(thr)
        d.has_threshold = 0 if thr is None else 1
        d.device = int(self.device)
        return d, (keys, counts, comb)  # keep the arrays alive during create

    @property
    def handle(self):
        """The covest_model* of include/covest_amd.h, created on first use (never at
        import or unpickle time: no HIP initialisation before a fork)."""
        if self._handle is None:
            L = _capi.lib()
            desc, keep = self._desc()
            h = ctypes.c_void_p()
            _capi.check(L.covest_model_create(ctypes.byref(desc), ctypes.byref(h)),
                        "covest_model_create")
            del keep
            self._handle = h
        return self._handle

    def on_device(self, device):
        """This model bound to HIP device `device`: itself where that is its device already, else a copy of its plain
        data (what pickling ships to a Pool worker) whose handle is created on that device on first use.  How ONE
        process puts the same histogram on several GPUs (covest_amd.grid.dense_grid_argmin(devices=...))."""
        if int(device) == int(self.device):
            return self
        clone = self.__class__.__new__(self.__class__)
        clone.__setstate__(self.__getstate__())
        clone.device = int(device)
        return clone

    def _register_grid(self, grid):
        """A grid handle borrows its model (include/covest_amd.h): the model closes the grids still open on it
        before it goes (DenseGrid calls this)."""
        import weakref
        if not hasattr(self, '_grids'):
            self._grids = []
        self._grids = [g for g in self._grids if g() is not None]
        self._grids.append(weakref.ref(grid))

    def close(self):
        for ref in getattr(self, '_grids', []):
            grid = ref()
            if grid is not None:
                grid.close()
        self._grids = []
        if getattr(self, '_handle', None) is not None:
            _capi.lib().covest_model_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __getstate__(self):
        # Pool workers receive plain data and re-open the library lazily
        # (covest/grid.py:48 and covest/covest.py:68 pickle bound methods of the model).
        state = dict(self.__dict__)
        state['_handle'] = None
        state.pop('_grids', None)
        return state

    def __setstate__(self, state):
        self.__dict__.update(state)
        self._handle = None

    @property
    def bins_evaluated(self):
        return int(_capi.lib().covest_model_bins_evaluated(self.handle))

    # ------------------------------------------------------------------ compute
    def _points_array(self, args_list):
        n_par = self.param_count
        pts = np.empty((len(args_list), n_par), dtype=np.float64)
        for i, a in enumerate(args_list):
            pts[i, :] = [float(v) for v in list(a)[:n_par]]
        return pts

    def loglikelihood_points(self, points, kernel="auto"):
        """LL for an (n, param_count) array of points -> ndarray (n,)."""
        pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, self.param_count)
        out = np.empty(len(pts), dtype=np.float64)
        if len(pts):
            _capi.check(_capi.lib().covest_eval_points(self.handle, len(pts), _as_dp(pts), _as_dp(out),
                                                       _capi.KERNELS[kernel]),
                        "covest_eval_points")
        return out

    def reference_overflows(self, points):
        """Where the REFERENCE's own evaluation would overflow to inf / NaN (its long-double pmf product is
        formed before it is scaled, c_src/covest_poissonmodule.c:19-24) while this library returns the finite
        value: boolean ndarray, one entry per point of an (n, param_count) array.  Host arithmetic only."""
        pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, self.param_count)
        out = np.zeros(len(pts), dtype=np.uint8)
        if len(pts):
            _capi.check(_capi.lib().covest_reference_overflow(
                self.handle, len(pts), _as_dp(pts), out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))),
                "covest_reference_overflow")
        return out.astype(bool)

    def compute_probabilities(self, *args, clamp=False):
        """covest/models.py:81-98 (:211-242 for repeats): {j: p_j} for every key."""
        par = np.asarray([float(v) for v in args[:self.param_count]], dtype=np.float64)
        out = np.empty(len(self.hist), dtype=np.float64)
        _capi.check(_capi.lib().covest_probabilities(self.handle, _as_dp(par), 1 if clamp else 0,
                                                     _as_dp(out)), "covest_probabilities")
        return dict(zip(self.hist.keys(), out.tolist()))

    def compute_loglikelihood(self, *args):
        """covest/models.py:100-107."""
        return float(self.loglikelihood_points(self._points_array([args]))[0])

    def compute_loglikelihood_multi(self, args_list, thread_count=constants.DEFAULT_THREAD_COUNT):
        """covest/models.py:109-117: {tuple(args): LL}.  One batched kernel launch
        replaces Pool.starmap; thread_count (same default as the reference, None included) is
        accepted and ignored."""
        args_list = list(args_list)
        lls = self.loglikelihood_points(self._points_array(args_list))
        return {tuple(args): float(ll) for args, ll in zip(args_list, lls)}


class RepeatsModel(BasicModel):
    """covest/models.py:173-242 -- adds the copy-number mixture b_o(q1, q2, q)."""
    params = BasicModel.params + ('q1', 'q2', 'q')
    _kind = _capi.MODEL_REPEATS

    def __init__(self, k, r, hist, tail, max_error=None, max_cov=None, threshold=1e-8,
                 min_single_copy_ratio=0.3, *args, **kwargs):
        # covest/models.py:177 does not forward max_cov: the repeats coverage bound is (0.01, None)
        super(RepeatsModel, self).__init__(k, r, hist, tail, max_error=max_error, **kwargs)
        self.repeats = True
        self.bounds = self.bounds + ((min_single_copy_ratio, 1), (0, 1), (0, 1))
        self.defaults = self.defaults + tuple(
            self._default_param(i, default=0.5) for i in range(2, 5)
        )
        self.threshold = threshold

    def _threshold(self):
        return self.threshold

    def get_hist_threshold_values(self, q123):
        """threshold_o (covest/models.py:185-208) for an (n, 3) array of (q1, q2, q).
        Host-side (libm pow as CPython); needs no GPU."""
        q = np.ascontiguousarray(q123, dtype=np.float64).reshape(-1, 3)
        out = np.empty(len(q), dtype=np.int32)
        thr = self.threshold
        _capi.check(_capi.lib().covest_threshold_o(
            len(q), _as_dp(q), 0.0 if thr is None else float(thr), 0 if thr is None else 1,
            int(self._keys.max()) if self._keys is not None else int(max(self.hist)), out.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))),
            "covest_threshold_o")
        return out


models = {
    cls.short_name(): cls for _, cls in inspect.getmembers(
        sys.modules[__name__],
        predicate=lambda x: inspect.isclass(x) and x.__name__.endswith(MODEL_CLASS_SUFFIX)
    )
}


def select_model(m):
    """covest/models.py:252-259: exact name, then prefix; ValueError otherwise."""
    if m in models:
        return models[m]
    for name, model in models.items():
        if name.startswith(m):
            return model
    raise ValueError('Not such model: {}.'.format(m))
