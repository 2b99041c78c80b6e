"""ctypes front-end of the parity oracle (oracle/covest_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of covest_oracle.c.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
covest_amd/ (the product) never does.

Parity status: pinned by tests/golden/*.json (generated from the reference by
tests/golden/make_golden.py) -- tests/test_oracle_golden.py.

The classes mirror the constructor of the reference models
(covest/models.py:19-31 BasicModel.__init__, :175-183 RepeatsModel.__init__) so
that tests can build the oracle and the HIP-backed model from the same arguments.
"""
import ctypes
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# COVEST_ORACLE_LIB: another build of the same source, e.g. the sanitizer build `make -C oracle
# libcovest_oracle_asan.so` (tests/test_oracle_sanitizers.py); it is loaded as is, never rebuilt here
_LIB_PATH = os.environ.get("COVEST_ORACLE_LIB") or os.path.join(_HERE, "libcovest_oracle.so")
_MAX_PARAMS = 5


class _OracleModelStruct(ctypes.Structure):
    _fields_ = [
        ("kind", ctypes.c_int),
        ("k", ctypes.c_int),
        ("r", ctypes.c_int),
        ("n_err", ctypes.c_int),
        ("comb", ctypes.POINTER(ctypes.c_double)),
        ("n_keys", ctypes.c_int64),
        ("keys", ctypes.POINTER(ctypes.c_int32)),
        ("counts", ctypes.POINTER(ctypes.c_double)),
        ("tail", ctypes.c_double),
        ("lo", ctypes.c_double * _MAX_PARAMS),
        ("hi", ctypes.c_double * _MAX_PARAMS),
        ("threshold", ctypes.c_double),
        ("has_threshold", ctypes.c_int),
    ]


def build(force=False):
    """Compile the C restatement (gcc, seconds).  Building the checker is not using it."""
    src = os.path.join(_HERE, "covest_oracle.c")
    if os.environ.get("COVEST_ORACLE_LIB"):
        return _LIB_PATH
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= os.path.getmtime(src)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "libcovest_oracle.so"],
                          stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        L.oracle_truncated_poisson.restype = ctypes.c_double
        L.oracle_truncated_poisson.argtypes = [ctypes.c_double, ctypes.c_int]
        L.oracle_threshold_o.restype = ctypes.c_int
        L.oracle_threshold_o.argtypes = [ctypes.c_double] * 4 + [ctypes.c_int, ctypes.c_int]
        P = ctypes.POINTER(_OracleModelStruct)
        DP = ctypes.POINTER(ctypes.c_double)
        L.oracle_probabilities.restype = None
        L.oracle_probabilities.argtypes = [P, DP, DP]
        L.oracle_loglikelihood.restype = ctypes.c_double
        L.oracle_loglikelihood.argtypes = [P, DP]
        L.oracle_loglikelihood_many.restype = None
        L.oracle_loglikelihood_many.argtypes = [P, ctypes.c_int64, DP, DP, ctypes.c_int]
        L.oracle_loglikelihood_many_fast.restype = None
        L.oracle_loglikelihood_many_fast.argtypes = [P, ctypes.c_int64, DP, DP, ctypes.c_int]
        L.oracle_first_min.restype = ctypes.c_int64
        L.oracle_first_min.argtypes = [DP, ctypes.c_int64, ctypes.c_double, DP]
        _lib = L
    return _lib


def truncated_poisson(l, j):
    """covest_poisson.truncated_poisson (c_src/covest_poissonmodule.c:7-35)."""
    return lib().oracle_truncated_poisson(float(l), int(j))


def threshold_o(q1, q2, q, threshold, hist_max):
    """RepeatsModel.get_hist_threshold (covest/models.py:185-191)."""
    has = 0 if threshold is None else 1
    return lib().oracle_threshold_o(float(q1), float(q2), float(q),
                                    0.0 if threshold is None else float(threshold), has,
                                    int(hist_max))


def comb_table(k):
    """self.comb of covest/models.py:25: comb(k, s) * 3**s for s in 0..k.

    The reference calls scipy.misc.comb, which scipy 1.15 ships as
    scipy.special.comb (same float routine)."""
    from scipy.special import comb
    return [comb(k, s) * (3 ** s) for s in range(k + 1)]


def _dp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


class OracleModel:
    """Oracle counterpart of covest.models.BasicModel / RepeatsModel."""

    def __init__(self, kind, k, r, hist, tail, max_error=None, max_cov=None, threshold=1e-8,
                 min_single_copy_ratio=0.3):
        assert kind in ("basic", "repeats")
        self.kind = kind
        self.k, self.r, self.hist, self.tail = k, r, hist, tail
        self.max_error = k + 1 if max_error is None else min(k + 1, max_error)
        if kind == "basic":
            self.bounds = ((0.01, max_cov), (0, 0.5))
        else:  # covest/models.py:177 drops max_cov for the repeats model
            self.bounds = ((0.01, None), (0, 0.5), (min_single_copy_ratio, 1), (0, 1), (0, 1))
        self.threshold = threshold
        self._comb = np.asarray(comb_table(k)[: self.max_error], dtype=np.float64)
        self._keys = np.asarray(list(hist.keys()), dtype=np.int32)
        self._counts = np.asarray([float(v) for v in hist.values()], dtype=np.float64)
        st = _OracleModelStruct()
        st.kind = 0 if kind == "basic" else 1
        st.k, st.r, st.n_err = k, r, self.max_error
        st.comb = _dp(self._comb)
        st.n_keys = len(self._keys)
        st.keys = self._keys.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
        st.counts = _dp(self._counts)
        st.tail = float(tail)
        for i in range(_MAX_PARAMS):
            lo, hi = self.bounds[i] if i < len(self.bounds) else (None, None)
            st.lo[i] = math.nan if lo is None else float(lo)
            st.hi[i] = math.nan if hi is None else float(hi)
        st.threshold = 0.0 if threshold is None else float(threshold)
        st.has_threshold = 0 if threshold is None else 1
        self._st = st

    @property
    def param_count(self):
        return 2 if self.kind == "basic" else 5

    def compute_probabilities(self, *args):
        """After fit_to_bounds, as compute_loglikelihood calls it (covest/models.py:101-102)."""
        par = np.asarray(args[: self.param_count], dtype=np.float64)
        out = np.empty(len(self._keys), dtype=np.float64)
        lib().oracle_probabilities(ctypes.byref(self._st), _dp(par), _dp(out))
        return dict(zip(self._keys.tolist(), out.tolist()))

    def compute_loglikelihood(self, *args):
        par = np.asarray(args[: self.param_count], dtype=np.float64)
        return lib().oracle_loglikelihood(ctypes.byref(self._st), _dp(par))

    def compute_loglikelihood_many_fast(self, points, n_threads=1):
        """The log-domain 'fast CPU mode' (see covest_oracle.c); for the baseline report only."""
        pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, self.param_count)
        out = np.empty(len(pts), dtype=np.float64)
        lib().oracle_loglikelihood_many_fast(ctypes.byref(self._st), len(pts), _dp(pts), _dp(out),
                                             int(n_threads))
        return out

    def compute_loglikelihood_many(self, points, n_threads=1):
        pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, self.param_count)
        out = np.empty(len(pts), dtype=np.float64)
        lib().oracle_loglikelihood_many(ctypes.byref(self._st), len(pts), _dp(pts), _dp(out),
                                        int(n_threads))
        return out


def first_min(vals, start=math.inf):
    """Strict-<, first-wins scan of covest/grid.py:65-70; (-1, start) if nothing wins."""
    v = np.ascontiguousarray(vals, dtype=np.float64)
    best = ctypes.c_double(0.0)
    arg = lib().oracle_first_min(_dp(v), len(v), float(start), ctypes.byref(best))
    return int(arg), best.value
