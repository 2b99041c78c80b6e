"""CPU restatement of the steps around the likelihood path.

TEST INFRASTRUCTURE ONLY (see oracle/covest_oracle.c): imported by tests/ alone; covest_amd/ never does.
Parity status: PINNED by tests/golden/hist_steps.json (made by tests/golden/make_golden_hist.py from the
reference) -- tests/test_hist_oracle_golden.py.

Restated: covest/histogram.py:12-45 (compute_coverage_apx), :47-75 (sample_histogram), :108-136
(remove_noise, get_trim, trim_hist), c_src/covest_poissonmodule.c:64-108 (poisson_dist, in x87 long
double like the extension), covest/utils.py:44-55 and covest/inverse.py:21-43 (fix_coverage).
"""
import math

import numpy as np

LD = np.longdouble
MAX_EXP = 200
NOISE_THRESHOLD = 10 ** -6   # covest/constants.py:19
AUTO_TRIM_PRECISION = 6      # :18


def poisson_dist(l, max_j, faithful=True):
    """covest_poisson.poisson_dist(l, max_j): [P(j) for j = 1..max_j].

    faithful=True follows the extension statement by statement, INCLUDING its defect for l > 200: the
    rescaling loop subtracts 200 from `l` itself, so from the first j on the recurrence multiplies by the
    reduced rate while the final division still uses e^(original l) (c_src/covest_poissonmodule.c:88-99).
    faithful=False is the Poisson pmf (what the GPU path computes), in long double via lgamma."""
    if l == 0 or l != l:
        return [0.0] * max_j
    if not faithful:
        ll = LD(l)
        return [float(np.exp(LD(j) * np.log(ll) - LD(math.lgamma(j + 1)) - ll)) if j < 171 else
                float(np.exp(LD(j) * np.log(ll) - _lgamma_ld(j + 1) - ll)) for j in range(1, max_j + 1)]
    out = []
    l = float(l)
    p1 = LD(1)
    d1 = np.exp(LD(MAX_EXP))
    d2 = np.exp(LD(l))
    for j in range(1, max_j + 1):
        p1 = p1 * LD(l / j)          # `l / j` is a double division in the C source
        p1c = p1
        while l > MAX_EXP and p1c > 0:
            p1c = p1c / d1
            l -= MAX_EXP
        out.append(float(p1c / d2))
    return out


def _lgamma_ld(n):
    """ln (n-1)! in long double (Stirling series; n >= 10)."""
    x = LD(n)
    inv = 1 / x
    inv2 = inv * inv
    series = inv * (LD(1) / 12 - inv2 * (LD(1) / 360 - inv2 * (LD(1) / 1260 - inv2 * (LD(1) / 1680))))
    return (x - LD(0.5)) * np.log(x) - x + LD(0.5) * np.log(2 * LD(np.pi)) + series


def binom_pmf(n, p, k):
    """scipy.stats.binom(n, p).pmf(k), in long double."""
    if k < 0 or k > n:
        return 0.0
    lg = lambda m: LD(math.lgamma(m + 1)) if m < 170 else _lgamma_ld(m + 1)  # noqa: E731
    return float(np.exp(lg(n) - lg(k) - lg(n - k) + LD(k) * np.log(LD(p)) + LD(n - k) * np.log1p(-LD(p))))


def thinning_pmf(i, prob, faithful=True):
    """The list `probs` of covest/histogram.py:59-64 for a source count i."""
    if i < 100:
        return [binom_pmf(i, prob, j) for j in range(1, i + 1)]
    return poisson_dist(i * prob, i, faithful=faithful)


def sample_expected(hist, factor=2, trim=None, faithful=True):
    """The real-valued histogram of sample_histogram before its randomised rounding
    (covest/histogram.py:47-69), as {j: value} in the reference's insertion order."""
    if trim is None:
        trim = get_trim(hist) if len(hist) > 300 else max(hist)
    else:
        trim = min(max(hist), trim * factor)
    kept = {k: v for k, v in hist.items() if k < trim}
    h = {}
    prob = 1.0 / factor
    for i, v in kept.items():
        for j, p in enumerate(thinning_pmf(i, prob, faithful=faithful)):
            h[j + 1] = h.get(j + 1, 0) + v * p
    return h


def thin_expected_c(keys, counts, factor, out_len, faithful=False):
    """The C twin (oracle/covest_oracle.c: oracle_thin_expected) of the double loop of sample_expected over
    already trimmed bins: ndarray[out_len].  Used where Python loops are too slow (bench, full sizes)."""
    import ctypes
    from . import covest_oracle
    L = covest_oracle.lib()
    L.oracle_thin_expected.restype = None
    L.oracle_thin_expected.argtypes = [ctypes.c_int64, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_double),
                                       ctypes.c_double, ctypes.c_int64, ctypes.POINTER(ctypes.c_double), ctypes.c_int]
    k = np.ascontiguousarray(keys, dtype=np.int32)
    c = np.ascontiguousarray(counts, dtype=np.float64)
    out = np.zeros(int(out_len), dtype=np.float64)
    L.oracle_thin_expected(len(k), k.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                           c.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), float(factor), int(out_len),
                           out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), 1 if faithful else 0)
    return out


def round_sampled(expected, uniforms):
    """covest/histogram.py:71-75 with the uniforms `random.random()` would have returned, in order."""
    it = iter(uniforms)
    out = {}
    for i, v in expected.items():
        d = v - round(v)
        out[i] = math.ceil(v) if next(it) < d else math.floor(v)
    return {k: v for k, v in out.items() if v > 0}


def fix_coverage(coverage):
    """covest/utils.py:51-52 through covest/inverse.py:21-43 (Newton, forward-difference slope)."""
    delta = 1e-8
    f = lambda c: (c - c * math.exp(-c)) / (1 - math.exp(-c) - c * math.exp(-c))  # noqa: E731
    root = lambda x: f(x) - coverage  # noqa: E731
    slope = lambda y: (root(y + delta) - root(y)) / delta  # noqa: E731
    guess = float(coverage) / 2
    d = root(guess) / slope(guess)
    while abs(d) > delta:
        guess -= d
        d = root(guess) / slope(guess)
    return guess


def compute_coverage_apx(hist, k, r):
    """covest/histogram.py:12-45."""
    ones = hist.get(1, 0)
    all_kmers = sum(i * h for i, h in hist.items())
    total_unique = sum(h for h in hist.values())
    if total_unique == 0:
        return 0.0, 1.0
    all_kmers -= ones
    unique = total_unique - ones
    try:
        cov = fix_coverage(all_kmers / unique)
        unique /= (1.0 - math.exp(-cov) - cov * math.exp(-cov))
        est_ones = unique * cov * math.exp(-cov)
        est_zeros = unique * math.exp(-cov)
        alpha = max(0.0, ones - est_ones) / (total_unique + est_zeros)
        p_ok = max(0.0, (cov * (alpha - 1)) / (alpha * cov - alpha - cov))
        e = 1 - p_ok ** (1.0 / k)
        if p_ok > 0:
            return float((cov / p_ok) * r / (r - k + 1)), float(e)
        return 0.0, float(e)
    except ZeroDivisionError:
        return 0.0, 1.0


def remove_noise(hist):
    total = sum(hist.values())
    return {k: v for k, v in hist.items() if v / total > NOISE_THRESHOLD}


def get_trim(hist, ignore_last=False):
    """covest/histogram.py:114-127."""
    hist = remove_noise(hist)
    ss = float(sum(hist.values()))
    if ignore_last:
        ss -= hist[max(hist)]
    s = 0.0
    trim = max(hist)
    for i, h in sorted(hist.items()):
        s += h
        if round(s / ss, AUTO_TRIM_PRECISION) >= 1:
            trim = i
            break
    return trim


def trim_hist(hist, threshold):
    """covest/histogram.py:130-136."""
    if threshold >= max(hist):
        return hist, 0
    kept = {k: v for k, v in hist.items() if k < threshold and v > 0}
    return kept, sum(v for k, v in hist.items() if k >= threshold)
