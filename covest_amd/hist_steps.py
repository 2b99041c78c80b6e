"""The steps before the likelihood path, on the GPU where they are heavy.

Counterparts, same names and argument meaning, of the reference's
  covest/data.py:22-41,176-182   load_histogram, save_histogram            (text format of `.hist` files)
  covest/histogram.py:12-45      compute_coverage_apx                      (first guess of c and e)
  covest/histogram.py:47-75      sample_histogram                          (down-sampling by `factor`)
  covest/histogram.py:78-105     auto_sample_hist
  covest/histogram.py:108-136    remove_noise, get_trim, trim_hist
  covest/histogram.py:139-166    process_histogram

Only sample_histogram is worth a kernel: its expected counts are an O(B^2) sum of thinning
probabilities (K-thin, csrc/thin_hist.hip, through covest_thin_histogram of the C ABI -- no CPU
fallback).  The rest is scalar host arithmetic, written so that it returns the reference's values.

Deliberate divergence: for source counts i with i / factor > 200 the reference's
poisson_dist is wrong (c_src/covest_poissonmodule.c:88-99 rescales the rate itself); K-thin computes the
Poisson pmf.  The randomised rounding draws from `rng` (default random.random, unseeded, as the reference).
"""
import ctypes
import math
import random

import numpy as np

from . import _capi

MAX_NOTRIM = 25                    # covest/constants.py:21
AUTO_SAMPLE_TARGET_COVERAGE = 12   # :17
AUTO_TRIM_PRECISION = 6            # :18
NOISE_THRESHOLD = 10 ** -6         # :19


class InvalidFormatException(Exception):
    def __init__(self, fname):
        self.fname = fname

    def __str__(self):
        return 'Unable to parse %s. Unsupported format.' % self.fname


# ------------------------------------------------------------------ `.hist` files
def load_histogram(fname):
    """(hist, meta): `count multiplicity` per line, `#key:value` lines are metadata."""
    hist, meta = {}, {}
    with open(fname) as f:
        for line in f:
            if line.startswith('#'):
                key, value = line[1:].strip().split(':')
                meta[key] = value
                continue
            fields = line.split()
            try:
                hist[int(fields[0])] = int(fields[1])
            except ValueError:
                raise InvalidFormatException(fname)
    return hist, meta


def save_histogram(hist, fname, meta=None):
    with open(fname, 'w') as f:
        for key, value in (meta or {}).items():
            f.write('#%s:%s\n' % (key, value))
        for count, n in hist.items():
            f.write('%d %d\n' % (count, n))


# ------------------------------------------------------------------ first guess
def _fix_coverage(coverage):
    """Invert c -> mean of a zero-and-one-truncated Poisson by Newton's method with a forward-difference
    slope (step and stopping rule 1e-8, start at coverage / 2), as covest/utils.py:51-52 does."""
    step = 1e-8

    def miss(c):
        return (c - c * math.exp(-c)) / (1 - math.exp(-c) - c * math.exp(-c)) - coverage

    x = float(coverage) / 2
    while True:
        delta = miss(x) / ((miss(x + step) - miss(x)) / step)
        if not abs(delta) > step:
            return x
        x -= delta


def compute_coverage_apx(hist, k, r):
    """(coverage, error rate) from the moments of the histogram without its first column."""
    singletons = hist.get(1, 0)
    distinct = sum(hist.values())
    if distinct == 0:
        return 0.0, 1.0
    occurrences = sum(i * n for i, n in hist.items()) - singletons
    distinct_multi = distinct - singletons
    try:
        cov = _fix_coverage(occurrences / distinct_multi)
        genomic = distinct_multi / (1.0 - math.exp(-cov) - cov * math.exp(-cov))
        genomic_once = genomic * cov * math.exp(-cov)
        genomic_never = genomic * math.exp(-cov)
        alpha = max(0.0, singletons - genomic_once) / (distinct + genomic_never)
        p_correct = max(0.0, (cov * (alpha - 1)) / (alpha * cov - alpha - cov))
        err = 1 - p_correct ** (1.0 / k)
        if p_correct > 0:
            return float((cov / p_correct) * r / (r - k + 1)), float(err)
        return 0.0, float(err)
    except ZeroDivisionError:
        return 0.0, 1.0


# ------------------------------------------------------------------ trimming
def remove_noise(hist):
    total = sum(hist.values())
    return {i: n for i, n in hist.items() if n / total > NOISE_THRESHOLD}


def get_trim(hist, ignore_last=False):
    """Smallest count at which the cumulative share of the de-noised histogram rounds to 1."""
    hist = remove_noise(hist)
    whole = float(sum(hist.values()))
    if ignore_last:
        whole -= hist[max(hist)]
    seen = 0.0
    for i in sorted(hist):
        seen += hist[i]
        if round(seen / whole, AUTO_TRIM_PRECISION) >= 1:
            return i
    return max(hist)


def trim_hist(hist, threshold):
    """(histogram below `threshold` without empty bins, mass at or above it)."""
    if threshold >= max(hist):
        return hist, 0
    tail = sum(n for i, n in hist.items() if i >= threshold)
    return {i: n for i, n in hist.items() if i < threshold and n > 0}, tail


# ------------------------------------------------------------------ down-sampling (K-thin)
def expected_sampled(hist, factor, device=-1):
    """Expected counts {j: value, j = 1..max key} after thinning by `factor` -- one GPU launch."""
    if not factor > 1:
        raise ValueError('sample factor must be > 1')
    if not hist:
        return {}
    keys = np.fromiter(hist.keys(), dtype=np.int32, count=len(hist))
    counts = np.fromiter((float(v) for v in hist.values()), dtype=np.float64, count=len(hist))
    top = int(keys.max())
    out = np.empty(top, dtype=np.float64)
    _capi.check(_capi.lib().covest_thin_histogram(
        int(device), len(keys), keys.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
        counts.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), float(factor), top,
        out.ctypes.data_as(ctypes.POINTER(ctypes.c_double))), 'covest_thin_histogram')
    return {j + 1: float(v) for j, v in enumerate(out)}


def sample_histogram(hist, factor=2, trim=None, rng=None, device=-1):
    """The histogram of a `factor`-fold down-sampled read set: expected counts on the GPU, then the
    reference's randomised rounding (round up with probability value - round(value))."""
    if trim is None:
        trim = get_trim(hist) if len(hist) > 300 else max(hist)
    else:
        trim = min(max(hist), trim * factor)
    rng = rng or random.random
    sampled = {}
    for j, value in expected_sampled({i: n for i, n in hist.items() if i < trim}, factor, device).items():
        up = rng() < value - round(value)
        sampled[j] = math.ceil(value) if up else math.floor(value)
    return {j: n for j, n in sampled.items() if n > 0}


def auto_sample_hist(hist, k, r, trim=None, rng=None, device=-1):
    """Smallest sample factor (doubling search, then bisection) whose sampled histogram has an
    approximate coverage at most AUTO_SAMPLE_TARGET_COVERAGE: (hist, factor, c, e)."""
    best, factor, stride = dict(hist), 1, 1
    c, e = compute_coverage_apx(hist, k, r)
    while c > AUTO_SAMPLE_TARGET_COVERAGE:
        factor += stride
        stride *= 2
        best = sample_histogram(hist, factor=factor, trim=trim, rng=rng, device=device)
        c, e = compute_coverage_apx(best, k, r)
    stride //= 4
    probe = factor - stride
    while stride >= 1:
        trial = sample_histogram(hist, factor=probe, trim=trim, rng=rng, device=device)
        c, e = compute_coverage_apx(trial, k, r)
        if c > AUTO_SAMPLE_TARGET_COVERAGE:
            probe += stride
        else:
            best, factor = trial, probe
            probe -= stride
        stride //= 2
    return best, factor, c, e


def process_histogram(hist, k, r, trim=None, sample_factor=None, max_notrim=MAX_NOTRIM, rng=None, device=-1):
    """(hist, tail, sample_factor, guessed c, guessed e): optional down-sampling, then trimming."""
    hist = dict(hist)
    tail = 0
    if sample_factor is not None and sample_factor > 1:
        hist = sample_histogram(hist, sample_factor, trim, rng=rng, device=device)
    if sample_factor is None and max(hist) > max_notrim:
        hist, sample_factor, c, e = auto_sample_hist(hist, k, r, trim=trim, rng=rng, device=device)
    else:
        c, e = compute_coverage_apx(hist, k, r)
        if sample_factor is None:
            sample_factor = 1
    if trim is None:
        if max(hist) > max_notrim:
            hist, tail = trim_hist(hist, get_trim(hist, ignore_last=True))
    elif trim > 0:
        hist, tail = trim_hist(hist, trim)
    return hist, tail, sample_factor, c, e
