#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE itself.

Runs only in the build container, where /root/reference exists.  It
  1. compiles the reference's C extension from the sources where they lie
     (`make -C oracle ref` -> oracle/_ref/, git-ignored, never copied),
  2. imports covest.models / covest.grid from /root/reference, and
  3. writes DATA ONLY (inputs + the numbers the reference returned) as JSON and
     `.hist` files next to this script.

One session-local alias is needed: covest/models.py:10 does
`from scipy.misc import comb`, which scipy >= 1.12 ships only as
`scipy.special.comb` (same routine).  Nothing of the reference is modified or
copied.  Interpreter-dependent semantics are recorded in every fixture's "env".

Usage:  python tests/golden/make_golden.py [section ...]
Sections: tp basic repeats threshold hists c1 c2 c2classes c2trim c2argmin c3 c3tail c3trim c3argmin gridtrace overflow   (default: all)
(c3argmin reads c3_candidates_gpu.json: flat indices written on the GPU box by tools/dump_c3_candidates.py;
 c2classes reads c2_classes_gpu.json: flat indices written on the GPU box by tools/dump_c2_classes.py)
"""
import itertools
import json
import math
import multiprocessing
import os
import platform
import random
import subprocess
import sys
import time

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REFERENCE = os.environ.get("COVEST_REFERENCE", "/root/reference")
REF_BUILD = os.path.join(REPO, "oracle", "_ref")


def _import_reference():
    subprocess.check_call(["make", "-C", os.path.join(REPO, "oracle"), "ref"],
                          stdout=subprocess.DEVNULL)
    os.environ.setdefault("MPLBACKEND", "Agg")
    import scipy.misc
    import scipy.special
    if not hasattr(scipy.misc, "comb"):
        scipy.misc.comb = scipy.special.comb
    sys.path.insert(0, REF_BUILD)
    sys.path.insert(0, REFERENCE)
    import covest.constants
    import covest.grid
    import covest.models
    covest.constants.VERBOSE = False
    return covest


covest = _import_reference()
from covest.models import BasicModel, RepeatsModel  # noqa: E402
from covest_poisson import truncated_poisson  # noqa: E402


def env():
    import numpy
    import scipy
    return {
        "python": platform.python_version(),
        "scipy": scipy.__version__,
        "numpy": numpy.__version__,
        "machine": platform.machine(),
        "libc": " ".join(platform.libc_ver()),
        "long_double_mant_dig": 64,
        "reference": "mhozza/covest v0.5.6",
        "note": "builtin sum() is naive left-to-right on this interpreter (< 3.12)",
    }


def dump(name, obj):
    obj = dict(obj)
    obj["env"] = env()
    path = os.path.join(HERE, name)
    with open(path, "w") as f:
        json.dump(obj, f, indent=0, separators=(",", ":"))
        f.write("\n")
    print("wrote", path, os.path.getsize(path), "bytes", flush=True)


def load_hist(path):
    """Plain reader of the `.hist` text format ("j count" lines, '#' comments):
    the format of covest/data.py:22-41, re-read here so Bio is not needed."""
    hist = {}
    with open(path) as f:
        for line in f:
            if not line.strip() or line[0] == "#":
                continue
            a, b = line.split()[:2]
            hist[int(a)] = int(b)
    return hist


def save_hist(path, hist, comment):
    with open(path, "w") as f:
        f.write("# %s\n" % comment)
        for j, v in hist.items():
            f.write("%d %d\n" % (j, v))


REF_HISTS = {
    "sim_c10_e0.05": "tests/data/simulated_c10_e0.05_r100_k21.hist",
    "sim_c10_e0.05_sparse": "tests/data/simulated_c10_e0.05_r100_k21_sparse.hist",
    "sim_c10_e0": "tests/data/simulated_c10_e0_r100_k21.hist",
}


def ref_hist(name):
    return load_hist(os.path.join(REFERENCE, REF_HISTS[name]))


# ----------------------------------------------------------------------------- (1) tp
def section_tp():
    lams = [1e-12, 1e-9, 1e-8, 2e-8, 1e-3, 0.5, 1.0, 10.0, 199.9, 200.0, 200.1, 400.0, 1e3,
            1.13e4, 1.14e4, 1.15e4, 1e5]
    js = [1, 2, 3, 10, 100, 1000, 10000]
    rows = [[l, j, truncated_poisson(l, j)] for l in lams for j in js]
    rnd = random.Random(20240521)
    for _ in range(400):
        l = math.exp(rnd.uniform(math.log(1e-6), math.log(8e3)))
        j = int(math.exp(rnd.uniform(0, math.log(9000)))) + 0
        rows.append([l, max(1, j), truncated_poisson(l, max(1, j))])
    dump("tp_table.json", {"what": "covest_poisson.truncated_poisson(l, j)", "rows": rows})


# ----------------------------------------------------------------------------- (2) basic
def basic_points(rnd):
    pts = [(10.0, 0.0), (10.0, 0.01), (12.5, 0.05), (10.0, 0.05), (0.01, 0.1), (0.001, 0.1),
           (10.0, 0.5), (10.0, 0.7), (0.01, 0.5), (1.0, 0.25), (50.0, 0.001), (3.0, 0.3),
           (-1.0, -0.1), (10.0, 1e-9), (10.019077633773197, 0.04999234428925103)]
    for _ in range(40):
        pts.append((math.exp(rnd.uniform(math.log(0.5), math.log(60))), rnd.uniform(0, 0.5)))
    return pts


def section_basic():
    rnd = random.Random(1)
    cases = []
    for hname in REF_HISTS:
        hist = ref_hist(hname)
        for max_error in (8, None):
            for tail in (0, 1000):
                m = BasicModel(21, 100, hist, tail, max_error=max_error)
                pts = basic_points(rnd)
                lls = [m.compute_loglikelihood(*p) for p in pts]
                detail = []
                for p in pts[:6]:
                    a = m.fit_to_bounds(p)
                    detail.append({"point": list(p),
                                   "lambda_s": list(m._get_lambda_s(m.correct_c(a[0]), a[1])),
                                   "p_j": [[j, v] for j, v in m.compute_probabilities(*a).items()]})
                cases.append({"hist": hname, "k": 21, "r": 100, "tail": tail,
                              "max_error": max_error, "points": [list(p) for p in pts],
                              "ll": lls, "detail": detail})
    # a max_cov bound (clamping from above) and another k/r
    hist = ref_hist("sim_c10_e0.05")
    m = BasicModel(21, 100, hist, 0, max_error=8, max_cov=8.0)
    pts = [(10.0, 0.05), (8.0, 0.05), (7.5, 0.04), (100.0, 0.6)]
    cases.append({"hist": "sim_c10_e0.05", "k": 21, "r": 100, "tail": 0, "max_error": 8,
                  "max_cov": 8.0, "points": [list(p) for p in pts],
                  "ll": [m.compute_loglikelihood(*p) for p in pts], "detail": []})
    m = BasicModel(31, 150, hist, 17, max_error=5)
    pts = basic_points(rnd)[:20]
    cases.append({"hist": "sim_c10_e0.05", "k": 31, "r": 150, "tail": 17, "max_error": 5,
                  "points": [list(p) for p in pts],
                  "ll": [m.compute_loglikelihood(*p) for p in pts], "detail": []})
    dump("basic_ll.json", {"what": "BasicModel.compute_loglikelihood", "cases": cases})


# ----------------------------------------------------------------------------- (4) repeats
def repeats_points(rnd):
    pts = [(10.0, 0.01, 0.9, 0.5, 0.5), (1.0, 0.25, 0.65, 0.5, 0.5),
           (10.0, 0.05, 0.6, 0.0, 0.3), (10.0, 0.05, 0.6, 0.5, 0.0),
           (10.0, 0.05, 0.6, 0.5, 1.0), (10.0, 0.05, 1.0, 0.5, 0.5),
           (10.0, 0.05, 0.3, 1.0, 0.5), (10.0, 0.0, 0.7, 0.4, 0.2),
           (10.0, 0.05, 0.1, 0.5, 0.5), (10.0, 0.05, 0.5, 0.5, 0.001),
           (10.007650292975498, 0.04998082334548104, 0.9990165657429007,
            0.9503033556711812, 0.6552973697071148),
           (0.001, 0.9, 0.2, 1.5, -0.2)]
    for _ in range(60):
        pts.append((math.exp(rnd.uniform(math.log(0.5), math.log(40))), rnd.uniform(0, 0.5),
                    rnd.uniform(0.3, 1.0), rnd.uniform(0, 1), rnd.uniform(0, 1)))
    return pts


def section_repeats():
    rnd = random.Random(2)
    cases = []
    for hname in REF_HISTS:
        hist = ref_hist(hname)
        for max_error, tail in ((8, 0), (8, 1000), (None, 0)):
            m = RepeatsModel(21, 100, hist, tail, max_error=max_error)
            pts = repeats_points(rnd)
            if max_error is None:
                pts = pts[:30]
            lls = [m.compute_loglikelihood(*p) for p in pts]
            detail = []
            for p in pts[:4]:
                a = m.fit_to_bounds(p)
                detail.append({"point": list(p),
                               "p_j": [[j, v] for j, v in m.compute_probabilities(*a).items()]})
            cases.append({"hist": hname, "k": 21, "r": 100, "tail": tail, "max_error": max_error,
                          "threshold": 1e-8, "min_single_copy_ratio": 0.3,
                          "points": [list(p) for p in pts], "ll": lls, "detail": detail})
    hist = ref_hist("sim_c10_e0.05")
    m = RepeatsModel(21, 100, hist, 5, max_error=8, threshold=None, min_single_copy_ratio=0.5)
    pts = repeats_points(rnd)[:20]
    cases.append({"hist": "sim_c10_e0.05", "k": 21, "r": 100, "tail": 5, "max_error": 8,
                  "threshold": None, "min_single_copy_ratio": 0.5,
                  "points": [list(p) for p in pts],
                  "ll": [m.compute_loglikelihood(*p) for p in pts], "detail": []})
    dump("repeats_ll.json", {"what": "RepeatsModel.compute_loglikelihood", "cases": cases})


# ----------------------------------------------------------------------------- (3) threshold
def section_threshold():
    rows = []
    q1s = [0.3, 0.5, 0.65, 0.9, 0.999, 1.0]
    q2s = [0.0, 0.25, 0.5, 0.95, 1.0]
    qs = [0.0, 0.001, 0.01, 0.05, 0.1, 0.3, 0.5, 0.9, 0.95, 1.0]
    for hist_max in (15, 256, 10000):
        m = RepeatsModel(21, 100, {hist_max: 1, 1: 1}, 0, max_error=8)
        for q1, q2, q in itertools.product(q1s, q2s, qs):
            rows.append([hist_max, q1, q2, q, m.get_hist_threshold(m.get_b_o(q1, q2, q), 1e-8)])
        rnd = random.Random(hist_max)
        for _ in range(300):
            q1, q2, q = rnd.uniform(0.3, 1), rnd.uniform(0, 1), rnd.uniform(0, 1)
            rows.append([hist_max, q1, q2, q, m.get_hist_threshold(m.get_b_o(q1, q2, q), 1e-8)])
    dump("threshold_o.json", {"what": "RepeatsModel.get_hist_threshold(get_b_o(q1,q2,q), 1e-8)",
                              "columns": ["hist_max", "q1", "q2", "q", "threshold_o"],
                              "rows": rows})


# ----------------------------------------------------------------------------- synthetic hists
SYNTH = {
    # name: (model, B, theta*, N)   -- SURVEY.md 8(d)
    "H256": ("basic", 256, (100.0, 0.02), 10 ** 7),
    "H10k_basic": ("basic", 10000, (4000.0, 0.02), 10 ** 7),
    "H10k_rep": ("repeats", 10000, (25.0, 0.02, 0.6, 0.5, 0.1), 10 ** 8),
}


def _tp_row(args):
    model_name, B, theta, j0, j1 = args
    keys = {j: 0 for j in range(1, B + 1)}
    cls = BasicModel if model_name == "basic" else RepeatsModel
    m = cls(21, 100, keys, 0, max_error=8)
    m.hist = {j: 0 for j in range(j0, j1)}
    if model_name == "repeats":  # threshold depends on max(hist): keep it at B
        m.get_hist_threshold = lambda b_o, thr, _m=m, _B=B: _threshold_with_max(_m, b_o, thr, _B)
    return list(m.compute_probabilities(*theta).items())


def _threshold_with_max(m, b_o, thr, hist_max):
    for o in range(1, hist_max):
        if b_o(o) <= thr:
            return o
    return hist_max


def synth_hist(name, pool):
    model_name, B, theta, N = SYNTH[name]
    path = os.path.join(HERE, name + ".hist")
    if os.path.exists(path):
        return load_hist(path)
    t0 = time.time()
    step = max(1, B // 64)
    jobs = [(model_name, B, theta, j0, min(B + 1, j0 + step)) for j0 in range(1, B + 1, step)]
    p = {}
    for part in pool.map(_tp_row, jobs):
        p.update(part)
    hist = {j: int(round(N * p[j])) for j in range(1, B + 1)}
    save_hist(path, hist, "%s: round(%d * p_j) of the reference %s model at %r, k=21 r=100 S=8, "
              "keys 1..%d" % (name, N, model_name, theta, B))
    print("synth", name, "nonzero", sum(1 for v in hist.values() if v), "sum", sum(hist.values()),
          "%.1fs" % (time.time() - t0), flush=True)
    return hist


def section_hists(pool):
    for name in SYNTH:
        synth_hist(name, pool)


# ----------------------------------------------------------------------------- (5) C1
def linspace(a, b, n):
    return [a + i * (b - a) / (n - 1) for i in range(n)]


def _ll_job(args):
    model_name, hist, tail, point = args
    cls = BasicModel if model_name == "basic" else RepeatsModel
    m = cls(21, 100, hist, tail, max_error=8)
    t0 = time.time()
    v = m.compute_loglikelihood(*point)
    return v, time.time() - t0


def _ll_sp_job(args):
    """_ll_job that also reports sp_j = fsum(p_j) as the reference's compute_loglikelihood saw it
    (covest/models.py:102-103), by listening in on its call of compute_probabilities."""
    model_name, hist, tail, point = args
    cls = BasicModel if model_name == "basic" else RepeatsModel
    m = cls(21, 100, hist, tail, max_error=8)
    seen = {}
    inner = m.compute_probabilities

    def listen(*a):
        probs = inner(*a)
        seen["sp"] = math.fsum(probs.values())
        return probs

    m.compute_probabilities = listen
    t0 = time.time()
    v = m.compute_loglikelihood(*point)
    return v, time.time() - t0, seen["sp"]


def section_c1(pool):
    hist = synth_hist("H256", pool)
    cs = [50 + i * 100 / 49 for i in range(50)]
    es = [0.001 + i * 0.099 / 49 for i in range(50)]
    grid = list(itertools.product(cs, es))
    t0 = time.time()
    res = pool.map(_ll_job, [("basic", hist, 0, p) for p in grid], chunksize=16)
    ll = [v for v, _ in res]
    best, arg = None, -1
    for i, v in enumerate(ll):
        if best is None or -v < best:
            best, arg = -v, i
    dump("c1_grid.json", {"what": "config 1: BasicModel on H256.hist, 50x50 (c,e) grid, "
                                  "itertools.product order", "hist": "H256", "k": 21, "r": 100,
                          "max_error": 8, "tail": 0, "c_axis": cs, "e_axis": es, "ll": ll,
                          "argmin_flat": arg, "min_negll": best,
                          "cpu_seconds_total_single_core": sum(t for _, t in res),
                          "wall_seconds_pool": time.time() - t0})


# ----------------------------------------------------------------------------- (6) C2 / C3 samples
def section_c2(pool):
    hist = synth_hist("H10k_basic", pool)
    cs = linspace(2000.0, 6000.0, 1000)
    es = linspace(0.001, 0.1, 1000)
    rnd = random.Random(20240521)
    idx = sorted(rnd.sample(range(10 ** 6), 256))
    pts = [(cs[i // 1000], es[i % 1000]) for i in idx]
    res = pool.map(_ll_job, [("basic", hist, 0, p) for p in pts], chunksize=4)
    dump("c2_sample.json", {"what": "config 2: BasicModel on H10k_basic.hist, 256 seeded points of "
                                    "the 1000x1000 grid c=linspace(2000,6000) e=linspace(0.001,0.1)",
                            "hist": "H10k_basic", "k": 21, "r": 100, "max_error": 8, "tail": 0,
                            "flat_index": idx, "points": [list(p) for p in pts],
                            "ll": [v for v, _ in res],
                            "cpu_seconds_per_point": [t for _, t in res]})


def c2_axes():
    return linspace(2000.0, 6000.0, 1000), linspace(0.001, 0.1, 1000)


def c2_point(i):
    cs, es = c2_axes()
    return (cs[i // 1000], es[i % 1000])


def section_c2classes(pool):
    """Config 2 where K-basic's closed form (round 3) decides: tools/dump_c2_classes.py, run on a GPU box with the
    diagnostic library, names flat indices of the C2 grid per ROUTE of the kernel -- lanes that walk key by key
    (themselves between the -inf bound and the clamp, or dragged along by their wave), closed-form lanes closest to
    the clamp and far from it, -inf-by-bound lanes closest to the bound and far from it, and the walking lanes whose
    smallest p_j is closest to half a grid step of the doubles (where the reference's term-by-term roundings decide
    between 4.94e-324 and 0, i.e. between a finite value and -inf).  Only the indices are taken
    from that file; the values are the reference's (BasicModel.compute_loglikelihood, covest/models.py:81-107)."""
    hist = synth_hist("H10k_basic", pool)
    src = os.environ.get("COVEST_C2_CLASSES", os.path.join(HERE, "c2_classes_gpu.json"))
    with open(src) as f:
        picked = json.load(f)
    classes = ["walk_self", "walk_dragged", "closed_edge", "closed_far", "neginf_edge", "neginf_far", "flush_edge"]
    idx = sorted(set(int(i) for c in classes for i in picked[c]))
    res = pool.map(_ll_job, [("basic", hist, 0, c2_point(i)) for i in idx], chunksize=2)
    at = {i: k for k, i in enumerate(idx)}
    for c in classes:
        vals = [res[at[int(i)]][0] for i in picked[c]]
        print("c2classes", c, len(vals), "finite", sum(1 for v in vals if math.isfinite(v)),
              "-inf", sum(1 for v in vals if v == -math.inf), flush=True)
    dump("c2_classes.json", {"what": "config 2 at K-basic's class boundaries: BasicModel on H10k_basic.hist at flat "
                                     "indices of the 1000x1000 grid chosen per route of the kernel "
                                     "(tools/dump_c2_classes.py); values: the reference's",
                             "hist": "H10k_basic", "k": 21, "r": 100, "max_error": 8, "tail": 0,
                             "class_counts_gpu": picked.get("class_counts"),
                             "classes": {c: [int(i) for i in picked[c]] for c in classes},
                             "flat_index": idx, "points": [list(c2_point(i)) for i in idx],
                             "ll": [v for v, _ in res]})


def section_c2trim(pool):
    """Config 2 on the histogram the reference's own pipeline would hand the model (as section c3trim): H10k_basic
    trimmed by get_trim(ignore_last=True) / trim_hist (covest/histogram.py:105-134), tail = the trimmed mass.
    COVEST_C2TRIM_POINTS (default 1024) seeded points of the C2 grid with the reference's LL and sp_j, and the
    reference's values at the best 96 points of the grid + the axis neighbours of the best one (chosen with the
    oracle's log-domain mode; the reference decides the winner under the scan of covest/grid.py:65-70)."""
    import covest.histogram as H
    import numpy as np
    hist = synth_hist("H10k_basic", pool)
    trim = H.get_trim(hist, ignore_last=True)
    thist, tail = H.trim_hist(hist, trim)
    save_hist(os.path.join(HERE, "H10k_basic_trim.hist"), thist,
              "H10k_basic trimmed by the reference (get_trim(ignore_last=True) = %d, trim_hist): tail = %d" % (trim, tail))
    n = 10 ** 6
    idx = sorted(random.Random(20241004).sample(range(n), int(os.environ.get("COVEST_C2TRIM_POINTS", "1024"))))
    res = pool.map(_ll_sp_job, [("basic", thist, tail, c2_point(i)) for i in idx], chunksize=8)
    sys.path.insert(0, REPO)
    from oracle import covest_oracle as orc
    om = orc.OracleModel("basic", 21, 100, thist, tail, max_error=8)
    cs, es = c2_axes()
    fast = np.concatenate([om.compute_loglikelihood_many_fast(
        np.array([(cs[ic], e) for ic in range(a, min(a + 50, 1000)) for e in es]), n_threads=8) for a in range(0, 1000, 50)])
    negll = np.where(np.isnan(fast), np.inf, -fast)
    order = np.argsort(negll, kind="stable")
    cand = set(int(i) for i in order[:96])
    top = np.unravel_index(int(order[0]), (1000, 1000))
    for d in range(2):
        for step in (-1, 1):
            j = list(top)
            j[d] += step
            if 0 <= j[d] < 1000:
                cand.add(int(np.ravel_multi_index(j, (1000, 1000))))
    cand = sorted(cand)
    cres = pool.map(_ll_sp_job, [("basic", thist, tail, c2_point(i)) for i in cand], chunksize=1)
    best, arg = None, -1
    for i, (v, _, _) in zip(cand, cres):
        if v == v and (best is None or -v < best):
            best, arg = -v, i
    print("c2trim: trim", trim, "keys", len(thist), "tail", tail, "reference winner", arg, best,
          "finite sample values", sum(1 for v, _, _ in res if math.isfinite(v)), flush=True)
    dump("c2_trim.json", {"what": "config 2 on H10k_basic trimmed as the reference's process_histogram trims it: "
                                  "BasicModel, seeded points of the 1000x1000 (c,e) grid with LL and sp_j = fsum(p_j), "
                                  "and the arg-min candidates",
                          "hist": "H10k_basic_trim", "source_hist": "H10k_basic", "trim": trim, "k": 21, "r": 100,
                          "max_error": 8, "tail": tail, "n_keys": len(thist),
                          "flat_index": idx, "ll": [v for v, _, _ in res], "sp": [sp for _, _, sp in res],
                          "cpu_seconds_total": sum(t for _, t, _ in res),
                          "candidates": {"flat_index": cand, "ll": [v for v, _, _ in cres],
                                         "sp": [sp for _, _, sp in cres],
                                         "reference_argmin_flat": arg, "reference_min_negll": best}})


def c3_axes():
    return (linspace(15.0, 30.0, 32), linspace(0.005, 0.08, 32), linspace(0.3, 0.95, 16), linspace(0.05, 0.95, 16))


def c3_point(i):
    cs, es, q1s, qs = c3_axes()
    ic, ie, iq1, iq = i // (32 * 256), (i // 256) % 32, (i // 16) % 16, i % 16
    return (cs[ic], es[ie], q1s[iq1], 0.5, qs[iq])


def _c3_eval(pool, hist, tail, idx, cache):
    """Reference LL at the given flat indices of the C3 grid; `cache` maps flat index -> (ll, seconds[, sp_j]) of
    an earlier run of this very script (the reference costs ~0.6 core-seconds per copy number and point here).
    With a tail the third list is the reference's sp_j per point (what its tail term was computed from)."""
    todo = [i for i in idx if i not in cache]
    # cheapest first would starve the pool at the end: longest (small q) first
    todo.sort(key=lambda i: c3_point(i)[4])
    res = pool.map(_ll_sp_job if tail else _ll_job, [("repeats", hist, tail, c3_point(i)) for i in todo], chunksize=1)
    got = dict(cache)
    got.update({i: r for i, r in zip(todo, res)})
    return [got[i][0] for i in idx], [got[i][1] for i in idx], [got[i][2] if tail else None for i in idx]


def _cached(name, tail):
    path = os.path.join(HERE, name)
    if not os.path.exists(path):
        return {}
    with open(path) as f:
        old = json.load(f)
    if old.get("tail") != tail:
        return {}
    out = {}
    sps = old.get("sp") or [None] * len(old["ll"])
    for i, p, v, t, sp in zip(old["flat_index"], old["points"], old["ll"], old["cpu_seconds_per_point"], sps):
        if list(c3_point(i)) == list(p) and (not tail or sp is not None):
            out[i] = (v, t, sp)
    return out


def _oracle_fast_ll(hist, tail, idx):
    """The repository's CPU oracle in its log-domain mode (oracle/covest_oracle.c), used here ONLY to choose WHERE
    the reference is asked (which points are finite, which are the best of a grid): every number that goes into a
    fixture is the reference's own."""
    sys.path.insert(0, REPO)
    from oracle import covest_oracle as orc
    import numpy as np
    om = orc.OracleModel("repeats", 21, 100, hist, tail, max_error=8)
    return om.compute_loglikelihood_many_fast(np.array([c3_point(i) for i in idx]), n_threads=8)


def section_c3(pool):
    """Seeded points of the C3 grid: the 64 of seed 20240522 plus 192 of seed 20240523 -- half
    of which are -inf, the grid being what it is (a counted key out of reach of a short weight vector) -- plus, so
    that the 1e-9 claim rests on >= 256 FINITE reference values, seeded points drawn from q <= 0.59 (the ten smallest
    values of the q axis) and kept where the likelihood is finite (pre-screened with the oracle's log-domain mode;
    the values themselves are the reference's)."""
    hist = synth_hist("H10k_rep", pool)
    n = 32 * 32 * 16 * 16
    idx = set(random.Random(20240522).sample(range(n), 64))
    for i in random.Random(20240523).sample(range(n), 400):
        if len(idx) >= 256:
            break
        idx.add(i)
    cache = _cached("c3_sample.json", 0)
    want_finite = int(os.environ.get("COVEST_C3_FINITE", "264"))
    have = sum(1 for i in idx if i in cache and math.isfinite(cache[i][0]))
    pool_idx = [i for i in random.Random(20240925).sample(range(n), 4000) if i % 16 <= 9 and i not in idx]
    screen = _oracle_fast_ll(hist, 0, pool_idx)
    for i, v in zip(pool_idx, screen):
        if have >= want_finite:
            break
        if math.isfinite(v):
            idx.add(i)
            have += 1
    idx = sorted(idx)
    ll, secs, _ = _c3_eval(pool, hist, 0, idx, cache)
    dump("c3_sample.json", {"what": "config 3: RepeatsModel on H10k_rep.hist, q2 fixed 0.5: 256 seeded points of the "
                                    "32x32x16x16 grid (c,e,q1,q) plus seeded points of its q <= 0.59 part where the "
                                    "likelihood is finite; finite values: %d" % sum(1 for v in ll if math.isfinite(v)),
                            "hist": "H10k_rep", "k": 21, "r": 100, "max_error": 8, "tail": 0,
                            "flat_index": idx, "points": [list(c3_point(i)) for i in idx],
                            "ll": ll, "cpu_seconds_per_point": secs})


def section_c3tail(pool):
    """The same grid with a tail (tail = 1000: every one of the 10 000 keys enters sp_j, covest/models.py:103-104)."""
    hist = synth_hist("H10k_rep", pool)
    idx = sorted(random.Random(20240524).sample(range(32 * 32 * 16 * 16), 48))
    ll, secs, sp = _c3_eval(pool, hist, 1000, idx, _cached("c3_tail_sample.json", 1000))
    dump("c3_tail_sample.json", {"what": "config 3 with tail = 1000: RepeatsModel on H10k_rep.hist, 48 seeded points "
                                         "of the 32x32x16x16 grid (c,e,q1,q), q2 fixed 0.5",
                                 "hist": "H10k_rep", "k": 21, "r": 100, "max_error": 8, "tail": 1000,
                                 "flat_index": idx, "points": [list(c3_point(i)) for i in idx],
                                 "ll": ll, "sp": sp, "cpu_seconds_per_point": secs})


def section_c3trim(pool):
    """Config 3 on the histogram the reference's own pipeline would hand the model: H10k_rep TRIMMED as
    process_histogram does it (covest/histogram.py:105-134: get_trim(ignore_last=True), trim_hist) -- the keys
    below the trim point with their zero-count keys dropped, tail = the trimmed mass.  Then 1 - sp_j is of the
    order tail / N (1e-4), the regime real inputs are in: the tail term tail * log(1 - sp_j)
    (covest/models.py:103-104) is WELL CONDITIONED, unlike on the untrimmed synthetic histogram where
    sp_j = 1 - a few ulp.  Writes the trimmed histogram, COVEST_C3TRIM_POINTS (default 4096) seeded points of the
    C3 grid with the reference's LL and sp_j, and the reference's values at the best 96 points of the grid and at
    the axis neighbours of the best one (chosen with the oracle's log-domain mode; the reference decides the
    winner among them under the scan of covest/grid.py:65-70)."""
    import covest.histogram as H
    hist = synth_hist("H10k_rep", pool)
    trim = H.get_trim(hist, ignore_last=True)
    thist, tail = H.trim_hist(hist, trim)
    save_hist(os.path.join(HERE, "H10k_rep_trim.hist"), thist,
              "H10k_rep trimmed by the reference (get_trim(ignore_last=True) = %d, trim_hist): tail = %d" % (trim, tail))
    n = 32 * 32 * 16 * 16
    idx = sorted(random.Random(20240926).sample(range(n), int(os.environ.get("COVEST_C3TRIM_POINTS", "4096"))))
    res = pool.map(_ll_sp_job, [("repeats", thist, tail, c3_point(i)) for i in idx], chunksize=8)
    # arg-min candidates: the best 96 of the whole grid by the oracle's log-domain mode + the 2 P axis neighbours
    import numpy as np
    fast = _oracle_fast_ll(thist, tail, range(n))
    negll = np.where(np.isnan(fast), np.inf, -fast)
    order = np.argsort(negll, kind="stable")
    cand = set(int(i) for i in order[:96])
    shape = (32, 32, 16, 16)
    top = np.unravel_index(int(order[0]), shape)
    for d in range(4):
        for step in (-1, 1):
            j = list(top)
            j[d] += step
            if 0 <= j[d] < shape[d]:
                cand.add(int(np.ravel_multi_index(j, shape)))
    cand = sorted(cand)
    cres = pool.map(_ll_sp_job, [("repeats", thist, tail, c3_point(i)) for i in cand], chunksize=1)
    best, arg = None, -1
    for i, (v, _, _) in zip(cand, cres):
        if v == v and (best is None or -v < best):
            best, arg = -v, i
    print("c3trim: trim", trim, "keys", len(thist), "tail", tail, "reference winner", arg, best,
          "finite sample values", sum(1 for v, _, _ in res if math.isfinite(v)), flush=True)
    dump("c3_trim.json", {"what": "config 3 on H10k_rep trimmed as the reference's process_histogram trims it: "
                                  "RepeatsModel, q2 fixed 0.5, seeded points of the 32x32x16x16 grid (c,e,q1,q) "
                                  "with LL and sp_j = fsum(p_j), and the arg-min candidates",
                          "hist": "H10k_rep_trim", "source_hist": "H10k_rep", "trim": trim, "k": 21, "r": 100,
                          "max_error": 8, "tail": tail, "n_keys": len(thist),
                          "axes": [list(a) for a in c3_axes()], "q2": 0.5,
                          "flat_index": idx, "ll": [v for v, _, _ in res], "sp": [sp for _, _, sp in res],
                          "cpu_seconds_total": sum(t for _, t, _ in res),
                          "candidates": {"flat_index": cand, "ll": [v for v, _, _ in cres],
                                         "sp": [sp for _, _, sp in cres],
                                         "reference_argmin_flat": arg, "reference_min_negll": best}})


def section_c3argmin(pool):
    """SURVEY.md 8(d) parity procedure for the headline grid: the REFERENCE evaluated at the GPU's top-64
    candidates and at the 2 P axis neighbours of the GPU's arg-min (flat indices written on the GPU box by
    tools/dump_c3_candidates.py; only indices are taken from that file), and the reference's own winner among
    them under the scan of covest/grid.py:65-70 (strict <, first index wins)."""
    hist = synth_hist("H10k_rep", pool)
    src = os.environ.get("COVEST_C3_CANDIDATES", os.path.join(HERE, "c3_candidates_gpu.json"))
    with open(src) as f:
        cand_all = json.load(f)
    out = {}
    for tail in (0, 1000):
        key = "tail%d" % tail
        if key not in cand_all:
            continue
        cand = sorted(int(i) for i in cand_all[key]["candidates"])
        cache = {}
        old_path = os.path.join(HERE, "c3_argmin.json")
        if os.path.exists(old_path):
            with open(old_path) as f:
                old = json.load(f).get(key, {})
            cache = {i: (v, t, sp) for i, v, t, sp in zip(old.get("flat_index", []), old.get("ll", []),
                                                          old.get("cpu_seconds_per_point", []),
                                                          old.get("sp") or [None] * len(old.get("ll", [])))
                     if not tail or sp is not None}
        cache.update(_cached("c3_sample.json" if tail == 0 else "c3_tail_sample.json", tail))
        ll, secs, sp = _c3_eval(pool, hist, tail, cand, cache)
        best, arg = None, -1
        for i, v in zip(cand, ll):
            if v == v and (best is None or -v < best):
                best, arg = -v, i
        out[key] = {"flat_index": cand, "points": [list(c3_point(i)) for i in cand], "ll": ll, "sp": sp,
                    "cpu_seconds_per_point": secs, "reference_argmin_flat": arg, "reference_min_negll": best,
                    "gpu_argmin_flat": int(cand_all[key]["argmin_flat"])}
        print("c3argmin", key, "reference winner", arg, best, "GPU said", cand_all[key]["argmin_flat"], flush=True)
    dump("c3_argmin.json", dict(out, what="config 3 arg-min candidates (GPU top-64 + axis neighbours of its "
                                          "arg-min) evaluated by the reference; RepeatsModel on H10k_rep.hist",
                                hist="H10k_rep", k=21, r=100, max_error=8))


def section_c2argmin(pool):
    """SURVEY.md 8(d) parity procedure for config 2 without a tail (round 5: until then its arg-min was judged by the
    oracle, config 3's by the reference): the REFERENCE evaluated at the GPU's top-64 candidates and at the 2 P axis
    neighbours of the GPU's arg-min (flat indices written on the GPU box by tools/dump_c3_candidates.py into
    c2_candidates_gpu.json; only indices are taken from that file), and the reference's own winner among them under
    the scan of covest/grid.py:65-70 (strict <, first index wins)."""
    hist = synth_hist("H10k_basic", pool)
    src = os.environ.get("COVEST_C2_CANDIDATES", os.path.join(HERE, "c2_candidates_gpu.json"))
    with open(src) as f:
        cand_all = json.load(f)
    cand = sorted(int(i) for i in cand_all["tail0"]["candidates"])
    res = pool.map(_ll_sp_job, [("basic", hist, 0, c2_point(i)) for i in cand], chunksize=2)
    ll = [v for v, _, _ in res]
    best, arg = None, -1
    for i, v in zip(cand, ll):
        if v == v and (best is None or -v < best):
            best, arg = -v, i
    out = {"tail0": {"flat_index": cand, "points": [list(c2_point(i)) for i in cand], "ll": ll, "sp": [s for _, _, s in res],
                     "cpu_seconds_per_point": [t for _, t, _ in res], "reference_argmin_flat": arg,
                     "reference_min_negll": best, "gpu_argmin_flat": int(cand_all["tail0"]["argmin_flat"])}}
    print("c2argmin reference winner", arg, best, "GPU said", cand_all["tail0"]["argmin_flat"], flush=True)
    dump("c2_argmin.json", dict(out, what="config 2 arg-min candidates (GPU top-64 + axis neighbours of its arg-min) "
                                          "evaluated by the reference; BasicModel on H10k_basic.hist, tail 0",
                                hist="H10k_basic", k=21, r=100, max_error=8))


# ----------------------------------------------------------------------------- (7) grid traces
class NegLogLikelihood:
    """Picklable adapter with the contract of CoverageEstimator.likelihood_f
    (covest/covest.py:26-31) for fix=None, err_scale=1: x -> -LL(*x)."""

    def __init__(self, model):
        self.model = model

    def __call__(self, x):
        return -self.model.compute_loglikelihood(*list(x))


def section_gridtrace():
    import covest.grid as G
    out = []
    for model_name in ("basic", "repeats"):
        hist = ref_hist("sim_c10_e0.05")
        cls = BasicModel if model_name == "basic" else RepeatsModel
        m = cls(21, 100, hist, 0, max_error=8)
        fn = NegLogLikelihood(m)
        log = []
        G.verbose_print = log.append
        guess = [10.0, 0.05] if model_name == "basic" else [10.0, 0.05, 0.65, 0.5, 0.5]
        t0 = time.time()
        res = G.optimize_grid(fn, guess, bounds=list(m.bounds), fix=None, n_threads=8)
        out.append({"model": model_name, "hist": "sim_c10_e0.05", "k": 21, "r": 100,
                    "max_error": 8, "tail": 0, "initial_guess": guess,
                    "bounds": [list(b) for b in m.bounds], "result": list(res),
                    "result_negll": fn(res), "log": log, "wall_seconds": time.time() - t0})
    dump("grid_trace.json", {"what": "covest.grid.optimize_grid(-LL, guess, bounds) verbose log",
                             "traces": out})


# ----------------------------------------------------------------------------- (8) the overflow domain
OVF_HISTS = {
    # keys at the top of the 10 000-key range: truncated_poisson(l, j) overflows to +inf once
    # j ln l - ln j! > ln LDBL_MAX = 11356.5 (c_src/covest_poissonmodule.c:19-24), i.e. l > ~11 459 at j = 10 000
    "ovf_top": {9990: 5, 9995: 12, 10000: 7},
    # ... plus a counted key far below: its p_j is 0 there (log -> -inf), so the sum is inf - inf = NaN
    "ovf_nan": {100: 3, 9990: 5, 9995: 12, 10000: 7},
}


def section_overflow():
    """Where the reference's long-double product overflows: its LL is +inf or NaN there and
    optimize_grid selects -(+inf) (covest/grid.py:65-70).  LL values across the band for both models, and the
    reference's own optimize_grid trace walking into it (basic model, free (c, e))."""
    import covest.grid as G
    cases = []
    rnd = random.Random(20241004)
    for hname, hist in OVF_HISTS.items():
        for tail in (0, 50):
            m = BasicModel(21, 100, hist, tail, max_error=8)
            pts = [(c, e) for c in (12000.0, 13000.0, 14000.0, 14200.0, 14300.0, 14350.0, 14400.0, 14500.0, 15000.0,
                                    16000.0, 20000.0, 40000.0) for e in (0.0, 0.001, 0.01)]
            pts += [(rnd.uniform(13500, 15500), rnd.uniform(0, 0.004)) for _ in range(24)]
            cases.append({"model": "basic", "hist": hname, "hist_items": [[j, v] for j, v in hist.items()], "k": 21,
                          "r": 100, "tail": tail, "max_error": 8, "points": [list(p) for p in pts],
                          "ll": [m.compute_loglikelihood(*p) for p in pts]})
        m = RepeatsModel(21, 100, hist, 0, max_error=8)
        pts = [(c, 0.001, q1, 0.5, q) for c in (60.0, 90.0, 100.0, 105.0, 110.0, 150.0) for q1, q in
               ((0.5, 0.1), (0.7, 0.2), (0.9, 0.05))]
        cases.append({"model": "repeats", "hist": hname, "hist_items": [[j, v] for j, v in hist.items()], "k": 21,
                      "r": 100, "tail": 0, "max_error": 8, "points": [list(p) for p in pts],
                      "ll": [m.compute_loglikelihood(*p) for p in pts]})
    traces = []
    for hname, guess in (("ovf_top", [13500.0, 0.001]), ("ovf_top", [11000.0, 0.002]), ("ovf_nan", [13500.0, 0.001])):
        hist = OVF_HISTS[hname]
        m = BasicModel(21, 100, hist, 0, max_error=8)
        fn = NegLogLikelihood(m)
        log = []
        G.verbose_print = log.append
        res = G.optimize_grid(fn, list(guess), bounds=list(m.bounds), fix=None, n_threads=8)
        traces.append({"model": "basic", "hist": hname, "hist_items": [[j, v] for j, v in hist.items()], "k": 21,
                       "r": 100, "tail": 0, "max_error": 8, "initial_guess": guess,
                       "bounds": [list(b) for b in m.bounds], "result": list(res), "result_negll": fn(res), "log": log})
        print("overflow trace", hname, guess, "->", list(res), fn(res), len([l for l in log if "Grid size" in l]),
              "iterations", flush=True)
    for c in cases:
        print("overflow", c["model"], c["hist"], "tail", c["tail"], "finite",
              sum(1 for v in c["ll"] if math.isfinite(v)), "+inf", sum(1 for v in c["ll"] if v == math.inf),
              "-inf", sum(1 for v in c["ll"] if v == -math.inf), "nan", sum(1 for v in c["ll"] if v != v), flush=True)
    dump("overflow.json", {"what": "the reference where its long-double pmf product overflows: compute_loglikelihood "
                                   "values (+inf / NaN included) and covest.grid.optimize_grid traces",
                           "cases": cases, "traces": traces})


def main():
    wanted = sys.argv[1:] or ["tp", "basic", "repeats", "threshold", "hists", "c1", "c2", "c2classes", "c2trim", "c2argmin", "c3", "c3tail",
                              "c3trim", "c3argmin", "gridtrace", "overflow"]
    pool = multiprocessing.Pool(int(os.environ.get("COVEST_GOLDEN_PROCS", "8")))
    for name in wanted:
        t0 = time.time()
        fn = globals()["section_" + name]
        if name in ("hists", "c1", "c2", "c2classes", "c2trim", "c2argmin", "c3", "c3tail", "c3trim", "c3argmin"):
            fn(pool)
        else:
            fn()
        print("section", name, "%.1fs" % (time.time() - t0), flush=True)
    pool.close()


if __name__ == "__main__":
    main()
