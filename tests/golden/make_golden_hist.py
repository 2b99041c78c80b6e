#!/usr/bin/env python3
"""Golden vectors for the steps around the likelihood path, made by running
the REFERENCE: covest/histogram.py (compute_coverage_apx, sample_histogram, get_trim, trim_hist,
process_histogram) and covest/data.py (load_histogram / save_histogram round trip, print_output).

Build container only (/root/reference).  Writes DATA ONLY: tests/golden/hist_steps.json.

sample_histogram ends with an unseeded randomised rounding (covest/histogram.py:71-74).  Two views are
recorded: the real-valued expected counts (the module's `ceil`/`floor` replaced, for this session only,
by the identity, so the function returns the values it would round) and the rounded counts under a
recorded sequence of uniforms fed to `random.random`.
"""
import io
import json
import os
import subprocess
import sys
import contextlib

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REFERENCE = os.environ.get("COVEST_REFERENCE", "/root/reference")
REF_BUILD = os.path.join(REPO, "oracle", "_ref")

subprocess.check_call(["make", "-C", os.path.join(REPO, "oracle"), "ref"], stdout=subprocess.DEVNULL)
os.environ.setdefault("MPLBACKEND", "Agg")
import scipy.misc  # noqa: E402
import scipy.special  # noqa: E402
if not hasattr(scipy.misc, "comb"):
    scipy.misc.comb = scipy.special.comb
import types  # noqa: E402
if "Bio" not in sys.modules:  # covest/data.py imports Bio.SeqIO for read files; not used here
    bio = types.ModuleType("Bio")
    bio.SeqIO = types.ModuleType("Bio.SeqIO")
    sys.modules["Bio"] = bio
    sys.modules["Bio.SeqIO"] = bio.SeqIO
sys.path.insert(0, REF_BUILD)
sys.path.insert(0, REFERENCE)
import covest.constants  # noqa: E402
covest.constants.VERBOSE = False
import covest.histogram as H  # noqa: E402
import covest.data as D  # noqa: E402
from covest.models import BasicModel, RepeatsModel  # noqa: E402
from covest_poisson import poisson_dist  # noqa: E402


def load(name):
    hist = {}
    with open(os.path.join(HERE, name + ".hist")) as f:
        for line in f:
            if line.strip() and line[0] != "#":
                a, b = line.split()[:2]
                hist[int(a)] = int(b)
    return hist


def items(d):
    return [[int(k), (float(v) if isinstance(v, float) else int(v))] for k, v in d.items()]


out = {"_made_by": "tests/golden/make_golden_hist.py", "cases": {}}

# ---- poisson_dist of the C extension, incl. the l > 200 regime (c_src/covest_poissonmodule.c:64-108) ----
out["poisson_dist"] = [{"l": l, "max_j": mj, "p": list(poisson_dist(l, mj))}
                       for l, mj in [(50.0, 100), (60.5, 121), (199.5, 399), (200.0, 400), (250.0, 500),
                                     (433.3, 650), (1250.0, 1500), (0.0, 5)]]

# ---- scipy's binomial pmf as sample_histogram calls it (covest/histogram.py:60-62) ----
from scipy.stats import binom  # noqa: E402
out["binom_pmf"] = []
for i, factor in [(1, 2), (7, 2), (30, 3), (99, 2), (99, 7), (64, 13)]:
    b = binom(i, 1.0 / factor)
    out["binom_pmf"].append({"i": i, "factor": factor, "p": [float(b.pmf(j)) for j in range(1, i + 1)]})

# ---- the histogram steps on fixtures already in tests/golden ----
hists = {"sim_c10_e0.05": load("sim_c10_e0.05"), "sim_c10_e0": load("sim_c10_e0"), "H256": load("H256")}
# a wide one: keys to 640 (poisson branch beyond 100, l up to 320 at factor 2), built from H10k_rep
wide = {k: v for k, v in load("H10k_rep").items() if k <= 640}
hists["H10k_rep_le640"] = wide

ident = lambda v: v  # noqa: E731
for name, hist in hists.items():
    case = {"k": 21, "r": 100}
    case["coverage_apx"] = list(H.compute_coverage_apx(hist, 21, 100))
    case["get_trim"] = int(H.get_trim(hist)) if len(hist) > 0 else None
    case["get_trim_ignore_last"] = int(H.get_trim(hist, ignore_last=True))
    thr = max(3, max(hist) // 2)
    th, tail = H.trim_hist(hist, thr)
    case["trim_hist"] = {"threshold": thr, "hist": items(th), "tail": int(tail)}
    case["sample"] = []
    for factor, trim in [(2, None), (3, None), (5, 40)]:
        saved = (H.ceil, H.floor)
        H.ceil = H.floor = ident
        try:
            real = H.sample_histogram(hist, factor=factor, trim=trim)
        finally:
            H.ceil, H.floor = saved
        # rounded under a recorded uniform sequence
        import random as _r
        gen = _r.Random(20240521 + factor)
        us = []

        def fake():
            u = gen.random()
            us.append(u)
            return u
        saved_rand = H.random.random
        H.random.random = fake
        try:
            rounded = H.sample_histogram(hist, factor=factor, trim=trim)
        finally:
            H.random.random = saved_rand
        case["sample"].append({"factor": factor, "trim": trim, "expected": items(real), "uniforms": us,
                               "rounded": items(rounded)})
    # process_histogram with fixed sample factor 1 (no sampling, trimming rules only)
    for trim in (None, 0, 12):
        ph, ptail, sf, c, e = H.process_histogram(hist, 21, 100, trim=trim, sample_factor=1)
        case.setdefault("process_sf1", []).append({"trim": trim, "hist": items(ph), "tail": int(ptail),
                                                   "sample_factor": sf, "c": c, "e": e})
    out["cases"][name] = case

# ---- print_output (covest/data.py:106-173): the YAML fields of one finished estimate ----
hist = hists["sim_c10_e0.05"]
for kind, cls, est in (("basic", BasicModel, (10.0, 0.05)), ("repeats", RepeatsModel, (10.0, 0.05, 0.8, 0.5, 0.3))):
    m = cls(21, 100, hist, 0, max_error=8)
    with contextlib.redirect_stdout(io.StringIO()):
        guess = list(m.defaults)  # covest/covest.py:152-155: the defaults with the guessed c and e
        guess[:2] = 9.0, 0.04
        data = D.print_output(hist, m, True, 2, estimated=list(est), guess=guess, orig=[None] * len(est),
                              reads_size=123456789, silent=True, orig_sample_factor=3, starting_points=4,
                              use_grid_search=True)
    out.setdefault("print_output", {})[kind] = {"guess": [float(g) for g in guess], "estimated": list(est),
                                                "fields": {k: v for k, v in data.items() if k != "version"}}

# ---- load/save round trip format (covest/data.py:22-41,176-182) ----
tmp = os.path.join("/tmp", "covest_golden_rt.hist")
D.save_histogram({3: 7, 1: 2, 10: 1}, tmp, {"tool": "x 1.0", "sample_factor": 6})
with open(tmp) as f:
    out["save_histogram_text"] = f.read()
h2, meta = D.load_histogram(tmp)
out["load_histogram"] = {"hist": items(h2), "meta": meta}

# ---- the whole default flow of covest/covest.py:main on the reference's own test histogram ----
import argparse  # noqa: E402
import yaml  # noqa: E402
import covest.covest as C  # noqa: E402
out["end_to_end"] = {}
for model_name in ("basic", "repeats"):
    args = argparse.Namespace(
        load=None, input_histogram=os.path.join(HERE, "sim_c10_e0.05.hist"), kmer_size=21, read_length=100, trim=None,
        sample_factor=None, error_scale=1, coverage=None, model=model_name, max_coverage=None, min_q1=0.3,
        error_rate=None, params=tuple(), fix=False, ll_only=False, start_original=False, starting_points=1, grid=False,
        thread_count=1, reads_size=None, plot=None)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf), contextlib.redirect_stderr(io.StringIO()):
        C.main(args)
    rec = yaml.safe_load(buf.getvalue())
    rec.pop("version", None)
    out["end_to_end"][model_name] = rec

with open(os.path.join(HERE, "hist_steps.json"), "w") as f:
    json.dump(out, f)
print("wrote hist_steps.json", os.path.getsize(os.path.join(HERE, "hist_steps.json")), "bytes")
