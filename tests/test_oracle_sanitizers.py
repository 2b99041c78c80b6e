"""The oracle's C restatement under AddressSanitizer + UndefinedBehaviorSanitizer.

Sanitizers run on the CPU build only (GPU ASan is not available on this pool).  The sanitizer build of
oracle/covest_oracle.c (`make -C oracle libcovest_oracle_asan.so`) is loaded into a child interpreter that has
the ASan runtime preloaded, and walks a subset of every golden fixture through every entry point the parity
tests use -- the pmf, threshold_o, p_j, the likelihood (threaded and not), the log-domain mode, first_min --
plus the histogram-thinning twin.  Any report (heap overflow, use after free, signed overflow, misaligned or
out-of-bounds index, ...) fails the test.  The reference itself has real UB at c_src/covest_poissonmodule.c:16
(an int passed for a double vararg); the restatement must not."""
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import json, math, os, sys
sys.path.insert(0, os.environ["COVEST_REPO"])
sys.path.insert(0, os.path.join(os.environ["COVEST_REPO"], "tests"))
import numpy as np
from conftest import load_golden, load_hist, rel_err
from oracle import covest_oracle as orc
assert orc._LIB_PATH.endswith("_asan.so")
n = 0
for l, j, want in load_golden("tp_table.json")["rows"][::3]:
    got = orc.truncated_poisson(l, j)
    assert (got == want) or rel_err(got, want) <= 1e-15, (l, j, got, want)
    n += 1
for hist_max, q1, q2, q, want in load_golden("threshold_o.json")["rows"][::7]:
    assert orc.threshold_o(q1, q2, q, 1e-8, int(hist_max)) == want
    n += 1
for kind, fname in (("basic", "basic_ll.json"), ("repeats", "repeats_ll.json")):
    for case in load_golden(fname)["cases"][::2]:
        kw = dict(max_error=case["max_error"])
        if kind == "basic":
            kw["max_cov"] = case.get("max_cov")
        else:
            kw["threshold"] = case.get("threshold", 1e-8)
            kw["min_single_copy_ratio"] = case.get("min_single_copy_ratio", 0.3)
        om = orc.OracleModel(kind, case["k"], case["r"], load_hist(case["hist"]), case["tail"], **kw)
        pts = np.array(case["points"][:12])
        for threads in (1, 3):
            got = om.compute_loglikelihood_many(pts, n_threads=threads)
            for a, b in zip(got, case["ll"]):
                assert rel_err(float(a), float(b)) <= 1e-13, (kind, case["hist"], a, b)
        om.compute_loglikelihood_many_fast(pts, n_threads=2)
        for d in case["detail"][:2]:
            probs = om.compute_probabilities(*d["point"])
            for jj, v in d["p_j"]:
                assert rel_err(probs[jj], v) <= 1e-13
        n += len(pts)
# ragged / empty inputs
om = orc.OracleModel("basic", 21, 100, {}, 0, max_error=8)
assert om.compute_loglikelihood(10.0, 0.05) == 0.0
om = orc.OracleModel("repeats", 21, 100, {40: 3, 2: 5_000_000_000, 7: 0, 1000: 1, 1: 12}, 7, max_error=22)
om.compute_loglikelihood_many(np.array([(10.0, 0.05, 0.7, 0.5, 0.5), (float("nan"), 0.1, 0.3, 0.0, 0.9)]), n_threads=2)
assert orc.first_min(np.array([math.nan, 3.0, 2.0, 2.0, math.inf])) == (2, 2.0)
assert orc.first_min(np.array([])) [0] == -1
print("sanitized oracle calls ok:", n)
'''


def _runtime(name):
    path = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    return path if os.path.isabs(path) and os.path.exists(path) else None


def test_oracle_under_asan_and_ubsan():
    asan = _runtime("libasan.so")
    if asan is None:
        pytest.skip("gcc ships no libasan here")
    subprocess.check_call(["make", "-C", os.path.join(REPO, "oracle"), "libcovest_oracle_asan.so"],
                          stdout=subprocess.DEVNULL)
    lib = os.path.join(REPO, "oracle", "libcovest_oracle_asan.so")
    env = dict(os.environ, COVEST_REPO=REPO, COVEST_ORACLE_LIB=lib, LD_PRELOAD=asan,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=97",  # (CPython itself 'leaks' at exit)
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=0")
    proc = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=900)
    report = proc.stdout + proc.stderr
    assert "AddressSanitizer" not in report and "runtime error:" not in report, report[-4000:]
    assert proc.returncode == 0, report[-4000:]
    assert "sanitized oracle calls ok" in proc.stdout
