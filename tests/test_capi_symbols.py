"""The C-ABI library loads on a CPU-only machine and exports exactly what
include/covest_amd.h declares.  No compute call is made here (no GPU)."""
import os
import re

from conftest import REPO


def _declared():
    text = open(os.path.join(REPO, "include", "covest_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(covest_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree(hip_lib):
    from covest_amd import _capi
    declared = _declared()
    assert declared == sorted(_capi.EXPORTS)
    for name in declared:
        assert hasattr(hip_lib, name), "libcovest_amd.so does not export %s" % name
    assert hip_lib.covest_abi_version() == 1


def test_no_cpu_fallback_in_product():
    """The product package never imports the oracle and has no CPU compute path."""
    pkg = os.path.join(REPO, "covest_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(root, f)).read()
                assert "covest_oracle" not in text and "import oracle" not in text, f


def test_no_value_changing_knob_in_the_shipped_library(hip_lib):
    """The diagnostics that change what the kernels compute -- phases skipped for instruction counts, the sharing or
    the sum items switched off for A/B runs, other assignment constants -- exist only in builds with -DCOVEST_DIAG
    (tools/bin/, loaded through COVEST_AMD_LIB).  The shipped library must not even contain their names."""
    from covest_amd import build
    if os.environ.get("COVEST_AMD_LIB"):
        import pytest
        pytest.skip("COVEST_AMD_LIB points at another build")
    blob = open(build.LIB_PATH, "rb").read()
    for knob in (b"COVEST_FACTORED_SKIP", b"COVEST_FACTORED_DIAG", b"COVEST_FACTORED_SHARE", b"COVEST_FACTORED_NBUF",
                 b"COVEST_NO_SUM_ITEMS", b"COVEST_FACTORED_BUILD_COST", b"COVEST_FACTORED_UNIT_OVERHEAD",
                 b"COVEST_FACTORED_SHARED_DIV", b"COVEST_FACTORED_MIN_SHARED", b"COVEST_KMER_EXP",
                 b"COVEST_FACTORED_LAST_BUILDER_EXTRA", b"COVEST_DIAG_QUEUE", b"COVEST_KMER_M", b"COVEST_KMER_LG", b"COVEST_KMER_SAMPLE"):
        assert knob not in blob, knob.decode()
    # what may stay: pure host-side I/O settings of the read parser (thread count, page-locked buffers)
    allowed = set(re.findall(rb"COVEST_[A-Z_]{4,}", blob)) - {b"COVEST_READER_THREADS", b"COVEST_READER_PINNED"}
    assert not {a for a in allowed if not a.startswith((b"COVEST_E_", b"COVEST_OK", b"COVEST_MODEL", b"COVEST_KERNEL",
                                                        b"COVEST_MAX", b"COVEST_ABI"))}, allowed


def test_compute_fails_loudly_without_device(hip_lib):
    """On a machine without a GPU every compute entry point raises -- it never
    silently computes on the CPU."""
    import pytest
    from covest_amd import BasicModel, _capi
    if hip_lib.covest_device_count() > 0:
        pytest.skip("a HIP device is present")
    m = BasicModel(21, 100, {1: 10, 2: 5}, 0, max_error=8)
    with pytest.raises(_capi.CovestHipError):
        m.compute_loglikelihood(10.0, 0.05)


def test_model_surface_without_device():
    """Attribute surface of covest/models.py that main() and print_output touch."""
    import pickle
    from covest_amd import BasicModel, RepeatsModel, models, select_model
    hist = {1: 10, 2: 5, 3: 1}
    b = BasicModel(21, 100, hist, 3, max_error=8, max_cov=50)
    assert b.params == ('coverage', 'error_rate') and b.param_count == 2
    assert b.bounds == ((0.01, 50), (0, 0.5)) and b.defaults == (1, 0.25)
    assert b.max_error == 8 and b.repeats is False and b.short_name() == 'basic'
    assert b.correct_c(10.0) == 10.0 * 80 / 100
    assert b.fit_to_bounds([100.0, -1.0]) == [50, 0]
    assert len(b.comb) == 22 and b.comb[0] == 1.0 and b.comb[1] == 63.0
    r = RepeatsModel(21, 100, hist, 3, max_error=8, max_cov=50)
    assert r.params == ('coverage', 'error_rate', 'q1', 'q2', 'q') and r.param_count == 5
    assert r.bounds == ((0.01, None), (0, 0.5), (0.3, 1), (0, 1), (0, 1))  # max_cov is dropped, models.py:177
    assert r.defaults == (1, 0.25, 0.65, 0.5, 0.5) and r.repeats is True and r.threshold == 1e-8
    assert BasicModel(5, 100, hist, 0).max_error == 6
    assert set(models) == {'basic', 'repeats'}
    assert select_model('basic') is BasicModel and select_model('repeat') is RepeatsModel
    assert select_model('r') is RepeatsModel
    import pytest
    with pytest.raises(ValueError):
        select_model('nope')
    # picklable (covest/grid.py:48 pickles the bound likelihood function), handle never travels
    r2 = pickle.loads(pickle.dumps(r))
    assert r2.hist == hist and r2._handle is None and r2.bounds == r.bounds


def test_import_order_guard():
    """One process, one HIP runtime: torch must be imported BEFORE libcovest_amd.so is loaded.  The
    other order is refused loudly wherever device memory would cross between the two (covest_amd._capi
    .require_shared_runtime) instead of handing torch a pointer of another runtime."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from covest_amd import _capi\n"
            "%s"
            "_capi.lib()\n"
            "%s"
            "try:\n"
            "    _capi.require_shared_runtime('test')\n"
            "    print('ok')\n"
            "except _capi.CovestHipError as e:\n"
            "    print('refused', 'two HIP runtimes' in str(e))\n")
    late = subprocess.run([sys.executable, "-c", code % (repo, "", "import torch\n")], capture_output=True, text=True)
    assert late.stdout.strip() == "refused True", (late.stdout, late.stderr[-500:])
    early = subprocess.run([sys.executable, "-c", code % (repo, "import torch\n", "")], capture_output=True, text=True)
    assert early.stdout.strip() == "ok", (early.stdout, early.stderr[-500:])
    none = subprocess.run([sys.executable, "-c", code % (repo, "", "")], capture_output=True, text=True)
    assert none.stdout.strip() == "ok", (none.stdout, none.stderr[-500:])
