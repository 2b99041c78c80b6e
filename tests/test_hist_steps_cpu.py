"""Host-side histogram steps of covest_amd.hist_steps (no GPU work): file format, first guess, trimming,
process_histogram without sampling -- against the reference's own outputs (tests/golden/hist_steps.json)."""
import pytest

from conftest import load_golden, load_hist, rel_err

from covest_amd import hist_steps as hs

G = load_golden("hist_steps.json")


def _hist(name):
    if name == "H10k_rep_le640":
        return {k: v for k, v in load_hist("H10k_rep").items() if k <= 640}
    return load_hist(name)


def test_save_and_load_round_trip(tmp_path):
    path = str(tmp_path / "rt.hist")
    hs.save_histogram({3: 7, 1: 2, 10: 1}, path, {"tool": "x 1.0", "sample_factor": 6})
    with open(path) as f:
        assert f.read() == G["save_histogram_text"]
    hist, meta = hs.load_histogram(path)
    assert [[k, v] for k, v in hist.items()] == G["load_histogram"]["hist"]
    assert meta == G["load_histogram"]["meta"]


def test_load_rejects_garbage(tmp_path):
    path = str(tmp_path / "bad.hist")
    with open(path, "w") as f:
        f.write("1 2\nx y\n")
    with pytest.raises(hs.InvalidFormatException):
        hs.load_histogram(path)


@pytest.mark.parametrize("name", sorted(G["cases"]))
def test_guess_and_trimming(name):
    case = G["cases"][name]
    hist = _hist(name)
    c, e = hs.compute_coverage_apx(hist, case["k"], case["r"])
    assert c == case["coverage_apx"][0] and e == case["coverage_apx"][1]  # same arithmetic: bit-identical
    assert hs.get_trim(hist) == case["get_trim"]
    assert hs.get_trim(hist, ignore_last=True) == case["get_trim_ignore_last"]
    kept, tail = hs.trim_hist(hist, case["trim_hist"]["threshold"])
    assert [[k, v] for k, v in kept.items()] == case["trim_hist"]["hist"] and tail == case["trim_hist"]["tail"]
    for want in case["process_sf1"]:
        ph, ptail, sf, pc, pe = hs.process_histogram(hist, case["k"], case["r"], trim=want["trim"], sample_factor=1)
        assert [[k, v] for k, v in ph.items()] == want["hist"]
        assert (ptail, sf, pc, pe) == (want["tail"], want["sample_factor"], want["c"], want["e"])


def test_degenerate_histograms():
    assert hs.compute_coverage_apx({}, 21, 100) == (0.0, 1.0)
    assert hs.compute_coverage_apx({1: 10}, 21, 100) == (0.0, 1.0)  # division by zero -> the reference's fallback
    assert hs.trim_hist({1: 5, 2: 3}, 10) == ({1: 5, 2: 3}, 0)
