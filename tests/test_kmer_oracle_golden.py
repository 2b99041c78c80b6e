"""Pin the k-mer oracle (oracle/kmer_oracle.py) to the golden vectors generated from the
reference's bin/kmer_hist.py.  CPU only; exact integer equality."""
import numpy as np

from conftest import load_golden


def test_histograms_match_reference():
    from oracle import kmer_oracle as ko
    g = load_golden("kmer_hist.json")
    assert len(g["cases"]) >= 26 and max(c["k"] for c in g["cases"]) == 130  # (k > 31: keys beyond 64 bits)
    for c in g["cases"]:
        assert ko.histogram(c["reads"], c["k"], c["nstrategy"]) == c["hist"], c["name"]
        codes, _ = ko.count_kmers(c["reads"], c["k"], c["nstrategy"])
        assert len(codes) == c["distinct"], c["name"]


def test_hash_helpers_match_reference():
    from oracle import kmer_oracle as ko
    g = load_golden("kmer_hist.json")
    for h in g["helpers"]:
        km = h["kmer"]
        k = len(km)
        assert int(ko.kmer_codes(ko.encode(km), k)[0]) == h["hash"]
        # rehash(old, b, k): the window hash after sliding one base (bin/kmer_hist.py:26-31)
        for b, key in (("a", "rehash_a"), ("t", "rehash_t")):
            assert int(ko.kmer_codes(ko.encode(km + b), k)[1]) == h[key]


def test_canonical_is_strand_symmetric():
    from oracle import kmer_oracle as ko
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    rng = np.random.default_rng(3)
    reads = ["".join(rng.choice(list("ACGT"), size=80)) for _ in range(40)]
    rc = ["".join(comp[b] for b in reversed(r)) for r in reads]
    for k in (5, 21, 31, 40, 70):
        a = ko.count_kmers(reads, k, canonical=True)
        b = ko.count_kmers(rc, k, canonical=True)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        assert ko.histogram(reads + rc, k, canonical=True)[1::2] == [0] * len(ko.histogram(reads + rc, k, canonical=True)[1::2])
