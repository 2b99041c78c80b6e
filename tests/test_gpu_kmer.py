"""Parity of the HIP k-mer histogram (through the C ABI) with the reference's
bin/kmer_hist.py (golden vectors) and with the numpy oracle.  Exact integer equality."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


def test_golden_histograms(hip_lib):
    from covest_amd import kmer_hist as kh
    g = load_golden("kmer_hist.json")
    for c in g["cases"]:
        # the reference's call pattern: one compute_counts per read into the same counts
        counts = None
        for r in c["reads"]:
            counts = kh.compute_counts(kh.preprocess(r, c["nstrategy"]), prev_counts=counts, k=c["k"])
        assert kh.compute_histogram(counts) == c["hist"], c["name"]
        assert len(counts) == c["distinct"], c["name"]
        # and batched: all reads in one launch, from a deliberately tiny table that has to grow
        batched = kh.KmerCounts(c["k"], min_slots=1024)
        batched.add_reads([kh.preprocess(r, c["nstrategy"]) for r in c["reads"]])
        assert batched.histogram() == c["hist"], c["name"]
        counts.close()
        batched.close()
    for h in g["helpers"]:
        assert kh.hash_kmer(h["kmer"]) == h["hash"]
        assert kh.rehash(h["hash"], "a", len(h["kmer"])) == h["rehash_a"]
        assert kh.rehash(h["hash"], "t", len(h["kmer"])) == h["rehash_t"]


@pytest.mark.parametrize("canonical", [False, True])
def test_random_reads_against_oracle(hip_lib, canonical):
    from covest_amd import kmer_hist as kh
    from oracle import kmer_oracle as ko
    rng = np.random.default_rng(11)
    genome = "".join(rng.choice(list("ACGT"), size=200_000))
    reads = []
    for _ in range(20_000):
        s = int(rng.integers(0, len(genome) - 100))
        r = list(genome[s:s + 100])
        for i in np.flatnonzero(rng.random(100) < 0.01):  # 1 % substitutions
            r[i] = "ACGT"[int(rng.integers(0, 4))]
        reads.append("".join(r))
    for k in (21, 31):
        counts = kh.KmerCounts(k, canonical=canonical)
        for i in range(0, len(reads), 3000):  # several launches into one table
            counts.add_reads(reads[i:i + 3000])
        want = ko.histogram(reads, k, canonical=canonical)
        assert counts.histogram() == want
        assert len(counts) == len(ko.count_kmers(reads, k, canonical=canonical)[0])
        counts.close()
    # k > 31: keys of 2, 4 and 8 words (kmer_wide.hip) -- the reference's integers have no limit (bin/kmer_hist.py:
    # 18-31); a small table that has to grow several times, several launches
    few = reads[:3000]
    for k in (32, 45, 63, 64, 90, 100):
        counts = kh.KmerCounts(k, canonical=canonical, min_slots=1024)
        for i in range(0, len(few), 700):
            counts.add_reads(few[i:i + 700])
        assert counts.histogram() == ko.histogram(few, k, canonical=canonical), k
        assert len(counts) == len(ko.count_kmers(few, k, canonical=canonical)[0]), k
        counts.close()
    with pytest.raises(Exception):
        kh.KmerCounts(256)


def test_edge_cases(hip_lib, tmp_path):
    from covest_amd import kmer_hist as kh
    from oracle import kmer_oracle as ko
    c = kh.KmerCounts(5)
    assert c.histogram() == [0] and len(c) == 0          # nothing counted yet
    c.add_reads([])
    c.add_reads(["", "ac", "ACGTA", "acgta"])            # empty, short, exactly k (twice, case-insensitive)
    assert c.histogram() == ko.histogram(["", "ac", "ACGTA", "acgta"], 5)
    with pytest.raises(KeyError):
        c.add_reads(["acgtx"])                           # single_hash raises KeyError on other letters
    c.close()
    # N strategies and the file front-end
    fa = tmp_path / "reads.fa"
    fa.write_text(">r1\nACGTNACGTACG\nTTTGACA\n>r2\nNNACGTACGTAC\n")
    assert kh.main(str(fa), None, 4, kh.NS_IGNORE) == ko.histogram(["ACGTNACGTACGTTTGACA", "NNACGTACGTAC"], 4, 0)
    out = tmp_path / "out.hist"
    hist = kh.main(str(fa), str(out), 4, kh.NS_SINGLE)
    assert hist == ko.histogram(["ACGTNACGTACGTTTGACA", "NNACGTACGTAC"], 4, 1)
    assert out.read_text().splitlines()[:2] == ["0 0", "1 %d" % hist[1]]
    fq = tmp_path / "reads.fq"
    fq.write_text("@a\nACGTACGT\n+\nIIIIIIII\n@b\nTTTTACGT\n+\nIIIIIIII\n")
    assert kh.main(str(fq), None, 4, kh.NS_IGNORE) == ko.histogram(["ACGTACGT", "TTTTACGT"], 4)
    # a k-mer seen more often than the LDS-binned range of the histogram kernel
    c = kh.KmerCounts(3)
    c.add_reads(["A" * 6000, "ACG"])
    h = c.histogram()
    assert len(h) == 5999 and h[5998] == 1 and h[1] == 1 and sum(h) == 2
    c.close()


def test_conservation_at_scale(hip_lib):
    """Size-independent properties at a size the oracle does not reach (2.5e6 reads, 2e8 windows):
    every window lands in exactly one bin (sum_i i*h_i = #windows), the canonical table of reads plus
    their reverse complements has only even counts over the same keys, and re-counting into a cleared
    table reproduces the histogram.  (numpy + the host-buffer API: PyTorch bundles its own HIP runtime,
    so this process, which loaded libcovest_amd.so first, must not initialise a second one.)"""
    import ctypes
    from covest_amd import _capi, kmer_hist as kh
    rng = np.random.default_rng(5)
    n_reads, L, k = 2_500_000, 100, 21
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    genome = rng.integers(0, 4, size=3_000_000, dtype=np.uint8)
    starts = rng.integers(0, genome.size - L, size=n_reads)
    codes = genome[starts[:, None] + np.arange(L)[None, :]]
    offsets = (np.arange(n_reads + 1, dtype=np.int64) * L)

    def add(counter, arr):
        blob = np.ascontiguousarray(lut[arr]).reshape(-1)
        counter._reserve_for(n_reads * (L - k + 1))
        _capi.check(_capi.lib().covest_kmer_add(
            counter._handle, blob.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)),
            offsets.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), n_reads), "covest_kmer_add")

    c = kh.KmerCounts(k, canonical=True)
    add(c, codes)
    h = c.histogram()
    assert sum(i * v for i, v in enumerate(h)) == n_reads * (L - k + 1)
    assert sum(h) == len(c)
    add(c, 3 - codes[:, ::-1])  # the reverse complements
    h2 = c.histogram()
    assert sum(i * v for i, v in enumerate(h2)) == 2 * n_reads * (L - k + 1)
    assert all(v == 0 for v in h2[1::2]) and sum(h2) == sum(h)  # strand symmetry: same keys, doubled counts
    c.clear()
    add(c, codes)
    assert c.histogram() == h
    c.close()


def test_one_gigabase_from_a_fasta_file(hip_lib, tmp_path):
    """BASELINE config 5's shape at 1 Gbp, through the file front-end (C++ reader -> covest_kmer_add): 10^7 reads of
    100 bases of a random 25 Mbp genome in a FASTA file (1.1 GB), no substitutions.  Properties that do not need
    an oracle of that size: every window lands in exactly one bin; the distinct 21-mers are (almost all of) the
    genome's; counting the file a second time into the same table doubles every count; and the canonical table of
    the same file has no more keys than the forward one."""
    from covest_amd import kmer_hist as kh
    rng = np.random.default_rng(20240601)
    n_reads, L, k, g_len = 10_000_000, 100, 21, 25_000_000
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    genome = lut[rng.integers(0, 4, size=g_len, dtype=np.uint8)]
    fa = tmp_path / "reads_1gbp.fa"
    chunk = 500_000
    hdr = np.frombuffer(b">r\n", dtype=np.uint8)
    with open(fa, "wb") as f:
        for a in range(0, n_reads, chunk):
            starts = rng.integers(0, g_len - L, size=chunk)
            rec = np.empty((chunk, 3 + L + 1), dtype=np.uint8)
            rec[:, :3] = hdr
            rec[:, 3:3 + L] = genome[starts[:, None] + np.arange(L)[None, :]]
            rec[:, -1] = 10
            f.write(rec.tobytes())
    windows = n_reads * (L - k + 1)
    counts = kh.KmerCounts(k, min_slots=4 * g_len)
    n_seen = 0
    for bases, offs, n, n_bases in kh.ReadBatches(str(fa), kh.NS_IGNORE, batch_bases=1 << 28):
        counts.add_packed(bases, offs, n, n_bases)
        n_seen += n
    assert n_seen == n_reads
    h1 = counts.histogram()
    distinct = len(counts)
    assert sum(i * v for i, v in enumerate(h1)) == windows
    assert sum(h1) == distinct
    # 40x coverage: all but a sliver of the genome's 21-mers are seen; a random 25 Mbp genome repeats almost none
    assert 0.98 * (g_len - k + 1) < distinct <= g_len - k + 1
    peak = int(np.argmax(h1[5:]) + 5)
    assert abs(peak - n_reads * (L - k + 1) / g_len) < 4  # Poisson with mean 32
    for bases, offs, n, n_bases in kh.ReadBatches(str(fa), kh.NS_IGNORE, batch_bases=1 << 28):
        counts.add_packed(bases, offs, n, n_bases)
    h2 = counts.histogram()
    assert len(counts) == distinct
    assert h2[1::2] == [0] * len(h2[1::2]) and h2[0::2][:len(h1)] == h1 and sum(h2) == sum(h1)
    counts.close()
    canon = kh.KmerCounts(k, canonical=True, min_slots=4 * g_len)
    for bases, offs, n, n_bases in kh.ReadBatches(str(fa), kh.NS_IGNORE, batch_bases=1 << 28):
        canon.add_packed(bases, offs, n, n_bases)
    hc = canon.histogram()
    assert sum(i * v for i, v in enumerate(hc)) == windows
    assert len(canon) <= distinct
    canon.close()


_FIXED_LEN_SCRIPT = r"""
import os, sys
import torch                      # first: ONE HIP runtime per process (INTEGRATION.md 8)
sys.path.insert(0, os.environ["COVEST_REPO"])
import numpy as np
from covest_amd import kmer_hist as kh
from oracle import kmer_oracle as ko
dev = torch.device("cuda", 0)
rng = np.random.default_rng(3)
for k, L, canonical in ((21, 100, True), (21, 100, False), (31, 33, True), (5, 5, False), (12, 150, True)):
    n = 3000
    genome = rng.integers(0, 4, size=20000, dtype=np.uint8)
    starts = rng.integers(0, genome.size - L, size=n)
    codes = genome[starts[:, None] + np.arange(L)[None, :]]
    ascii_reads = np.frombuffer(b"ACGT", dtype=np.uint8)[codes]
    reads = ["".join(map(chr, row)) for row in ascii_reads]
    d_bases = torch.from_numpy(np.ascontiguousarray(ascii_reads).reshape(-1)).to(dev)
    d_offs = torch.arange(n + 1, dtype=torch.int64, device=dev) * L
    fixed = kh.KmerCounts(k, canonical=canonical)
    fixed.add_device(d_bases.data_ptr(), n, L)                     # reads of one length: windows numbered through
    ragged = kh.KmerCounts(k, canonical=canonical)
    ragged.add_device(d_bases.data_ptr(), n, L, d_offsets_ptr=d_offs.data_ptr())  # the same reads through offsets
    torch.cuda.synchronize()
    want = ko.histogram(reads, k, canonical=canonical)
    assert fixed.histogram() == want, (k, L, canonical)
    assert ragged.histogram() == want, (k, L, canonical)
    fixed.close(); ragged.close()
print("fixed-length ok")
"""


def test_reads_of_one_length_resident_in_hbm(hip_lib):
    """covest_kmer_add_device without offsets (every read read_len long -- the layout config 5's synthetic reads
    have in HBM) numbers the windows of all reads through and gives a wave 64 consecutive ones; with offsets a wave
    takes a read.  Both against the numpy oracle: several k, read lengths down to exactly k, both strand modes."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, COVEST_REPO=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    proc = subprocess.run([sys.executable, "-c", _FIXED_LEN_SCRIPT], env=env, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0 and "fixed-length ok" in proc.stdout, proc.stdout[-2000:] + proc.stderr[-4000:]


_PARTITIONED_SCRIPT = r"""
import os, sys
import torch                      # first: ONE HIP runtime per process (INTEGRATION.md 8)
sys.path.insert(0, os.environ["COVEST_REPO"])
import numpy as np
from covest_amd import kmer_hist as kh
from oracle import kmer_oracle as ko
dev = torch.device("cuda", 0)
rng = np.random.default_rng(5)
LUT = np.frombuffer(b"ACGT", dtype=np.uint8)
taken = []
def run(name, reads, k, canonical, fixed_len=None, expect=None):
    blob = np.frombuffer("".join(reads).encode(), dtype=np.uint8)
    d_bases = torch.from_numpy(blob.copy()).to(dev) if blob.size else torch.zeros(1, dtype=torch.uint8, device=dev)
    lens = np.array([len(r) for r in reads], dtype=np.int64)
    offs = np.zeros(len(reads) + 1, dtype=np.int64); np.cumsum(lens, out=offs[1:])
    d_offs = torch.from_numpy(offs).to(dev)
    want = ko.histogram(reads, k, canonical=canonical)
    want_distinct = len(ko.count_kmers(reads, k, canonical=canonical)[0])
    for layout in (("fixed",) if fixed_len else ()) + ("offsets",):
        c = kh.KmerCounts(k, canonical=canonical, min_slots=1 << 12)
        if layout == "fixed":
            path = c.count_reads_device(d_bases.data_ptr(), len(reads), fixed_len)
        else:
            path = c.count_reads_device(d_bases.data_ptr(), len(reads), 0, d_offsets_ptr=d_offs.data_ptr(), n_bases=int(offs[-1]))
        got = c.histogram()
        assert got == want, (name, layout, path, k, canonical, got[:8], want[:8])
        assert len(c) == want_distinct, (name, layout, path)
        if expect:
            assert path == expect, (name, layout, path, getattr(c, "why_not_partitioned", ""))
        taken.append((name, layout, path))
        info = c.partition_info() if path == "partitioned" else None
        # the counter is reusable: a second count of the same reads gives the same histogram, not twice the counts
        if layout == "offsets":
            c.count_reads_device(d_bases.data_ptr(), len(reads), 0, d_offsets_ptr=d_offs.data_ptr(), n_bases=int(offs[-1]))
            assert c.histogram() == want, (name, "second count")
        if path == "partitioned":
            try:
                c.add_device(d_bases.data_ptr(), 1, 0, d_offsets_ptr=d_offs.data_ptr())
                raise SystemExit("add after a partitioned count must be refused")
            except kh._capi.CovestHipError:
                pass
            c.clear()
            c.add_reads(reads[:50])      # ... and after clear() the counter is an ordinary one again
            assert c.histogram() == ko.histogram(reads[:50], k, canonical=canonical), (name, "after clear")
        c.close()
    return info

def genome_reads(n, L, g_len, err=0.01):
    genome = rng.integers(0, 4, size=g_len, dtype=np.uint8)
    starts = rng.integers(0, g_len - L, size=n)
    codes = genome[starts[:, None] + np.arange(L)[None, :]]
    flips = rng.random(codes.shape) < err
    codes = np.where(flips, rng.integers(0, 4, size=codes.shape, dtype=np.uint8), codes)
    return ["".join(map(chr, row)) for row in LUT[codes]]

for k, canonical in ((21, True), (21, False), (19, True), (25, False), (31, True)):
    L = 100
    reads = genome_reads(20000, L, 40000)                       # 40x-50x coverage with 1 % substitutions
    run("genome k=%d" % k, reads, k, canonical, fixed_len=L, expect="partitioned")
# reads of different lengths, some shorter than k (they count the hash of what there is), an empty one
ragged = [r[:int(n)] for r, n in zip(genome_reads(4000, 150, 30000), rng.integers(0, 151, size=4000))] + ["", "ACGT"]
run("ragged", ragged, 21, True)
run("ragged forward", ragged, 23, False)
# low complexity: a handful of minimizers hold everything -- their buckets go to a workgroup each; one key (poly-A)
# is seen more than 2^20 times: beyond the dense bins
low = ["A" * 100] * 14000 + ["AC" * 50, "ACG" * 33 + "A", "T" * 100] * 3000 + genome_reads(2000, 100, 5000)
info = run("low complexity", low, 21, True, fixed_len=100)
assert info["buckets_by_workgroup"] > 0, info
# highly repetitive: 12 000 copies of 25 reads -- counts beyond the LDS bins
rep = genome_reads(25, 100, 2000, err=0.0) * 12000
run("repeats", rep, 21, True, fixed_len=100)
# enough reads for pass 0 to SAMPLE them (one block of 1920 bytes / one read in 2 .. 16)
many = genome_reads(100000, 100, 200000)
info = run("sampled", many, 21, True, fixed_len=100, expect="partitioned")
S = info["sampled_1_in"]
assert S > 1, info
# ... and a sample that misleads: outside the sampled blocks (one in S) and reads, one read in 20 is a copy of the same
# one -- its buckets get far more records than they were given room for, overflow, and are counted through the table
misled = list(many)
block = 8 * (256 - 16)
for r in range(len(misled)):
    if (r * 100 // block) % S != 0 and ((r + 1) * 100 // block) % S != 0 and r % S != 0 and r % 20 == 1:
        misled[r] = many[7]
info = run("misled sample", misled, 21, True, fixed_len=100, expect="partitioned")
assert info["overflowed_records"] > 0 and info["buckets_through_table"] > 0, info
# one minimizer, thousands of distinct k-mers around it: more than a workgroup's LDS table holds -- to the table.
# (white box: the m-mer hash of kmer_bulk.hip restated to find a 13-mer that wins wherever it occurs)
def mmer_hash(code, m=13):
    rc = 0
    for j in range(m):
        rc |= (3 - ((code >> (2 * j)) & 3)) << (2 * (m - 1 - j))
    h = ((min(code, rc) + 1) * 0x9E3779B1) & 0xFFFFFFFF
    return h ^ (h >> 15)
cands = rng.integers(0, 4 ** 13, size=200000)
core = int(min(cands, key=lambda c: mmer_hash(int(c))))
core_s = "".join("ACGT"[(core >> (2 * j)) & 3] for j in range(13))
flank = LUT[rng.integers(0, 4, size=(3000, 16), dtype=np.uint8)]
around = ["".join(map(chr, f[:8])) + core_s + "".join(map(chr, f[8:])) for f in flank]
info = run("one minimizer", around + genome_reads(3000, 29, 20000), 21, True, fixed_len=29, expect="partitioned")
assert info["buckets_through_table"] > 0, info
# reads of different lengths through the tiles: more reads in a tile than its table holds (crumbs of 0 .. 6 bases between
# long reads), reads longer than a tile, a byte run that does not start at 0
crumbs = []
for r in genome_reads(300, 400, 20000):
    crumbs.append(r)
    crumbs.extend(r[:int(m)] for m in rng.integers(0, 7, size=int(rng.integers(0, 120))))
run("crumbs", crumbs, 21, True)
run("long reads", genome_reads(200, 1500, 30000), 23, True)
def run_shifted(reads, k):
    blob = np.frombuffer(("ACGTACGTAC" + "".join(reads)).encode(), dtype=np.uint8)
    lens = np.array([len(r) for r in reads], dtype=np.int64)
    offs = np.full(len(reads) + 1, 10, dtype=np.int64); offs[1:] += np.cumsum(lens)
    d_bases, d_offs = torch.from_numpy(blob.copy()).to(dev), torch.from_numpy(offs).to(dev)
    c = kh.KmerCounts(k, canonical=True, min_slots=1 << 12)
    assert c.count_reads_device(d_bases.data_ptr(), len(reads), 0, d_offsets_ptr=d_offs.data_ptr(), n_bases=int(lens.sum())) == "partitioned"
    assert c.histogram() == ko.histogram(reads, k, canonical=True), "offsets[0] != 0"
    c.close()
run_shifted(ragged, 21)
# random shapes: k, strand mode, read length (fixed and ragged), genome size, error rate
for seed in range(16):
    r2 = np.random.default_rng(1000 + seed)
    k = int(r2.integers(19, 32))
    L = int(r2.integers(k, 260))
    n = int(r2.integers(50, 4000))
    reads = genome_reads(n, L, int(r2.integers(L + 1, 60000)), err=float(r2.choice([0.0, 0.01, 0.1])))
    if seed % 2:
        reads = [r[:int(m)] for r, m in zip(reads, r2.integers(0, L + 1, size=n))]
        run("fuzz %d ragged" % seed, reads, k, bool(seed & 2))
    else:
        run("fuzz %d" % seed, reads, k, bool(seed & 2), fixed_len=L, expect="partitioned")
# the smallest inputs: one read of exactly k bases, no read at all, reads of no bases, a read one base short of k
run("one window", ["ACGTTGCAAGGCTTAACCGGT"], 21, True, fixed_len=21)
run("no reads", [], 21, True)
run("empty reads", ["", "", "A"], 21, True)
run("k - 1 bases", ["A" * 20, "ACGTTGCAAGGCTTAACCGG"], 21, False)
# k the partitioned path does not take
run("k=12", genome_reads(3000, 60, 5000), 12, True, fixed_len=60, expect="table")
assert any(p == "partitioned" for _, _, p in taken)
# COVEST_E_NOMEM and what becomes of the memory (ADVICE round 3: a failed partitioned call left its buckets' records
# allocated -- gigabytes at scale -- when the table path the wrapper falls back to needed them).  The caller's cap
# (covest_kmer_memory_limit) makes the call answer COVEST_E_NOMEM before it allocates; the wrapper clears and counts
# through the table, exactly; a partitioned count that did succeed gives its records back at clear().
big = genome_reads(250000, 100, 1500000)          # 25 Mbp: ~4e6 records, ~100 MB of room for them
blob = torch.from_numpy(np.frombuffer("".join(big).encode(), dtype=np.uint8).copy()).to(dev)
want = ko.histogram(big, 21, canonical=True)
torch.cuda.synchronize()
c = kh.KmerCounts(21, canonical=True, min_slots=1 << 12)
free0 = torch.cuda.mem_get_info()[0]
assert c.count_reads_device(blob.data_ptr(), len(big), 100) == "partitioned"
assert c.histogram() == want
held = free0 - torch.cuda.mem_get_info()[0]
assert held > 64 << 20, held                      # the records (and the small per-bucket arrays) are what it holds
c.clear()
torch.cuda.synchronize()
after_clear = free0 - torch.cuda.mem_get_info()[0]
assert after_clear < held - (64 << 20), (held, after_clear)    # ... and clear() gives the records back
c.memory_limit(1 << 20)                           # a cap the buckets cannot meet
assert c.count_reads_device(blob.data_ptr(), len(big), 100) == "table"
assert "limit" in c.why_not_partitioned, c.why_not_partitioned
assert c.histogram() == want                      # the fall-back counted every k-mer
c.memory_limit(0)
c.clear()
assert c.count_reads_device(blob.data_ptr(), len(big), 100) == "partitioned" and c.histogram() == want
c.close()
print("partitioned ok", taken)
"""


def test_partitioned_count_of_resident_reads(hip_lib):
    """covest_kmer_count_reads_device (kmer_bulk.hip: minimizer buckets of super-k-mer records, counted in LDS) against
    the numpy oracle, exact: genome-like reads (both layouts, several k, both strand modes), reads of different
    lengths with some shorter than k, low-complexity reads whose buckets overflow into the table, repeats whose counts
    leave the LDS bins, a k the path declines (the wrapper then counts through the table)."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, COVEST_REPO=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    proc = subprocess.run([sys.executable, "-c", _PARTITIONED_SCRIPT], env=env, capture_output=True, text=True, timeout=900)
    assert proc.returncode == 0 and "partitioned ok" in proc.stdout, proc.stdout[-3000:] + proc.stderr[-4000:]


_SCALE_SCRIPT = r"""
import os, sys
import torch                      # first: ONE HIP runtime per process (INTEGRATION.md 8)
sys.path.insert(0, os.environ["COVEST_REPO"])
from covest_amd import kmer_hist as kh
dev = torch.device("cuda", 0)
free, total = torch.cuda.mem_get_info()
k, L = 21, 100
lut = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=dev)

def synthetic_reads(n_reads, seed):
    # 40x of a random genome, 1 % substitutions: bench.py's C5 input
    gen = torch.Generator(device=dev); gen.manual_seed(seed)
    g_len = max(1_000_000, n_reads * L // 40)
    genome = lut[torch.randint(0, 4, (g_len,), device=dev, generator=gen)]
    reads = torch.empty(n_reads * L, dtype=torch.uint8, device=dev)
    ar = torch.arange(L, device=dev)
    for a in range(0, n_reads, 2_000_000):
        b = min(n_reads, a + 2_000_000)
        starts = torch.randint(0, g_len - L, (b - a,), device=dev, generator=gen)
        r = genome[starts[:, None] + ar[None, :]]
        err = torch.rand(r.shape, device=dev, generator=gen) < 0.01
        r = torch.where(err, lut[torch.randint(0, 4, r.shape, device=dev, generator=gen)], r)
        reads[a * L:b * L] = r.reshape(-1)
    return reads

# 1 Gbp: the partitioned path and the table in HBM give the SAME histogram, bin for bin
n = 10_000_000
reads = synthetic_reads(n, 11)
part = kh.KmerCounts(k, canonical=True, min_slots=1 << 20)
assert part.count_reads_device(reads.data_ptr(), n, L) == "partitioned", getattr(part, "why_not_partitioned", "")
h_part, d_part = part.histogram(), len(part)
info = part.partition_info()
assert info["sampled_1_in"] > 1 and info["records"] > n, info
table = kh.KmerCounts(k, canonical=True, min_slots=1 << 29)
table.add_device(reads.data_ptr(), n, L, reserve=False)
assert table.histogram() == h_part and len(table) == d_part, "1 Gbp: partitioned != table"
assert sum(i * v for i, v in enumerate(h_part)) == n * (L - k + 1) and sum(h_part) == d_part
table.close()
# the same bytes and 1.6 Gbp more, cut into reads of 30 .. 170 bases: 2.6e9 bytes = two launches of the tiles (a launch
# takes less than 2^31); against the table again
more = synthetic_reads(16_000_000, 13)
both = torch.cat([reads, more]); del reads, more
lens = torch.randint(30, 171, (int(both.numel() / 100 * 1.02),), device=dev, dtype=torch.int64)
ends = torch.cumsum(lens, 0)
n_r = int((ends <= both.numel()).sum().item())
offs = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), ends[:n_r]])
assert int(offs[-1].item()) > 2 ** 31
assert part.count_reads_device(both.data_ptr(), n_r, 0, d_offsets_ptr=offs.data_ptr(), n_bases=int(offs[-1].item())) == "partitioned"
h_r, d_r = part.histogram(), len(part)
table = kh.KmerCounts(k, canonical=True, min_slots=1 << 30)
table.add_device(both.data_ptr(), n_r, 0, d_offsets_ptr=offs.data_ptr(), reserve=False)
assert table.histogram() == h_r and len(table) == d_r, "ragged 2.6 Gbp: partitioned != table"
table.close(); del both, offs, ends, lens
# 10 Gbp (BASELINE.json config 5's size): every window lands in exactly one bin, the bins add up to the distinct keys,
# and counting twice gives the same histogram (the counter is emptied, the buckets' room is found again)
if free < 200e9:
    print("scale ok (10 Gbp skipped: %.0f GB of HBM free)" % (free / 1e9)); sys.exit(0)
n = 100_000_000
reads = synthetic_reads(n, 12)
assert part.count_reads_device(reads.data_ptr(), n, L) == "partitioned", getattr(part, "why_not_partitioned", "")
h10, d10 = part.histogram(), len(part)
assert sum(i * v for i, v in enumerate(h10)) == n * (L - k + 1), "10 Gbp: windows lost or counted twice"
assert sum(h10) == d10 and d10 > 10 * d_part // 2
assert part.count_reads_device(reads.data_ptr(), n, L) == "partitioned"
assert part.histogram() == h10
peak = max(range(5, len(h10)), key=lambda i: h10[i])
assert 20 <= peak <= 36, peak        # 40x coverage, 80 of 100 windows a read, 19 % of them hit by an error: ~26
print("scale ok", info, part.partition_info())
"""


def test_partitioned_at_scale(hip_lib):
    """covest_kmer_count_reads_device at sizes no oracle reaches.  1 Gbp (8e8 windows): the histogram equals, bin for
    bin, the one of the open-addressing table in HBM (rounds 1-2, itself checked against the oracle at small sizes).
    10 Gbp (8e9 windows, BASELINE.json config 5; skipped below 200 GB of free HBM): conservation -- sum_i i*h_i = the
    windows, sum_i h_i = the distinct keys --, a second count reproduces the histogram, the coverage peak sits where
    the input puts it."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, COVEST_REPO=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    proc = subprocess.run([sys.executable, "-c", _SCALE_SCRIPT], env=env, capture_output=True, text=True, timeout=900)
    assert proc.returncode == 0 and "scale ok" in proc.stdout, proc.stdout[-3000:] + proc.stderr[-4000:]
