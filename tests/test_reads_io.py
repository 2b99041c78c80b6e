"""The C++ FASTA / FASTQ reader (csrc/reads_io.cpp, covest_reads_* in include/covest_amd.h) against the
reference's load_reads + preprocess semantics (bin/kmer_hist.py:44-54, :67-74).  Host code only: runs without a GPU."""
import ctypes
import random

import pytest

from covest_amd import kmer_hist as kh


def _batches(path, strategy, batch_bases):
    out = []
    for bases, offs, n, n_bases in kh.ReadBatches(str(path), strategy, batch_bases=batch_bases):
        blob = ctypes.string_at(bases, n_bases).decode("ascii")
        assert offs[0] == 0 and offs[n] == n_bases
        out += [blob[offs[i]:offs[i + 1]] for i in range(n)]
    return out


def test_fasta_records_and_n_strategies(tmp_path):
    fa = tmp_path / "r.fa"
    fa.write_text("text before the first header\n>r1 some description\nACGTNACGTACG\r\nTTTGACA\n>empty\n>r2\nNNACGTACGTAC")
    raw = ["ACGTNACGTACGTTTGACA", "", "NNACGTACGTAC"]
    assert list(kh.load_reads(str(fa))) == raw
    for strategy in (kh.NS_IGNORE, kh.NS_SINGLE):
        assert list(kh.load_reads(str(fa), strategy)) == [kh.preprocess(r, strategy) for r in raw]
    rnd = list(kh.load_reads(str(fa), kh.NS_RANDOM))
    assert [len(r) for r in rnd] == [len(r) for r in raw]
    assert all(set(r) <= set("acgt") for r in rnd)
    assert all(a == b.lower() or b == "N" for r, w in zip(rnd, raw) for a, b in zip(r, w))


def test_fastq_records(tmp_path):
    fq = tmp_path / "r.fastq"
    fq.write_text("@a\nACGTACGT\n+\nIIIIIIII\n@b\nTTTTNCGT\n+\n@III>III\n@c\nAC")  # a quality line starting with '@'
    assert list(kh.load_reads(str(fq), kh.NS_IGNORE)) == ["acgtacgt", "ttttcgt", "ac"]
    assert list(kh.load_reads(str(fq), kh.NS_SINGLE)) == ["acgtacgt", "ttttacgt", "ac"]


def test_other_letters_are_a_keyerror(tmp_path):
    for name, text in (("b.fa", ">x\nACGT\nACRT\n"), ("b.fq", "@x\nACGU\n+\nIIII\n")):
        f = tmp_path / name
        f.write_text(text)
        with pytest.raises(KeyError):  # single_hash, bin/kmer_hist.py:15
            list(kh.load_reads(str(f), kh.NS_IGNORE))
    with pytest.raises(ValueError):
        kh.ReadBatches(str(tmp_path / "b.fa"), 7)
    with pytest.raises(Exception):
        kh.ReadBatches(str(tmp_path / "missing.fa"), kh.NS_IGNORE)


@pytest.mark.parametrize("batch_bases", [1, 1000, 1 << 22])
def test_batches_are_whole_reads_whatever_their_size(tmp_path, batch_bases):
    rng = random.Random(5)
    reads = ["".join(rng.choice("ACGTNacgtn") for _ in range(rng.randint(0, 400))) for _ in range(3000)]
    fa = tmp_path / "many.fa"
    with open(fa, "w") as f:
        for i, r in enumerate(reads):
            f.write(">read_%d %s\n" % (i, "x" * rng.randint(0, 50)))
            for j in range(0, len(r), 70):
                f.write(r[j:j + 70] + "\n")
    assert _batches(fa, kh.NS_IGNORE, batch_bases) == [kh.preprocess(r, kh.NS_IGNORE) for r in reads]


def test_a_file_larger_than_the_read_buffer(tmp_path):
    # 8 MiB of file per fread: headers, sequence lines and records all straddle the refills
    rng = random.Random(9)
    line = "".join(rng.choice("ACGT") for _ in range(997))
    fa = tmp_path / "big.fa"
    n = 20000
    with open(fa, "w") as f:
        for i in range(n):
            f.write(">r%d\n%s\n%s\n" % (i, line[i % 100:], "N" * (i % 3)))
    got = _batches(fa, kh.NS_SINGLE, 1 << 20)
    assert len(got) == n
    assert all(got[i] == (line[i % 100:] + "a" * (i % 3)).lower() for i in range(0, n, 97))
    assert sum(len(g) for g in got) == sum(997 - i % 100 + i % 3 for i in range(n))


def test_threads_cut_the_file_at_record_boundaries(tmp_path, monkeypatch):
    """Several threads parse one batch (the span is cut at record starts): the same reads whatever the thread count,
    for FASTA and for a FASTQ whose quality lines keep starting with '@' (the character that also opens a header);
    NS_RANDOM depends on the seed and on where an N stands in the file, not on the threads."""
    rng = random.Random(21)
    reads = ["".join(rng.choice("ACGTN" if i % 50 == 0 else "ACGT") for _ in range(rng.randint(1, 300)))
             for i in range(60000)]
    fa, fq = tmp_path / "t.fa", tmp_path / "t.fq"
    with open(fa, "w") as f:
        for i, r in enumerate(reads):
            f.write(">r%d\n%s\n" % (i, r))
    with open(fq, "w") as f:
        for i, r in enumerate(reads):
            qual = ("@" if i % 3 else "+") + "I" * (len(r) - 1)
            f.write("@r%d\n%s\n+\n%s\n" % (i, r, qual))
    want = [kh.preprocess(r, kh.NS_IGNORE) for r in reads]
    results = {}
    for threads in ("1", "3", "8"):
        monkeypatch.setenv("COVEST_READER_THREADS", threads)
        for path in (fa, fq):
            assert _batches(path, kh.NS_IGNORE, 1 << 22) == want, (threads, path.name)
        results[threads] = _batches(fa, kh.NS_RANDOM, 1 << 22)
    assert results["1"] == results["3"] == results["8"]
    assert [len(r) for r in results["1"]] == [len(r) for r in reads]
    other_seed = []
    for bases, offs, n, n_bases in kh.ReadBatches(str(fa), kh.NS_RANDOM, batch_bases=1 << 22, seed=5):
        blob = ctypes.string_at(bases, n_bases).decode("ascii")
        other_seed += [blob[offs[i]:offs[i + 1]] for i in range(n)]
    assert other_seed != results["1"] and [len(r) for r in other_seed] == [len(r) for r in reads]


def test_degenerate_files(tmp_path):
    empty = tmp_path / "empty.fa"
    empty.write_text("")
    assert list(kh.load_reads(str(empty), kh.NS_IGNORE)) == []
    junk = tmp_path / "junk.fa"
    junk.write_text("no header anywhere\nACGT\n")
    assert list(kh.load_reads(str(junk), kh.NS_IGNORE)) == []
    one = tmp_path / "one.fa"  # one record much longer than a batch, no newline at the end of the file
    rng = random.Random(3)
    seq = "".join(rng.choice("acgt") for _ in range(3_000_000))
    one.write_text(">chr\n" + "\n".join(seq[i:i + 80] for i in range(0, len(seq), 80)))
    got = _batches(one, kh.NS_IGNORE, 1000)
    assert len(got) == 1 and got[0] == seq
    lone = tmp_path / "lone.fq"
    lone.write_text("@r\nACGT")
    assert list(kh.load_reads(str(lone), kh.NS_IGNORE)) == ["acgt"]


def test_fastq_blank_lines_and_malformed_records(tmp_path):
    """Blank lines between records and at the end of the file are skipped (they used to shift the 4-line framing and
    to add an empty read, i.e. k-mer 0); a record that breaks the framing is an error naming its byte offset, never
    garbage that gets counted."""
    fq = tmp_path / "blank.fq"
    fq.write_text("@a\nACGT\n+\nIIII\n\n\n@b\nTTGA\n+\n@III\n\r\n@c\nGG\n+\nII\n\n\n\n")
    assert list(kh.load_reads(str(fq), kh.NS_IGNORE)) == ["acgt", "ttga", "gg"]
    for name, text in (("nohead.fq", "ACGT\n+\nIIII\n"),
                       ("noplus.fq", "@a\nACGT\n\nIIII\n"),
                       ("shortqual.fq", "@a\nACGT\nACGT\n+\nIIIIII\n"),          # the file ends inside the quality
                       ("longqual.fq", "@a\nACGT\n+\nIIII\nII\n@b\nAC\n+\nII\n")):  # a quality line too many
        f = tmp_path / name
        f.write_text(text)
        with pytest.raises(Exception) as err:
            list(kh.load_reads(str(f), kh.NS_IGNORE))
        assert "malformed FASTQ record at byte" in str(err.value)


def test_fastq_with_wrapped_lines(tmp_path):
    """FASTQ whose sequence and quality run over several lines -- Bio.SeqIO, which the reference delegates to
    (covest/data.py:44-54), reads it; rounds 2-3 of this reader refused it.  Quality lines that start with '@' or '+',
    blank lines between records, records of one line among wrapped ones, CRLF, a file that only starts to wrap after
    thousands of 4-line records, batches smaller than a record: the reads are the concatenated sequence lines."""
    rng = random.Random(11)
    want, text = [], []
    for i in range(300):
        seq = "".join(rng.choice("ACGT") for _ in range(rng.randrange(1, 200)))
        width = rng.choice((7, 60, 1000))
        qual = "".join(rng.choice("@+I#5") for _ in seq)
        qwidth = rng.choice((7, 60, 1000))
        eol = rng.choice(("\n", "\r\n"))
        text.append("@r%d%s" % (i, eol))
        text += [seq[a:a + width] + eol for a in range(0, len(seq), width)]
        text.append("+%s" % eol)
        text += [qual[a:a + qwidth] + eol for a in range(0, len(qual), qwidth)]
        if i % 17 == 0:
            text.append(eol)
        want.append(seq.lower())
    fq = tmp_path / "wrapped.fq"
    fq.write_text("".join(text), newline="")
    assert list(kh.load_reads(str(fq), kh.NS_IGNORE)) == want
    assert _batches(fq, kh.NS_IGNORE, 10) == want          # batches smaller than a record
    # 4-line records first (several pieces for the threads), then wrapped ones: the strict parser hands over
    plain = ["".join(rng.choice("acgt") for _ in range(100)) for _ in range(40000)]
    late = tmp_path / "late.fastq"
    late.write_text("".join("@p\n%s\n+\n%s\n" % (s, "I" * 100) for s in plain) + "".join(text), newline="")
    assert list(kh.load_reads(str(late), kh.NS_IGNORE)) == plain + want
    # N strategies see the wrapped sequence as one read
    n = tmp_path / "n.fq"
    n.write_text("@a\nACNN\nNGT\n+\nIIII\nIII\n")
    assert list(kh.load_reads(str(n), kh.NS_IGNORE)) == ["acgt"]
    assert list(kh.load_reads(str(n), kh.NS_SINGLE)) == ["acaaagt"]
    # only the QUALITY wraps, its second line starts with '@', behind more plain records than the first look at the file
    # reads: the 4-line parser takes "@III" for a header and "@x" for a sequence -- a "bad base" that
    # is none; the general grammar reads the file as Bio.SeqIO does
    q = tmp_path / "quality_wraps.fq"
    q.write_text("".join("@p%d\nACGTACGTAC\n+\nIIIIIIIIII\n" % i for i in range(400)) +
                 "@w\nACGTACGT\n+\nIIII\n@III\n@x\nTTTTGGGG\n+\nIIIIIIII\n")
    assert list(kh.load_reads(str(q), kh.NS_IGNORE)) == ["acgtacgtac"] * 400 + ["acgtacgt", "ttttgggg"]
    # ... and a base that IS bad stays an error, whichever grammar looks at it
    b = tmp_path / "bad_base.fq"
    b.write_text("".join("@p%d\nACGTACGTAC\n+\nIIIIIIIIII\n" % i for i in range(400)) + "@w\nACGXACGT\n+\nIIIIIIII\n")
    with pytest.raises(Exception) as err:
        list(kh.load_reads(str(b), kh.NS_IGNORE))
    assert "outside acgtn" in str(err.value) or isinstance(err.value, KeyError)
