"""k-mer abundance histogram on the GPU: the counterpart of the reference's bin/kmer_hist.py
(BASELINE.json config 5).

Same functions, same argument meaning: `preprocess` (:44-54), `compute_counts(seq,
prev_counts=None, k=20)` (:34-41), `compute_histogram(counts)` (:57-64), `main` (:77-89), plus
the 2-bit hash helpers (:14-31).  The `counts` object is a hash table in HBM behind the C ABI
(covest_kmer_* in include/covest_amd.h) instead of a Python dict; every k-mer is counted by the
HIP kernel kmer_count.hip.  There is no CPU fallback.

Differences from the reference, all deliberate and listed in DESIGN.md:
  * `main` passes its `k` on (the reference's main ignores -k and always counts 20-mers, :81);
  * `canonical=True` (jellyfish -C: a k-mer and its reverse complement are one key) is offered
    because BASELINE.json's config 5 asks for it; the default is the reference's forward strand;
  * k <= 255: k <= 31 packs key and empty marker into one 64-bit word (kmer_count.hip, the fast path); larger k takes
    keys of 2, 4 or 8 words (kmer_wide.hip) -- the reference's Python integers have no limit; counts are 64-bit.
"""
import ctypes
import random
from os import path

import numpy as np

from . import _capi

NS_IGNORE = 0
NS_SINGLE = 1
NS_RANDOM = 2

_CODES = {'a': 0, 'c': 1, 'g': 2, 't': 3}
_VALID = np.zeros(256, dtype=bool)
for _ch in "acgtACGT":
    _VALID[ord(_ch)] = True


def single_hash(b):
    """bin/kmer_hist.py:14-15."""
    return _CODES[b]


def hash_kmer(kmer):
    """bin/kmer_hist.py:18-23."""
    h = 0
    for b in kmer:
        h <<= 2
        h |= single_hash(b)
    return h


def rehash(old_hash, b, k):
    """bin/kmer_hist.py:26-31."""
    h = old_hash & ((1 << (2 * k - 2)) - 1)
    h <<= 2
    h |= single_hash(b)
    return h


def preprocess(seq, nstrategy=NS_IGNORE):
    """bin/kmer_hist.py:44-54 (host string work, as in the reference)."""
    seq = str(seq).lower()
    if nstrategy == NS_IGNORE:
        seq = seq.replace('n', '')
    elif nstrategy == NS_SINGLE:
        seq = seq.replace('n', 'a')
    elif nstrategy == NS_RANDOM:
        seq = ''.join(random.choice(['a', 'c', 'g', 't']) if b == 'n' else b for b in seq)
    else:
        raise ValueError('Invalid N strategy')
    return seq


def scatter_rate(slots, ops=1 << 28, device=0):
    """Returning 64-bit atomic adds per second at pseudo-random places of a `slots`-word array on this device
    (covest_kmer_scatter_rate): the measured roof of pass 1 of the partitioned path."""
    out = ctypes.c_double()
    _capi.check(_capi.lib().covest_kmer_scatter_rate(int(device), int(slots), int(ops), ctypes.byref(out)),
                "covest_kmer_scatter_rate")
    return out.value


class KmerCounts:
    """The `counts` of compute_counts: an open-addressing table in HBM (covest_kmer*)."""

    def __init__(self, k=20, canonical=False, device=-1, min_slots=1 << 16):
        self.k = int(k)
        self.canonical = bool(canonical)
        self._distinct = 0  # distinct k-mers in the table when last measured
        self._added = 0     # k-mer occurrences inserted since: _distinct + _added bounds the distinct count
        h = ctypes.c_void_p()
        _capi.check(_capi.lib().covest_kmer_create(self.k, 1 if canonical else 0, int(min_slots),
                                                   int(device), ctypes.byref(h)), "covest_kmer_create")
        self._handle = h

    def close(self):
        if getattr(self, "_handle", None) is not None:
            _capi.lib().covest_kmer_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def slots(self):
        return int(_capi.lib().covest_kmer_slots(self._handle))

    def _reserve_for(self, n_new):
        """Keep the table at most half full of DISTINCT k-mers.  The distinct count is bounded by what was
        measured last plus every occurrence inserted since; only when that bound no longer fits is the table
        asked (one small kernel), and it grows for the measured count plus the batch about to be added --
        never for the cumulative number of occurrences (10 Gbp of reads of a 100 Mbp genome need a table for
        ~1e8 k-mers, not 1e10)."""
        n_new = int(n_new)
        if 2 * (self._distinct + self._added + n_new) + 1024 > self.slots:
            self._distinct, self._added = self.stats()[1], 0
            _capi.check(_capi.lib().covest_kmer_reserve(self._handle, 2 * (self._distinct + n_new) + 1024),
                        "covest_kmer_reserve")
        self._added += n_new

    def add_reads(self, reads):
        """Count the k-mers of preprocessed reads (only a/c/g/t, either case) in one launch."""
        reads = [r if isinstance(r, str) else str(r) for r in reads]
        if not reads:
            return self
        lens = np.fromiter((len(r) for r in reads), dtype=np.int64, count=len(reads))
        offsets = np.zeros(len(reads) + 1, dtype=np.int64)
        np.cumsum(lens, out=offsets[1:])
        blob = np.frombuffer("".join(reads).encode("ascii"), dtype=np.uint8)
        if blob.size and not _VALID[blob].all():
            raise KeyError("base outside acgt")  # single_hash raises KeyError (bin/kmer_hist.py:15)
        self._reserve_for(int(np.maximum(lens - self.k + 1, 1).sum()))
        blob = np.ascontiguousarray(blob)
        _capi.check(_capi.lib().covest_kmer_add(
            self._handle, blob.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)),
            offsets.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), len(reads)), "covest_kmer_add")
        return self

    def add_packed(self, bases, offsets, n_reads, n_bases):
        """Preprocessed reads in the packed host layout of covest_kmer_add (ctypes pointers: the bases back to
        back, offsets[n_reads + 1]) -- what ReadBatches yields."""
        if n_reads <= 0:
            return self
        # k-mers the batch can add: sum(max(len - k + 1, 1)) <= n_bases + n_reads
        self._reserve_for(int(n_bases) + int(n_reads))
        _capi.check(_capi.lib().covest_kmer_add(self._handle, bases, offsets, int(n_reads)), "covest_kmer_add")
        return self

    def add_device(self, d_bases_ptr, n_reads, read_len, d_offsets_ptr=None, stream=None, reserve=True):
        """Reads already resident in HBM (raw device pointers); asynchronous on `stream`.
        reserve=False: the caller sized the table for the DISTINCT k-mers it expects (an
        overflow is reported by the next histogram call)."""
        n_new = int(n_reads) * max(int(read_len) - self.k + 1, 1)
        if reserve:
            self._reserve_for(n_new)
        else:  # still occurrences inserted since the last measurement: later batches must see them in the bound
            self._added += n_new
        _capi.require_shared_runtime("KmerCounts.add_device")
        _capi.check(_capi.lib().covest_kmer_add_device(
            self._handle, ctypes.c_void_p(d_bases_ptr), ctypes.c_void_p(d_offsets_ptr or 0),
            int(n_reads), int(read_len), ctypes.c_void_p(stream or 0)), "covest_kmer_add_device")
        return self

    def count_reads_device(self, d_bases_ptr, n_reads, read_len, d_offsets_ptr=None, n_bases=None, stream=None):
        """ALL the k-mers of reads resident in HBM into this (emptied) counter -- the loop of main,
        bin/kmer_hist.py:77-89 -- by the partitioned path where it applies (covest_kmer_count_reads_device: k = 19..31,
        no read shorter than k, buckets that fit the device), else through the table (clear + add_device).  Afterwards
        the counter answers histogram() / len() exactly; after the partitioned path it holds no dict to add to
        (clear() first).  Returns the name of the path taken."""
        n_bases = int(n_bases if n_bases is not None else int(n_reads) * max(int(read_len), 0))
        _capi.require_shared_runtime("KmerCounts.count_reads_device")
        rc = _capi.lib().covest_kmer_count_reads_device(
            self._handle, ctypes.c_void_p(d_bases_ptr), ctypes.c_void_p(d_offsets_ptr or 0), int(n_reads), int(read_len),
            n_bases, ctypes.c_void_p(stream or 0))
        if rc == 0:
            self._distinct = self._added = 0
            return "partitioned"
        if rc not in (_capi.COVEST_E_UNSUPPORTED, _capi.COVEST_E_NOMEM):
            _capi.check(rc, "covest_kmer_count_reads_device")
        self.why_not_partitioned = _capi.last_error()
        self.clear(stream)
        if d_offsets_ptr:
            self._reserve_for(n_bases + int(n_reads))
            self.add_device(d_bases_ptr, n_reads, 0, d_offsets_ptr=d_offsets_ptr, stream=stream, reserve=False)
        else:
            self.add_device(d_bases_ptr, n_reads, read_len, stream=stream)
        return "table"

    def memory_limit(self, max_bytes):
        """Bytes the partitioned path may allocate for its buckets' records (0: no cap but the free device memory); a
        count_reads_device that would pass it takes the table path instead (covest_kmer_memory_limit)."""
        _capi.check(_capi.lib().covest_kmer_memory_limit(self._handle, int(max_bytes)), "covest_kmer_memory_limit")
        return self

    def partition_info(self):
        """How the last partitioned count went (covest_kmer_partition_info), as a dict."""
        out = (ctypes.c_int64 * 8)()
        _capi.check(_capi.lib().covest_kmer_partition_info(self._handle, out), "covest_kmer_partition_info")
        names = ("buckets", "minimizer", "sampled_1_in", "room_records", "overflowed_records", "buckets_by_workgroup",
                 "buckets_through_table", "records")
        info = dict(zip(names, list(out)))
        ms = (ctypes.c_double * 4)()
        _capi.check(_capi.lib().covest_kmer_partition_ms(self._handle, ms), "covest_kmer_partition_ms")
        info["ms"] = dict(zip(("place", "scatter", "count", "table"), [round(v, 4) for v in ms]))
        return info

    def clear(self, stream=None):
        """Drop every count but keep the table (a fresh `defaultdict(int)` of the same size)."""
        self._distinct = self._added = 0
        _capi.check(_capi.lib().covest_kmer_clear(self._handle, ctypes.c_void_p(stream or 0)),
                    "covest_kmer_clear")

    def stats(self):
        """(max count + 1, distinct k-mers)."""
        need, distinct = ctypes.c_int64(), ctypes.c_int64()
        _capi.check(_capi.lib().covest_kmer_histogram(self._handle, None, 0, ctypes.byref(need),
                                                      ctypes.byref(distinct)), "covest_kmer_histogram")
        return need.value, distinct.value

    def histogram(self):
        need, _ = self.stats()
        out = np.zeros(need, dtype=np.int64)
        _capi.check(_capi.lib().covest_kmer_histogram(
            self._handle, out.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), need, None, None),
            "covest_kmer_histogram")
        return out.tolist()

    def __len__(self):
        return self.stats()[1]


def compute_counts(seq, prev_counts=None, k=20, canonical=False):
    """bin/kmer_hist.py:34-41.  `seq` may also be a list of reads (one launch for all)."""
    counts = KmerCounts(k, canonical=canonical) if prev_counts is None else prev_counts
    counts.add_reads([seq] if isinstance(seq, str) else list(seq))
    return counts


def compute_histogram(counts):
    """bin/kmer_hist.py:57-64: [number of k-mers seen i times for i in 0..max count]."""
    return counts.histogram()


class ReadBatches:
    """load_reads + preprocess (bin/kmer_hist.py:67-74, :44-54) in the library's C++ reader (csrc/reads_io.cpp):
    iterating yields (bases_ptr, offsets_ptr, n_reads, n_bases) batches in the packed layout covest_kmer_add takes
    -- no per-read work in Python.  The pointers are the reader's own buffers, valid until the next batch."""

    def __init__(self, fname, n_strategy=NS_IGNORE, batch_bases=1 << 26, seed=0):
        if n_strategy not in (NS_IGNORE, NS_SINGLE, NS_RANDOM):
            raise ValueError('Invalid N strategy')
        self.batch_bases = int(batch_bases)
        h = ctypes.c_void_p()
        _capi.check(_capi.lib().covest_reads_open(str(fname).encode(), int(n_strategy), int(seed), ctypes.byref(h)),
                    "covest_reads_open")
        self._handle = h

    def close(self):
        if getattr(self, "_handle", None) is not None:
            _capi.lib().covest_reads_close(self._handle)
            self._handle = None

    __del__ = close

    @property
    def bytes_read(self):
        return int(_capi.lib().covest_reads_bytes(self._handle))

    def _next(self):
        """One covest_reads_next call: (bases, offsets, n_reads, n_bases), or an exception instance."""
        L = _capi.lib()
        bases = ctypes.POINTER(ctypes.c_uint8)()
        offsets = ctypes.POINTER(ctypes.c_int64)()
        n = ctypes.c_int64()
        rc = L.covest_reads_next(self._handle, self.batch_bases, ctypes.byref(bases), ctypes.byref(offsets),
                                 ctypes.byref(n))
        if rc != 0:
            msg = _capi.last_error()  # (thread-local in the library: read it on the thread that made the call)
            if rc == _capi.COVEST_E_INVALID and "outside acgtn" in msg:
                return KeyError(msg)  # single_hash raises KeyError (bin/kmer_hist.py:15)
            return _capi.CovestHipError("covest_reads_next failed (%d): %s" % (rc, msg))
        return bases, offsets, n.value, int(offsets[n.value])

    def __iter__(self):
        """Batches in file order.  The reader keeps two batch buffers, so the NEXT batch is parsed on a helper
        thread (the library's own threads do the work; ctypes releases the GIL) while the caller -- the GPU --
        is busy with the one just handed out."""
        import concurrent.futures
        with concurrent.futures.ThreadPoolExecutor(max_workers=1) as pool:
            pending = pool.submit(self._next)
            while True:
                got = pending.result()
                if isinstance(got, Exception):
                    raise got
                if got[2] == 0:
                    return
                pending = pool.submit(self._next)  # fills the OTHER buffer: `got` stays valid until the call after
                yield got


def load_reads(fname, n_strategy=None):
    """Sequences of a FASTA or FASTQ file, one str per record (bin/kmer_hist.py:67-74; the reference delegates the
    parsing to Bio.SeqIO).  With n_strategy they come preprocessed (:44-54).  Kept for callers that want strings:
    `main` feeds the packed batches of ReadBatches to the counter directly."""
    if n_strategy is None:  # the records as they stand (the reader always applies preprocess): parsed here
        _, ext = path.splitext(fname)
        fastq = ext in ('.fq', '.fastq')
        with open(fname) as f:
            if fastq:
                for i, line in enumerate(f):
                    if i % 4 == 1:
                        yield line.strip()
            else:
                chunk, seen = [], False
                for line in f:
                    if line.startswith('>'):
                        if seen:
                            yield ''.join(chunk)
                        chunk, seen = [], True
                    elif seen:
                        chunk.append(''.join(line.split()))
                if seen:
                    yield ''.join(chunk)
        return
    for bases, offsets, n, n_bases in ReadBatches(fname, n_strategy, batch_bases=1 << 24):
        blob = ctypes.string_at(bases, n_bases).decode("ascii")
        for i in range(n):
            yield blob[offsets[i]:offsets[i + 1]]


def main(fname, out_fname, k, n_strategy, canonical=False, batch=1 << 26):
    """bin/kmer_hist.py:77-89.  The file is parsed and preprocessed by the C++ reader, `batch` bases per launch."""
    counts = KmerCounts(k, canonical=canonical)
    for bases, offsets, n, n_bases in ReadBatches(fname, n_strategy, batch_bases=batch):
        counts.add_packed(bases, offsets, n, n_bases)
    hist = compute_histogram(counts)
    if out_fname:
        with open(out_fname, 'w') as f:
            for i, v in enumerate(hist):
                f.write('{} {}\n'.format(i, v))
    else:
        print(hist)
    return hist
