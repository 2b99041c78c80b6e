// reads_io.cpp -- the file front-end of the k-mer histogram: FASTA / FASTQ records
// to the packed {bases, offsets} batches covest_kmer_add takes.
//
// Reference restated (paths in the reference checkout):
//   load_reads   bin/kmer_hist.py:67-74   format by extension (.fq / .fastq: FASTQ, anything else FASTA), one
//                                         sequence per record (the reference delegates the parsing to Bio.SeqIO)
//   preprocess   bin/kmer_hist.py:44-54   lower case; N dropped (IGNORE), replaced by 'a' (SINGLE) or by a random
//                                         base (RANDOM)
//   single_hash  bin/kmer_hist.py:14-15   any other letter is a KeyError: here COVEST_E_INVALID naming the letter
// The reference does this per character in Python (a generator, str.join, a dict lookup per base); here one
// table-driven pass over the file's bytes writes the batch -- no per-read work on the Python side at all.
// Host code only: nothing here touches the GPU.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/covest_amd.h"

namespace covest {
int set_error(int code, const std::string &msg); // capi.cpp: records the message for covest_last_error
}

namespace {

enum : uint8_t { kBase = 0, kN = 1, kSpace = 2, kBad = 3 };

struct ByteClass {
    uint8_t cls[256];
    uint8_t lower[256];
    ByteClass()
    {
        for (int c = 0; c < 256; ++c) {
            cls[c] = kBad;
            lower[c] = (uint8_t)((c >= 'A' && c <= 'Z') ? c + 32 : c);
        }
        for (const char *p = "acgtACGT"; *p; ++p)
            cls[(uint8_t)*p] = kBase;
        cls[(uint8_t)'n'] = cls[(uint8_t)'N'] = kN;
        for (const char *p = " \t\r\n\v\f"; *p; ++p)
            cls[(uint8_t)*p] = kSpace;
    }
};
const ByteClass kBytes;

} // namespace

struct covest_reads {
    FILE *f = nullptr;
    bool fastq = false;
    int n_strategy = 0;
    uint64_t rng = 0;
    std::vector<uint8_t> buf; // file bytes not consumed yet: [pos, end)
    size_t pos = 0, end = 0;
    bool eof = false;
    // parser state, kept across buffer refills
    bool in_record = false;   // FASTA: a header has been seen, its sequence is open
    bool in_header = false;   // inside a header line
    bool at_line_start = true;
    int fq_line = 0;          // FASTQ: line of the record (0 = @id, 1 = sequence, 2 = +, 3 = quality)
    // (a plain growable buffer: std::vector::resize would zero-fill every byte before it is written)
    uint8_t *base_buf = nullptr;
    size_t n_bases = 0, cap_bases = 0;
    ~covest_reads() { std::free(base_buf); }
    uint8_t *grow(size_t extra)
    {
        if (n_bases + extra > cap_bases) {
            size_t cap = cap_bases ? cap_bases : (size_t)1 << 20;
            while (cap < n_bases + extra)
                cap *= 2;
            uint8_t *nb = static_cast<uint8_t *>(std::realloc(base_buf, cap));
            if (!nb)
                throw std::bad_alloc();
            base_buf = nb;
            cap_bases = cap;
        }
        return base_buf + n_bases;
    }
    std::vector<int64_t> offsets;
    int64_t records = 0, bytes_read = 0;
};

namespace {

uint64_t next_random(uint64_t &s) // splitmix64
{
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

bool refill(covest_reads *r)
{
    if (r->eof)
        return false;
    r->pos = 0;
    r->end = std::fread(r->buf.data(), 1, r->buf.size(), r->f);
    r->bytes_read += (int64_t)r->end;
    if (r->end == 0)
        r->eof = true;
    return r->end > 0;
}

// One sequence byte into the open read.  Returns false on a letter single_hash would reject.
inline bool put_base(covest_reads *r, uint8_t ch)
{
    const uint8_t c = kBytes.cls[ch];
    if (c == kBase) {
        *r->grow(1) = kBytes.lower[ch];
        ++r->n_bases;
        return true;
    }
    if (c == kN) {
        if (r->n_strategy == 1) {
            *r->grow(1) = (uint8_t)'a';
            ++r->n_bases;
        } else if (r->n_strategy == 2) {
            *r->grow(1) = (uint8_t)"acgt"[next_random(r->rng) & 3];
            ++r->n_bases;
        }
        return true; // IGNORE: dropped
    }
    return c == kSpace;
}

// A run of sequence bytes [p, e) without a newline: the common case -- nothing but a/c/g/t -- is one pass that
// lower-cases into place; anything else goes through put_base byte by byte.  Returns the offending byte's
// position, or nullptr.
inline const uint8_t *put_span(covest_reads *r, const uint8_t *p, const uint8_t *e)
{
    const size_t n = (size_t)(e - p);
    uint8_t *out = r->grow(n);
    uint8_t seen = 0;
    for (size_t i = 0; i < n; ++i) {
        out[i] = kBytes.lower[p[i]];
        seen |= kBytes.cls[p[i]];
    }
    if (seen == kBase) {
        r->n_bases += n;
        return nullptr;
    }
    for (; p < e; ++p)
        if (!put_base(r, *p))
            return p;
    return nullptr;
}

} // namespace

extern "C" {

int covest_reads_open(const char *path, int32_t n_strategy, uint64_t seed, covest_reads **out)
{
    if (!path || !out)
        return covest::set_error(COVEST_E_INVALID, "covest_reads_open: null argument");
    if (n_strategy < 0 || n_strategy > 2)
        return covest::set_error(COVEST_E_INVALID, "covest_reads_open: invalid N strategy (0 IGNORE, 1 SINGLE, 2 RANDOM)");
    FILE *f = std::fopen(path, "rb");
    if (!f)
        return covest::set_error(COVEST_E_INVALID, std::string("covest_reads_open: cannot open ") + path);
    covest_reads *r = new covest_reads;
    r->f = f;
    const char *dot = std::strrchr(path, '.');
    const char *slash = std::strrchr(path, '/');
    if (dot && (!slash || dot > slash))
        r->fastq = std::strcmp(dot, ".fq") == 0 || std::strcmp(dot, ".fastq") == 0;
    r->n_strategy = n_strategy;
    r->rng = seed;
    r->buf.resize((size_t)8 << 20);
    r->offsets.push_back(0);
    *out = r;
    return COVEST_OK;
}

void covest_reads_close(covest_reads *r)
{
    if (!r)
        return;
    if (r->f)
        std::fclose(r->f);
    delete r;
}

int covest_reads_next(covest_reads *r, int64_t max_bases, const uint8_t **bases, const int64_t **offsets,
                      int64_t *n_reads)
{
    if (!r || !bases || !offsets || !n_reads)
        return covest::set_error(COVEST_E_INVALID, "covest_reads_next: null argument");
    if (max_bases < 1)
        max_bases = 1;
    try {
    // a record left open by the previous batch (FASTA: its sequence may go on) moves to the front
    const size_t closed = (size_t)r->offsets.back();
    std::memmove(r->base_buf, r->base_buf + closed, r->n_bases - closed);
    r->n_bases -= closed;
    r->offsets.assign(1, 0);
    bool full = false;
    auto close_read = [&]() {
        r->offsets.push_back((int64_t)r->n_bases);
        ++r->records;
        if ((int64_t)r->n_bases >= max_bases)
            full = true;
    };
    while (!full) {
        if (r->pos == r->end && !refill(r))
            break;
        const uint8_t *p = r->buf.data() + r->pos, *e = r->buf.data() + r->end;
        if (!r->fastq) {
            while (p < e && !full) {
                const uint8_t ch = *p++;
                if (r->in_header) {
                    if (ch != '\n') { // skip to the end of the header line (or of the buffer)
                        const uint8_t *nl = static_cast<const uint8_t *>(std::memchr(p, '\n', (size_t)(e - p)));
                        if (!nl) {
                            p = e;
                            continue;
                        }
                        p = nl + 1;
                    }
                    r->in_header = false;
                    r->at_line_start = true;
                    continue;
                }
                if (r->at_line_start && ch == '>') {
                    if (r->in_record)
                        close_read(); // (an empty record is a read too: it counts k-mer 0, bin/kmer_hist.py:36-37)
                    r->in_record = true;
                    r->in_header = true;
                    r->at_line_start = false;
                    continue;
                }
                r->at_line_start = ch == '\n';
                if (!r->in_record || ch == '\n')
                    continue; // (text before the first header)
                // the rest of the line that is in the buffer, in one go
                --p;
                const uint8_t *nl = static_cast<const uint8_t *>(std::memchr(p, '\n', (size_t)(e - p)));
                const uint8_t *stop = nl ? nl : e;
                if (const uint8_t *bad = put_span(r, p, stop)) {
                    r->pos = (size_t)(bad + 1 - r->buf.data());
                    return covest::set_error(COVEST_E_INVALID, std::string("covest_reads_next: base '") + (char)*bad +
                                                              "' outside acgtn (single_hash raises KeyError)");
                }
                p = stop;
            }
        } else {
            while (p < e && !full) {
                const uint8_t ch = *p++;
                if (ch == '\n') {
                    if (r->fq_line == 1)
                        close_read();
                    r->fq_line = (r->fq_line + 1) & 3;
                    continue;
                }
                // the rest of the line that is in the buffer: sequence bytes in one go, the other lines skipped
                --p;
                const uint8_t *nl = static_cast<const uint8_t *>(std::memchr(p, '\n', (size_t)(e - p)));
                const uint8_t *stop = nl ? nl : e;
                if (r->fq_line == 1) {
                    if (const uint8_t *bad = put_span(r, p, stop)) {
                        r->pos = (size_t)(bad + 1 - r->buf.data());
                        return covest::set_error(COVEST_E_INVALID, std::string("covest_reads_next: base '") + (char)*bad +
                                                                  "' outside acgtn (single_hash raises KeyError)");
                    }
                }
                p = stop;
            }
        }
        r->pos = (size_t)(p - r->buf.data());
    }
    if (!full && r->eof) { // the end of the file closes what is open
        if (!r->fastq && r->in_record) {
            close_read();
            r->in_record = false;
        } else if (r->fastq && r->fq_line == 1 && (int64_t)r->n_bases > r->offsets.back()) {
            close_read(); // a last sequence line without its newline
            r->fq_line = 2;
        }
    }
    } catch (const std::bad_alloc &) {
        return covest::set_error(COVEST_E_NOMEM, "covest_reads_next: out of host memory");
    }
    *bases = r->base_buf ? r->base_buf : reinterpret_cast<const uint8_t *>("");
    *offsets = r->offsets.data();
    *n_reads = (int64_t)r->offsets.size() - 1;
    return COVEST_OK;
}

int64_t covest_reads_bytes(const covest_reads *r) { return r ? r->bytes_read : 0; }

} // extern "C"
