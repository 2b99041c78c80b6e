// reads_io.cpp -- the file front-end of the k-mer histogram: FASTA / FASTQ records
// to the packed {bases, offsets} batches covest_kmer_add takes.
//
// Reference restated (paths in the reference checkout):
//   load_reads   bin/kmer_hist.py:67-74   format by extension (.fq / .fastq: FASTQ, anything else FASTA), one
//                                         sequence per record (the reference delegates the parsing to Bio.SeqIO)
//   preprocess   bin/kmer_hist.py:44-54   lower case; N dropped (IGNORE), replaced by 'a' (SINGLE) or by a random
//                                         base (RANDOM)
//   single_hash  bin/kmer_hist.py:14-15   any other letter is a KeyError: here COVEST_E_INVALID naming the letter
// The reference does this per character in Python (a generator, str.join, a dict lookup per base).  Here the file
// is mapped, a batch is a span of it that ends on a record boundary, the span is cut at record boundaries into one
// piece per thread, every piece is parsed by a table-driven pass over its bytes, and the pieces are copied side by
// side into the batch -- which lives in page-locked memory when the process has a HIP device, so that
// covest_kmer_add's copy to the device runs at the speed of the bus.  Two batch buffers alternate: a batch stays
// valid while the next one is being produced (a caller can parse batch i + 1 while the GPU counts batch i).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#if defined(__SSE2__)
#include <emmintrin.h>
#endif
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "../../include/covest_amd.h"

namespace covest {
int set_error(int code, const std::string &msg); // host_common.cpp: records the message for covest_last_error
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

// a plain growable byte buffer (std::vector::resize would zero-fill every byte before it is written)
struct Bytes {
    uint8_t *p = nullptr;
    size_t n = 0, cap = 0;
    Bytes() = default;
    Bytes(const Bytes &) = delete;
    Bytes &operator=(const Bytes &) = delete;
    Bytes(Bytes &&o) noexcept : p(o.p), n(o.n), cap(o.cap) { o.p = nullptr, o.n = o.cap = 0; }
    ~Bytes() { std::free(p); }
    uint8_t *grow(size_t extra)
    {
        if (n + extra > cap) {
            size_t c = cap ? cap : (size_t)1 << 16;
            while (c < n + extra)
                c *= 2;
            uint8_t *q = static_cast<uint8_t *>(std::realloc(p, c));
            if (!q)
                throw std::bad_alloc();
            p = q;
            cap = c;
        }
        return p + n;
    }
};

// what one thread makes of its piece of the span
struct Piece {
    Bytes bases;
    std::vector<int64_t> lens;   // one per record, in file order
    const uint8_t *bad = nullptr; // first letter single_hash would reject
    const uint8_t *malformed = nullptr; // FASTQ: first line that breaks the 4-line framing ('@' / '+' expected)
    bool oom = false;
};

// a batch handed to the caller: page-locked when a HIP device is there, plain memory otherwise
struct Batch {
    uint8_t *bases = nullptr;
    size_t cap = 0;
    bool pinned = false;
    std::vector<int64_t> offsets;
    void release()
    {
        if (bases) {
            if (pinned)
                (void)hipHostFree(bases);
            else
                std::free(bases);
        }
        bases = nullptr;
        cap = 0;
    }
    bool reserve(size_t n, bool want_pinned)
    {
        if (n <= cap)
            return true;
        release();
        size_t c = (size_t)1 << 20;
        while (c < n)
            c *= 2;
        if (want_pinned && hipHostMalloc(reinterpret_cast<void **>(&bases), c, hipHostMallocDefault) == hipSuccess) {
            pinned = true;
        } else {
            (void)hipGetLastError();
            bases = static_cast<uint8_t *>(std::malloc(c));
            pinned = false;
        }
        cap = bases ? c : 0;
        return bases != nullptr;
    }
};

inline uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

} // namespace

struct covest_reads {
    int fd = -1;
    const uint8_t *map = nullptr;
    size_t size = 0, pos = 0; // pos: the next unparsed byte -- the file's start, or a record's first byte
    bool fastq = false;
    bool wrapped = false; // FASTQ whose sequence / quality run over several lines (Bio.SeqIO reads those too): the general
                          // grammar, one record after the other, single-threaded (see parse_fastq_wrapped)
    const uint8_t *strict_malformed = nullptr; // where the 4-line parser gave up and handed over to the general grammar
    const uint8_t *strict_bad = nullptr;       // ... or the "base" it refused there (a quality line it took for a sequence)
    int n_strategy = 0;
    uint64_t seed = 0;
    int n_threads = 1;
    bool want_pinned = false;
    Batch batch[2];
    int cur = 0;
    std::vector<Piece> pieces; // the threads' buffers, kept from batch to batch (their pages stay touched)
    ~covest_reads()
    {
        batch[0].release();
        batch[1].release();
        if (map && size)
            ::munmap(const_cast<uint8_t *>(map), size);
        if (fd >= 0)
            ::close(fd);
    }
};

namespace {

// The sequence bytes [p, e) of one line into the piece.  The common case -- nothing but a/c/g/t -- is one pass that
// lower-cases into place; anything else goes byte by byte.  Returns false at a letter outside acgtn (piece.bad).
inline bool put_line(const covest_reads *r, Piece &pc, const uint8_t *p, const uint8_t *e)
{
    const size_t n = (size_t)(e - p);
    uint8_t *out = pc.bases.grow(n);
    uint8_t seen = 0;
    size_t i = 0;
#if defined(__SSE2__)
    // 16 bytes at a time: c | 0x20 lower-cases a letter; the line is clean if every byte then is one of a c g t
    const __m128i bit5 = _mm_set1_epi8(0x20), la = _mm_set1_epi8('a'), lc = _mm_set1_epi8('c'),
                  lg = _mm_set1_epi8('g'), lt = _mm_set1_epi8('t');
    __m128i all_ok = _mm_set1_epi8((char)0xFF);
    for (; i + 16 <= n; i += 16) {
        const __m128i v = _mm_or_si128(_mm_loadu_si128(reinterpret_cast<const __m128i *>(p + i)), bit5);
        const __m128i ok = _mm_or_si128(_mm_or_si128(_mm_cmpeq_epi8(v, la), _mm_cmpeq_epi8(v, lc)),
                                        _mm_or_si128(_mm_cmpeq_epi8(v, lg), _mm_cmpeq_epi8(v, lt)));
        all_ok = _mm_and_si128(all_ok, ok);
        _mm_storeu_si128(reinterpret_cast<__m128i *>(out + i), v);
    }
    if (_mm_movemask_epi8(all_ok) != 0xFFFF)
        seen = kBad; // (something else in there: sorted out byte by byte below)
#endif
    for (; i < n; ++i) {
        out[i] = kBytes.lower[p[i]];
        seen |= kBytes.cls[p[i]];
    }
    if (seen == kBase) {
        pc.bases.n += n;
        return true;
    }
    size_t w = 0;
    for (; p < e; ++p) {
        const uint8_t c = kBytes.cls[*p];
        if (c == kBase) {
            out[w++] = kBytes.lower[*p];
        } else if (c == kN) {
            if (r->n_strategy == 1)
                out[w++] = (uint8_t)'a';
            else if (r->n_strategy == 2) // a function of the seed and of WHERE the N stands: the same whatever the threads
                out[w++] = (uint8_t)"acgt"[mix64(r->seed + 0x9E3779B97F4A7C15ull * (uint64_t)(p - r->map + 1)) & 3];
            // IGNORE: dropped
        } else if (c != kSpace) {
            pc.bad = p;
            return false;
        }
    }
    pc.bases.n += w;
    return true;
}

// Whole records of [b, e): b is the file's start (FASTA: text before the first header is skipped) or a record's
// first byte, e a record's first byte or the end of the file.
void parse_piece(const covest_reads *r, const uint8_t *b, const uint8_t *e, Piece &pc)
{
    try {
        const uint8_t *p = b;
        if (!r->fastq) {
            bool in_record = false;
            int64_t start = 0;
            while (p < e) {
                const uint8_t *nl = static_cast<const uint8_t *>(std::memchr(p, '\n', (size_t)(e - p)));
                const uint8_t *stop = nl ? nl : e;
                if (*p == '>') {
                    if (in_record) // (an empty record is a read too: it counts k-mer 0, bin/kmer_hist.py:36-37)
                        pc.lens.push_back((int64_t)pc.bases.n - start);
                    in_record = true;
                    start = (int64_t)pc.bases.n;
                } else if (in_record && !put_line(r, pc, p, stop)) {
                    return;
                }
                p = nl ? nl + 1 : e;
            }
            if (in_record)
                pc.lens.push_back((int64_t)pc.bases.n - start);
        } else {
            // Strict 4-line records (what sequencers write, and what can be cut into pieces for the threads; a file with
            // wrapped sequence lines goes through parse_fastq_wrapped instead): @id / sequence / + / quality.  Blank lines between records
            // (and at the end of the file) are skipped; a record whose first line does not start with '@' or whose
            // third does not start with '+' is reported with its byte offset instead of being counted as garbage.
            int line = 0; // 0 = @id, 1 = sequence, 2 = +, 3 = quality
            while (p < e) {
                const uint8_t *nl = static_cast<const uint8_t *>(std::memchr(p, '\n', (size_t)(e - p)));
                const uint8_t *stop = nl ? nl : e;
                const bool blank = stop == p || (stop == p + 1 && *p == '\r');
                if (line == 0 && blank) { // between records
                    p = nl ? nl + 1 : e;
                    continue;
                }
                if ((line == 0 && *p != '@') || (line == 2 && (blank || *p != '+'))) {
                    pc.malformed = p;
                    return;
                }
                if (line == 1) {
                    const int64_t start = (int64_t)pc.bases.n;
                    if (!put_line(r, pc, p, stop))
                        return;
                    pc.lens.push_back((int64_t)pc.bases.n - start);
                }
                line = (line + 1) & 3;
                p = nl ? nl + 1 : e;
            }
        }
    } catch (const std::bad_alloc &) {
        pc.oom = true;
    }
}

// FASTQ by its GENERAL grammar, what the reference's Bio.SeqIO accepts (round 4; rounds 2-3 refused it): '@' header,
// sequence lines up to the line that starts with '+', then quality lines until they hold as many characters as the
// sequence did.  Quality lines may start with '@' or '+', which is why record boundaries cannot be found by looking at
// line starts alone -- a wrapped file is parsed one record after the other, from a known record start: whole records of
// [b, e) into `pc` until `max_bases` bases are there (at least one record); returns where it stopped (a record's first
// byte, or e), nullptr on an error (pc.malformed / pc.bad / pc.oom say which).
inline size_t line_chars(const uint8_t *p, const uint8_t *stop) // characters of a line that are not white space
{
    size_t n = 0;
    for (; p < stop; ++p)
        n += kBytes.cls[*p] != kSpace;
    return n;
}

const uint8_t *parse_fastq_wrapped(const covest_reads *r, const uint8_t *b, const uint8_t *e, Piece &pc, int64_t max_bases)
{
    try {
        const uint8_t *p = b;
        auto next_line = [&](const uint8_t *&stop) { // [p, stop) = the line at p; p moves behind it
            const uint8_t *line = p;
            const uint8_t *nl = static_cast<const uint8_t *>(std::memchr(p, '\n', (size_t)(e - p)));
            stop = nl ? nl : e;
            p = nl ? nl + 1 : e;
            return line;
        };
        while (p < e) {
            const uint8_t *rec = p, *stop;
            const uint8_t *line = next_line(stop);
            if (stop == line || (stop == line + 1 && *line == '\r'))
                continue; // a blank line between records
            if (*line != '@') {
                pc.malformed = line;
                return nullptr;
            }
            const int64_t start = (int64_t)pc.bases.n;
            size_t seq_chars = 0;
            bool plus = false;
            while (p < e) {
                line = next_line(stop);
                if (line < stop && *line == '+') {
                    plus = true;
                    break;
                }
                seq_chars += line_chars(line, stop);
                if (!put_line(r, pc, line, stop))
                    return nullptr;
            }
            if (!plus) { // the file ends inside the sequence
                pc.malformed = rec;
                return nullptr;
            }
            size_t qual_chars = 0;
            while (qual_chars < seq_chars && p < e) {
                line = next_line(stop);
                qual_chars += line_chars(line, stop);
            }
            if (qual_chars != seq_chars) { // shorter (the file ends) or longer (a line too many) than the sequence
                pc.malformed = rec;
                return nullptr;
            }
            pc.lens.push_back((int64_t)pc.bases.n - start);
            if ((int64_t)pc.bases.n >= max_bases)
                break;
        }
        return p;
    } catch (const std::bad_alloc &) {
        pc.oom = true;
        return nullptr;
    }
}

// Does the file look wrapped?  The first records by the general grammar: one whose sequence takes more than one line says
// yes.  (A file that starts with 4-line records and wraps later is caught when the strict parser meets the first such
// record: covest_reads_next then goes over to the general grammar from that batch on.)
bool fastq_looks_wrapped(const covest_reads *r)
{
    const uint8_t *p = r->map, *e = r->map + std::min<size_t>(r->size, (size_t)1 << 20);
    int records = 0;
    while (p < e && records < 256) {
        const uint8_t *nl = static_cast<const uint8_t *>(std::memchr(p, '\n', (size_t)(e - p)));
        const uint8_t *stop = nl ? nl : e;
        if (stop == p || (stop == p + 1 && *p == '\r')) {
            p = nl ? nl + 1 : e;
            continue;
        }
        if (*p != '@')
            return false; // (not a record start: let the strict parser report it)
        p = nl ? nl + 1 : e;
        int seq_lines = 0;
        size_t seq_chars = 0;
        bool plus = false;
        while (p < e) {
            nl = static_cast<const uint8_t *>(std::memchr(p, '\n', (size_t)(e - p)));
            stop = nl ? nl : e;
            const uint8_t *line = p;
            p = nl ? nl + 1 : e;
            if (line < stop && *line == '+') {
                plus = true;
                break;
            }
            ++seq_lines;
            seq_chars += line_chars(line, stop);
        }
        if (!plus)
            return false;
        if (seq_lines > 1)
            return true;
        size_t qual_chars = 0;
        int qual_lines = 0;
        while (qual_chars < seq_chars && p < e) {
            nl = static_cast<const uint8_t *>(std::memchr(p, '\n', (size_t)(e - p)));
            stop = nl ? nl : e;
            qual_chars += line_chars(p, stop);
            p = nl ? nl + 1 : e;
            ++qual_lines;
        }
        if (qual_lines > 1)
            return true;
        ++records;
    }
    return false;
}

// The first byte of the first record that starts at or after p (the end of the file if there is none).
const uint8_t *next_record(const covest_reads *r, const uint8_t *p)
{
    const uint8_t *end = r->map + r->size;
    if (p <= r->map)
        return r->map;
    const uint8_t mark = r->fastq ? '@' : '>';
    --p; // (a record may start exactly at p: look for the newline in front of it)
    while (p < end) {
        const uint8_t *nl = static_cast<const uint8_t *>(std::memchr(p, '\n', (size_t)(end - p)));
        if (!nl || nl + 1 >= end)
            return end;
        const uint8_t *c = nl + 1;
        if (*c == mark) {
            if (!r->fastq)
                return c;
            // FASTQ: '@' also opens quality lines.  A header is followed by the sequence line and then by '+'; a
            // quality line that starts with '@' is followed by the next header and ITS sequence line, never a '+'.
            const uint8_t *l2 = static_cast<const uint8_t *>(std::memchr(c, '\n', (size_t)(end - c)));
            const uint8_t *l3 = l2 ? static_cast<const uint8_t *>(std::memchr(l2 + 1, '\n', (size_t)(end - l2 - 1))) : nullptr;
            if (l3 && l3 + 1 < end && l3[1] == '+')
                return c;
        }
        p = c;
    }
    return end;
}

} // namespace

extern "C" {

int covest_reads_open(const char *path, int32_t n_strategy, uint64_t seed, covest_reads **out)
{
    if (!path || !out)
        return covest::set_error(COVEST_E_INVALID, "covest_reads_open: null argument");
    if (n_strategy < 0 || n_strategy > 2)
        return covest::set_error(COVEST_E_INVALID, "covest_reads_open: invalid N strategy (0 IGNORE, 1 SINGLE, 2 RANDOM)");
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0)
        return covest::set_error(COVEST_E_INVALID, std::string("covest_reads_open: cannot open ") + path);
    struct stat st;
    if (::fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) {
        ::close(fd);
        return covest::set_error(COVEST_E_INVALID, std::string("covest_reads_open: not a regular file: ") + path);
    }
    covest_reads *r = new (std::nothrow) covest_reads;
    if (!r) {
        ::close(fd);
        return covest::set_error(COVEST_E_NOMEM, "covest_reads_open: out of host memory");
    }
    r->fd = fd;
    r->size = (size_t)st.st_size;
    if (r->size) {
        void *m = ::mmap(nullptr, r->size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) {
            delete r;
            return covest::set_error(COVEST_E_INVALID, std::string("covest_reads_open: cannot map ") + path);
        }
        r->map = static_cast<const uint8_t *>(m);
        (void)::madvise(m, r->size, MADV_SEQUENTIAL);
    }
    const char *dot = std::strrchr(path, '.');
    const char *slash = std::strrchr(path, '/');
    if (dot && (!slash || dot > slash))
        r->fastq = std::strcmp(dot, ".fq") == 0 || std::strcmp(dot, ".fastq") == 0;
    r->n_strategy = n_strategy;
    r->seed = seed;
    r->wrapped = r->fastq && r->size && fastq_looks_wrapped(r);
    // threads: COVEST_READER_THREADS, or what the machine offers, 16 at most
    unsigned hw = std::thread::hardware_concurrency();
    int nt = hw ? (int)std::min(hw, 16u) : 4;
    if (const char *e = std::getenv("COVEST_READER_THREADS"))
        nt = std::max(1, std::atoi(e));
    r->n_threads = nt;
    // page-locked batches when the process has a HIP device (COVEST_READER_PINNED=0: never)
    int n_dev = 0;
    const char *pin = std::getenv("COVEST_READER_PINNED");
    r->want_pinned = !(pin && std::atoi(pin) == 0) && hipGetDeviceCount(&n_dev) == hipSuccess && n_dev > 0;
    (void)hipGetLastError();
    *out = r;
    return COVEST_OK;
}

void covest_reads_close(covest_reads *r) { delete r; }

int covest_reads_next(covest_reads *r, int64_t max_bases, const uint8_t **bases, const int64_t **offsets,
                      int64_t *n_reads)
{
    if (!r || !bases || !offsets || !n_reads)
        return covest::set_error(COVEST_E_INVALID, "covest_reads_next: null argument");
    if (max_bases < 1)
        max_bases = 1;
    Batch &out = r->batch[r->cur];
    r->cur ^= 1;
    out.offsets.assign(1, 0);
    *bases = reinterpret_cast<const uint8_t *>("");
    *offsets = out.offsets.data();
    *n_reads = 0;
    if (r->pos >= r->size)
        return COVEST_OK;
    // the span: about max_bases bases' worth of file (headers, line ends and -- FASTQ -- qualities on top), up to
    // the next record boundary; one record at least
    const uint8_t *begin = r->map + r->pos, *end = r->map + r->size;
    if (r->wrapped) { // FASTQ by the general grammar: one piece, whole records until max_bases are there
        if (r->pieces.empty()) {
            try {
                r->pieces.resize(1);
            } catch (const std::bad_alloc &) {
                return covest::set_error(COVEST_E_NOMEM, "covest_reads_next: out of host memory");
            }
        }
        Piece &pc = r->pieces[0];
        pc.bases.n = 0;
        pc.lens.clear();
        pc.bad = pc.malformed = nullptr;
        pc.oom = false;
        const uint8_t *stop = parse_fastq_wrapped(r, begin, end, pc, max_bases);
        if (!stop) {
            if (pc.oom)
                return covest::set_error(COVEST_E_NOMEM, "covest_reads_next: out of host memory");
            if (r->strict_malformed) // neither grammar takes the record the 4-line parser stopped at: name THAT line
                return covest::set_error(COVEST_E_INVALID, "covest_reads_next: malformed FASTQ record at byte " +
                                                              std::to_string((long long)(r->strict_malformed - r->map)) +
                                                              " (@id, sequence, +, quality -- or wrapped: sequence lines, "
                                                              "+, as many quality characters)");
            if (pc.malformed && r->strict_bad) // (the 4-line parser's complaint stands: the general grammar has no reading either)
                return covest::set_error(COVEST_E_INVALID, std::string("covest_reads_next: base '") + (char)*r->strict_bad +
                                                              "' outside acgtn (single_hash raises KeyError)");
            if (pc.malformed)
                return covest::set_error(COVEST_E_INVALID, "covest_reads_next: malformed FASTQ record at byte " +
                                                              std::to_string((long long)(pc.malformed - r->map)) +
                                                              " (@id, sequence lines, +, as many quality characters)");
            return covest::set_error(COVEST_E_INVALID, std::string("covest_reads_next: base '") + (char)*pc.bad +
                                                          "' outside acgtn (single_hash raises KeyError)");
        }
        if (!out.reserve(std::max<size_t>(pc.bases.n, 1), r->want_pinned))
            return covest::set_error(COVEST_E_NOMEM, "covest_reads_next: out of host memory");
        try {
            out.offsets.resize(pc.lens.size() + 1);
        } catch (const std::bad_alloc &) {
            return covest::set_error(COVEST_E_NOMEM, "covest_reads_next: out of host memory");
        }
        if (pc.bases.n)
            std::memcpy(out.bases, pc.bases.p, pc.bases.n);
        int64_t at = 0;
        for (size_t k = 0; k < pc.lens.size(); ++k) {
            at += pc.lens[k];
            out.offsets[k + 1] = at;
        }
        r->pos = (size_t)(stop - r->map);
        if (r->strict_malformed && stop > r->strict_malformed)
            r->strict_malformed = nullptr; // (the general grammar took what the 4-line parser could not)
        if (r->strict_bad && stop > r->strict_bad)
            r->strict_bad = nullptr;
        *bases = out.bases;
        *offsets = out.offsets.data();
        *n_reads = (int64_t)pc.lens.size();
        return COVEST_OK;
    }
    const double per_base = r->fastq ? 2.1 : 1.08;
    const size_t want = (size_t)std::min<double>((double)(end - begin), (double)max_bases * per_base + 64.0);
    const uint8_t *stop = next_record(r, begin + std::max<size_t>(want, 1));
    if (stop <= begin)
        stop = end;
    // one piece per thread, cut at record boundaries (small spans: one piece)
    const size_t span = (size_t)(stop - begin);
    int n_pieces = (int)std::min<size_t>((size_t)r->n_threads, std::max<size_t>(1, span >> 20));
    std::vector<const uint8_t *> cut((size_t)n_pieces + 1);
    cut[0] = begin;
    cut[(size_t)n_pieces] = stop;
    for (int i = 1; i < n_pieces; ++i) {
        const uint8_t *c = next_record(r, begin + span / (size_t)n_pieces * (size_t)i);
        cut[(size_t)i] = std::min(std::max(c, cut[(size_t)i - 1]), stop);
    }
    std::vector<Piece> &pieces = r->pieces;
    try {
        if (pieces.size() < (size_t)n_pieces)
            pieces.resize((size_t)n_pieces);
    } catch (const std::bad_alloc &) {
        return covest::set_error(COVEST_E_NOMEM, "covest_reads_next: out of host memory");
    }
    for (int i = 0; i < n_pieces; ++i) {
        pieces[(size_t)i].bases.n = 0;
        pieces[(size_t)i].lens.clear();
        pieces[(size_t)i].bad = nullptr;
        pieces[(size_t)i].malformed = nullptr;
        pieces[(size_t)i].oom = false;
    }
    if (n_pieces == 1) {
        parse_piece(r, cut[0], cut[1], pieces[0]);
    } else {
        std::vector<std::thread> workers;
        for (int i = 0; i < n_pieces; ++i)
            workers.emplace_back(parse_piece, r, cut[(size_t)i], cut[(size_t)i + 1], std::ref(pieces[(size_t)i]));
        for (std::thread &w : workers)
            w.join();
    }
    size_t total = 0, n_rec = 0;
    for (int i = 0; i < n_pieces; ++i) {
        const Piece &pc = pieces[(size_t)i];
        if (pc.oom)
            return covest::set_error(COVEST_E_NOMEM, "covest_reads_next: out of host memory");
        if (pc.malformed) { // not a 4-line record: a file that wraps its lines from here on?  The general grammar decides
            r->wrapped = true;
            r->strict_malformed = pc.malformed;
            r->cur ^= 1; // (this call's batch buffer again)
            return covest_reads_next(r, max_bases, bases, offsets, n_reads);
        }
        // A "bad base" of the 4-line parser in a FASTQ file may be a QUALITY character: where only the quality wraps and
        // its second line starts with '@' -- behind more plain records than fastq_looks_wrapped reads -- the parser takes
        // that line for a header and the next '@id' line for a sequence (400 plain records, then
        // "@w / ACGTACGT / + / IIII / @III / @x ..." is a file Bio.SeqIO reads).  The general grammar decides from this
        // batch's start; the complaint is kept for the case that it has no reading either.
        if (pc.bad && r->fastq) {
            r->wrapped = true;
            r->strict_bad = pc.bad;
            r->cur ^= 1; // (this call's batch buffer again)
            return covest_reads_next(r, max_bases, bases, offsets, n_reads);
        }
        if (pc.bad) // (pieces are in file order: the first one reported is the first in the file)
            return covest::set_error(COVEST_E_INVALID, std::string("covest_reads_next: base '") + (char)*pc.bad +
                                                          "' outside acgtn (single_hash raises KeyError)");
        total += pc.bases.n;
        n_rec += pc.lens.size();
    }
    if (!out.reserve(std::max<size_t>(total, 1), r->want_pinned))
        return covest::set_error(COVEST_E_NOMEM, "covest_reads_next: out of host memory");
    try {
        out.offsets.resize(n_rec + 1);
    } catch (const std::bad_alloc &) {
        return covest::set_error(COVEST_E_NOMEM, "covest_reads_next: out of host memory");
    }
    // the pieces side by side: bases copied (in parallel), lengths turned into offsets
    std::vector<size_t> base_at((size_t)n_pieces + 1, 0), rec_at((size_t)n_pieces + 1, 0);
    for (int i = 0; i < n_pieces; ++i) {
        base_at[(size_t)i + 1] = base_at[(size_t)i] + pieces[(size_t)i].bases.n;
        rec_at[(size_t)i + 1] = rec_at[(size_t)i] + pieces[(size_t)i].lens.size();
    }
    auto place = [&](int i) {
        const Piece &pc = pieces[(size_t)i];
        if (pc.bases.n)
            std::memcpy(out.bases + base_at[(size_t)i], pc.bases.p, pc.bases.n);
        int64_t at = (int64_t)base_at[(size_t)i];
        int64_t *o = out.offsets.data() + rec_at[(size_t)i];
        for (size_t k = 0; k < pc.lens.size(); ++k) {
            at += pc.lens[k];
            o[k + 1] = at;
        }
    };
    if (n_pieces == 1) {
        place(0);
    } else {
        std::vector<std::thread> workers;
        for (int i = 0; i < n_pieces; ++i)
            workers.emplace_back(place, i);
        for (std::thread &w : workers)
            w.join();
    }
    r->pos = (size_t)(stop - r->map);
    *bases = out.bases;
    *offsets = out.offsets.data();
    *n_reads = (int64_t)n_rec;
    if (n_rec == 0 && r->pos < r->size) // a span without a single record (text before the first header): go on
        return covest_reads_next(r, max_bases, bases, offsets, n_reads);
    return COVEST_OK;
}

int64_t covest_reads_bytes(const covest_reads *r) { return r ? (int64_t)r->pos : 0; }

} // extern "C"
