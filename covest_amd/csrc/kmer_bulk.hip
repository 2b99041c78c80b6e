// kmer_bulk.hip -- K-kmer, partitioned: the whole of bin/kmer_hist.py's main loop (:77-89: compute_counts over every
// read, then compute_histogram) for reads resident in HBM, WITHOUT a table in HBM.
//
// K-kmer's table (kmer_count.hip) pays one scattered memory-side operation pair per k-mer OCCURRENCE: 2e10 a
// second, whatever the table's size (tools/microbench_atomics.hip).  Here an occurrence costs LDS work only:
//
//   pass 1  (kmer_scatter_*)   every window's MINIMIZER -- the m-mer of smallest hash among the k - m + 1 it holds,
//           canonical when the keys are -- names a BUCKET, a function of the k-mer alone: every occurrence of a key
//           lands in the same bucket.  Consecutive windows of a read mostly share their minimizer, so a wave cuts its 64
//           windows into runs of equal bucket ("super-k-mers") and writes ONE 16-byte record per run -- the run's
//           bases as 2-bit codes and their number -- behind the bucket's cursor: one returning atomic and one store
//           per ~6 windows, 2.6 bytes of HBM traffic per k-mer instead of 16 scattered ones.
//   pass 2  (kmer_bucket_count_kernel)   a persistent workgroup takes bucket after bucket: the records' k-mers go into
//           a hash table in LDS (64-bit compare-and-swap on the key, 32-bit add on the count), the table is swept
//           into the workgroup's count-of-counts bins (LDS too, flushed once at the end).
//
// Exact by construction -- and by a fall-back for everything that does not fit: a bucket that overflows its space in
// HBM (minimizers are not equally frequent; low-complexity reads put everything into a few buckets), a bucket whose
// distinct keys do not fit the LDS table, a read shorter than k: all of THAT bucket's k-mers (records in place and
// records in the overflow list alike -- a key must be counted in one place only) go to the open-addressing table of
// kmer_count.hip, whose histogram is added at the end.
//
// Reference restated: bin/kmer_hist.py:18-41 (codes, counts), :57-64 (count-of-counts); `canonical` as in kmer_count.hip.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "kernels.h"
#include "kmer_table.h"
#include "wave.h"

namespace covest {

namespace {

using namespace kmer;
typedef unsigned long long u64;

// The bucket of a k-mer: a hash of its minimizer.  `h`: the k-mer's code (first base in the highest bits), `rc`: its
// reverse complement's.  canonical: the m-mers count as the smaller of themselves and their reverse complements, which
// makes the function symmetric in (h, rc); otherwise the forward m-mers only.  Either way a function of the KEY.
__device__ __forceinline__ unsigned bucket_of(u64 h, u64 rc, const KmerBulk &p)
{
    const u64 mm = (1ull << (2 * p.m)) - 1ull;
    u64 best = ~0ull;
    for (int i = 0; i < p.w; ++i) {
        const u64 a = (h >> (2 * i)) & mm;                     // m-mer i of the key (counted from its end) ...
        u64 c = a;
        if (p.canonical) {
            const u64 b = (rc >> (2 * (p.w - 1 - i))) & mm;    // ... and its reverse complement
            c = a < b ? a : b;
        }
        const u64 x = (c + 1ull) * 0x9E3779B97F4A7C15ull;
        best = x < best ? x : best;
    }
    const u64 h2 = best * 0xD6E8FEB86659FD93ull; // (the minimum of w hashes is small: spread it again)
    return (unsigned)(h2 >> (64 - p.log2_buckets));
}

// Up to 32 bases of a read from seq[s] on as 2-bit codes, little-endian: base i at bits 2i.
__device__ __forceinline__ u64 pack_bases(const unsigned char *__restrict__ seq, int64_t s, int64_t len, int n)
{
    u64 le = 0;
    if (s + 32 <= len) {
        for (int j = 0; j < 8; ++j) { // (whole words: the bases beyond n are masked off below)
            unsigned w;
            __builtin_memcpy(&w, seq + s + 4 * j, 4);
            unsigned x = (w >> 1) & 0x03030303u;
            x ^= (x >> 1) & 0x01010101u;
            le |= (u64)((x * 0x01041040u) >> 24) << (8 * j);
        }
    } else {
        for (int i = 0; i < n; ++i)
            le |= (u64)base_code(seq[s + i]) << (2 * i);
    }
    return n < 32 ? le & ((1ull << (2 * n)) - 1ull) : le;
}

__device__ __forceinline__ void append_record(unsigned b, u64 code, int n_bases, const KmerBulk &p)
{
    const unsigned pos = atomicAdd(&p.cursor[b], 1u);
    if (pos < p.cap) {
        p.recs[(u64)b * p.cap + pos] = make_ulonglong2(code, (u64)n_bases);
    } else { // the bucket is full: the record goes to the list, and the WHOLE bucket to the table later (pass 2)
        const u64 at = atomicAdd(p.ovf_count, 1ull);
        if (at < p.overflow_cap)
            p.overflow[at] = make_ulonglong2(code, (u64)n_bases);
    }
}

// One wave, one window per lane (`valid`: the lane has one): runs of consecutive windows of one read with one bucket,
// cut into pieces of at most p.max_run windows, each piece a record behind its bucket's cursor.
__device__ __forceinline__ void scatter_wave(const unsigned char *__restrict__ seq, int64_t s, int64_t len, int64_t read_id,
                                             bool valid, const KmerBulk &p)
{
    const int lane = threadIdx.x & (kWave - 1);
    unsigned b = 0;
    if (valid) {
        u64 h, rc;
        window_codes(seq, s, len, p.k, h, rc);
        b = bucket_of(h, rc, p);
    }
    const unsigned prev_b = __shfl_up(b, 1, kWave);
    const int64_t prev_r = __shfl_up(read_id, 1, kWave);
    const bool prev_valid = __shfl_up((int)valid, 1, kWave) != 0;
    const bool head = valid && (lane == 0 || !prev_valid || prev_r != read_id || prev_b != b);
    const u64 headmask = __ballot(head);
    const u64 validmask = __ballot(valid);
    if (!valid)
        return;
    // this lane's run starts at the highest head at or below it; pieces of max_run windows
    const u64 upto = headmask & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull));
    const int hp = 63 - __clzll((long long)upto);
    const bool piece_head = ((lane - hp) % p.max_run) == 0;
    if (!piece_head)
        return;
    // the piece ends before the next head, the next invalid lane, its max_run-th window, or the wave's end
    const u64 above = lane == 63 ? 0ull : (~0ull << (lane + 1));
    const u64 stop = (headmask | ~validmask) & above;
    int run = (stop ? __ffsll((long long)stop) - 1 : 64) - lane;
    run = min(run, p.max_run);
    const int n_bases = run + p.k - 1;
    append_record(b, pack_bases(seq, s, len, n_bases), n_bases, p);
}

// Reads of one length: the windows of all reads numbered through, 64 consecutive ones per wave (as kmer_count.hip).
__global__ __launch_bounds__(256) void kmer_scatter_fixed_kernel(const unsigned char *__restrict__ bases, int64_t n_reads,
                                                                 int64_t len, const KmerBulk p)
{
    const int64_t n_windows = len - p.k + 1;
    const int64_t total = n_reads * n_windows;
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = w < total;
    const int64_t r = valid ? w / n_windows : 0;
    scatter_wave(bases + r * len, valid ? w - r * n_windows : 0, len, r, valid, p);
}

// One wave per read (reads of any length).
__global__ __launch_bounds__(256) void kmer_scatter_kernel(const unsigned char *__restrict__ bases,
                                                           const int64_t *__restrict__ offsets, int64_t n_reads,
                                                           const KmerBulk p, const KmerTable t, int *overflow)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t r = (int64_t)blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave;
    if (r >= n_reads)
        return;
    const int64_t p0 = offsets[r];
    const int64_t len = offsets[r + 1] - p0;
    const unsigned char *seq = bases + p0;
    if (len < p.k) {
        // hash_kmer of what there is (bin/kmer_hist.py:36-37): an integer below 4^len -- the very key of the k-mer
        // "a" * (k - len) + read, which a window elsewhere may spell out (a read of k - 1 bases behind an 'a' in the
        // genome: one in four).  So it is counted AS that k-mer, through its bucket: a key must live in one place.
        if (lane == 0) {
            u64 le = 0;
            for (int i = 0; i < (int)len; ++i)
                le |= (u64)base_code(seq[i]) << (2 * (p.k - (int)len + i));
            u64 h, rc;
            codes_from_le(le, p.k, h, rc);
            append_record(bucket_of(h, rc, p), le, p.k, p);
        }
        return;
    }
    const int64_t n_windows = len - p.k + 1;
    for (int64_t s0 = 0; s0 < n_windows; s0 += kWave) { // wave-uniform trip count
        const int64_t s = s0 + lane;
        scatter_wave(seq, s < n_windows ? s : 0, len, r, s < n_windows, p);
    }
}

// Every k-mer of a record into the table in HBM.
__device__ __forceinline__ void record_to_table(ulonglong2 rec, const KmerBulk &p, const KmerTable &t, int *overflow)
{
    const int n_k = (int)rec.y - p.k + 1;
    const u64 kmask = p.k < 32 ? (1ull << (2 * p.k)) - 1ull : ~0ull;
    for (int j = 0; j < n_k; ++j) {
        u64 h, rc;
        codes_from_le((rec.x >> (2 * j)) & kmask, p.k, h, rc);
        if (p.canonical && rc < h) {
            const u64 x = h;
            h = rc;
            rc = x;
        }
        table_add(t, h, rc, 1ull, overflow);
    }
}

constexpr int kLdsSlots = 4096;      // LDS hash table of a bucket: 32 KB of keys + 16 KB of counts
constexpr int kLdsHistBins = 4096;   // count-of-counts bins kept in LDS per workgroup
constexpr u64 kLdsEmpty = ~0ull;

// Pass 2.  stats: [0] max count, [1] distinct keys, [2] entries of `big` (counts >= hist_len), [3] buckets sent to the
// table.  hist: dense count-of-counts for counts < hist_len.
__global__ __launch_bounds__(256) void kmer_bucket_count_kernel(const KmerBulk p, const KmerTable t, int *overflow,
                                                                u64 *__restrict__ hist, u64 hist_len,
                                                                u64 *__restrict__ stats, u64 *__restrict__ big, u64 big_cap)
{
    __shared__ u64 keys[kLdsSlots];
    __shared__ unsigned cnts[kLdsSlots];
    __shared__ unsigned bins[kLdsHistBins];
    __shared__ unsigned n_kmers_s, failed_s;
    const int tid = threadIdx.x;
    for (int i = tid; i < kLdsHistBins; i += blockDim.x)
        bins[i] = 0u;
    u64 distinct = 0, mx = 0, to_table = 0;
    const unsigned n_buckets = 1u << p.log2_buckets;
    const u64 kmask = p.k < 32 ? (1ull << (2 * p.k)) - 1ull : ~0ull;
    for (unsigned b = blockIdx.x; b < n_buckets; b += gridDim.x) {
        const unsigned filled = p.cursor[b];
        if (filled == 0)
            continue; // (workgroup-uniform)
        const unsigned n = min(filled, p.cap);
        const ulonglong2 *recs = p.recs + (u64)b * p.cap;
        bool fall_back = filled > p.cap; // records of this bucket sit in the overflow list too: everything to the table
        unsigned slots = 0;
        if (!fall_back) {
            // the bucket's k-mers, to size the table: a power of two >= twice their number (distinct keys are fewer)
            if (tid == 0) {
                n_kmers_s = 0u;
                failed_s = 0u;
            }
            __syncthreads();
            unsigned mine = 0;
            for (unsigned i = tid; i < n; i += blockDim.x)
                mine += (unsigned)recs[i].y - (unsigned)p.k + 1u;
            for (int off = 32; off >= 1; off >>= 1)
                mine += __shfl_xor(mine, off, kWave);
            if ((tid & (kWave - 1)) == 0)
                atomicAdd(&n_kmers_s, mine);
            __syncthreads();
            const unsigned n_kmers = n_kmers_s;
            slots = 256;
            while (slots < 2u * n_kmers && slots < (unsigned)kLdsSlots)
                slots <<= 1;
            for (unsigned i = tid; i < slots; i += blockDim.x) {
                keys[i] = kLdsEmpty;
                cnts[i] = 0u;
            }
            __syncthreads();
            const unsigned smask = slots - 1u;
            for (unsigned i = tid; i < n; i += blockDim.x) {
                const ulonglong2 rec = recs[i];
                const int n_k = (int)rec.y - p.k + 1;
                for (int j = 0; j < n_k; ++j) {
                    u64 h, rc;
                    codes_from_le((rec.x >> (2 * j)) & kmask, p.k, h, rc);
                    const u64 key = (p.canonical && rc < h) ? rc : h;
                    unsigned at = (unsigned)((key * 0x9E3779B97F4A7C15ull) >> 40) & smask;
                    bool done = false;
                    for (unsigned probe = 0; probe < slots; ++probe) {
                        u64 cur = keys[at];
                        if (cur == kLdsEmpty)
                            cur = atomicCAS(&keys[at], kLdsEmpty, key);
                        if (cur == kLdsEmpty || cur == key) {
                            atomicAdd(&cnts[at], 1u);
                            done = true;
                            break;
                        }
                        at = (at + 1u) & smask;
                    }
                    if (!done)
                        failed_s = 1u; // more distinct keys than slots: the bucket goes to the table instead
                }
            }
            __syncthreads();
            fall_back = failed_s != 0u;
            if (!fall_back) {
                for (unsigned i = tid; i < slots; i += blockDim.x)
                    if (keys[i] != kLdsEmpty) {
                        const u64 c = cnts[i];
                        ++distinct;
                        mx = c > mx ? c : mx;
                        if (c < (u64)kLdsHistBins) {
                            atomicAdd(&bins[c], 1u);
                        } else if (c < hist_len) {
                            atomicAdd(&hist[c], 1ull);
                        } else {
                            const u64 at = atomicAdd(&stats[2], 1ull);
                            if (at < big_cap)
                                big[at] = c;
                        }
                    }
            }
            __syncthreads(); // (the table is cleared again for the next bucket)
        }
        if (fall_back) {
            for (unsigned i = tid; i < n; i += blockDim.x)
                record_to_table(recs[i], p, t, overflow);
            if (tid == 0)
                ++to_table;
        }
    }
    __syncthreads();
    for (int i = tid; i < kLdsHistBins; i += blockDim.x)
        if (bins[i] != 0u && (u64)i < hist_len)
            atomicAdd(&hist[i], (u64)bins[i]);
    for (int off = 32; off >= 1; off >>= 1) {
        distinct += __shfl_xor(distinct, off, kWave);
        const u64 o = __shfl_xor(mx, off, kWave);
        mx = o > mx ? o : mx;
    }
    if ((tid & (kWave - 1)) == 0) {
        atomicMax(&stats[0], mx);
        atomicAdd(&stats[1], distinct);
    }
    if (tid == 0 && to_table)
        atomicAdd(&stats[3], to_table);
}

// The records that found their bucket full: their k-mers into the table (their buckets' other records follow in pass 2).
__global__ __launch_bounds__(256) void kmer_overflow_to_table_kernel(const KmerBulk p, u64 n, const KmerTable t, int *overflow)
{
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x)
        record_to_table(p.overflow[i], p, t, overflow);
}

// k-mer occurrences the table will have to take (an upper bound: max_run per record): the records of the buckets that
// overflowed, in place and in the list.  out[0]
__global__ __launch_bounds__(256) void kmer_fallback_bound_kernel(const KmerBulk p, u64 *out)
{
    const unsigned n_buckets = 1u << p.log2_buckets;
    u64 mine = 0;
    for (unsigned b = blockIdx.x * blockDim.x + threadIdx.x; b < n_buckets; b += gridDim.x * blockDim.x) {
        const unsigned filled = p.cursor[b];
        if (filled > p.cap)
            mine += (u64)filled * (u64)p.max_run;
    }
    for (int off = 32; off >= 1; off >>= 1)
        mine += __shfl_xor(mine, off, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0 && mine)
        atomicAdd(out, mine);
}

} // namespace

hipError_t launch_kmer_scatter(const unsigned char *bases, const int64_t *offsets, int64_t n_reads, int64_t fixed_len,
                               const KmerBulk &p, const KmerTable &t, int *overflow, hipStream_t stream)
{
    if (n_reads <= 0)
        return hipSuccess;
    if (!offsets) { // (the host sends reads shorter than k to the table path)
        const int64_t n_windows = fixed_len - p.k + 1;
        const int64_t reads_per_launch = std::max<int64_t>(1, (((int64_t)1 << 31) - 256) / n_windows);
        for (int64_t first = 0; first < n_reads; first += reads_per_launch) {
            const int64_t n = std::min(n_reads - first, reads_per_launch);
            const int64_t total = n * n_windows;
            hipLaunchKernelGGL(kmer_scatter_fixed_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream,
                               bases + first * fixed_len, n, fixed_len, p);
        }
        return hipGetLastError();
    }
    const int reads_per_block = 4;
    const int64_t reads_per_launch = (int64_t)reads_per_block << 23;
    for (int64_t first = 0; first < n_reads; first += reads_per_launch) {
        const int64_t n = n_reads - first < reads_per_launch ? n_reads - first : reads_per_launch;
        const dim3 grid((unsigned)((n + reads_per_block - 1) / reads_per_block));
        hipLaunchKernelGGL(kmer_scatter_kernel, grid, dim3(reads_per_block * kWave), 0, stream, bases, offsets + first, n, p, t,
                           overflow);
    }
    return hipGetLastError();
}

hipError_t launch_kmer_fallback_bound(const KmerBulk &p, unsigned long long *out, hipStream_t stream)
{
    hipLaunchKernelGGL(kmer_fallback_bound_kernel, dim3(1024), dim3(256), 0, stream, p, out);
    return hipGetLastError();
}

hipError_t launch_kmer_overflow_to_table(const KmerBulk &p, unsigned long long n, const KmerTable &t, int *overflow,
                                         hipStream_t stream)
{
    if (n == 0)
        return hipSuccess;
    const unsigned long long blocks = std::min<unsigned long long>((n + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(kmer_overflow_to_table_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, p, n, t, overflow);
    return hipGetLastError();
}

hipError_t launch_kmer_bucket_count(const KmerBulk &p, const KmerTable &t, int *overflow, unsigned long long *hist,
                                    unsigned long long hist_len, unsigned long long *stats, unsigned long long *big,
                                    unsigned long long big_cap, int n_workgroups, hipStream_t stream)
{
    hipLaunchKernelGGL(kmer_bucket_count_kernel, dim3((unsigned)n_workgroups), dim3(256), 0, stream, p, t, overflow, hist, hist_len,
                       stats, big, big_cap);
    return hipGetLastError();
}

} // namespace covest
