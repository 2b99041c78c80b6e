// kmer_bulk.hip -- K-kmer, partitioned: the whole of bin/kmer_hist.py's main loop (:77-89: compute_counts over every
// read, then compute_histogram) for reads resident in HBM, WITHOUT a table in HBM.
//
// K-kmer's table (kmer_count.hip) pays one scattered memory-side operation pair per k-mer OCCURRENCE: 2e10 a
// second, whatever the table's size (tools/microbench_atomics.hip).  Here an occurrence costs LDS work only:
//
//   pass 0  (kmer_tile_kernel<COUNT>, kmer_caps_*)   the records pass 1 will write, counted per bucket on a SAMPLE of
//           the reads (1 block of tiles in `sample`; everything when the input is small), turned into room per bucket
//           (the estimate plus three standard deviations) and, by a prefix sum, into the buckets' places in ONE array:
//           minimizers are far from equally frequent, a fixed room per bucket would be wasted on most and short for some.
//   pass 1  (kmer_tile_kernel<SCATTER>)   every window's MINIMIZER -- the m-mer of smallest hash among the k - m + 1
//           it holds, canonical when the keys are -- names a BUCKET, a function of the k-mer alone: every occurrence
//           of a key lands in the same bucket.  Consecutive windows of a read mostly share their minimizer: the runs
//           of equal bucket ("super-k-mers") become ONE 16-byte record each -- the run's bases as 2-bit codes and their
//           number -- behind the bucket's cursor: one returning atomic and one store per ~5 windows, 6 bytes of HBM
//           traffic per k-mer instead of 16 scattered ones.  A workgroup works through tiles of 256 consecutive BYTES
//           of the read array: every thread packs the 16 bases from its byte on (one unaligned 16-byte load), hashes the
//           m-mer that starts there ONCE (LDS), and the window's minimizer is the smallest of w neighbouring entries.
//   pass 2  (kmer_wave_count_kernel)   one WAVE per bucket, no barrier anywhere: the records' k-mers go into the wave's
//           own hash table in LDS (1024 slots; 64-bit compare-and-swap on the key, 32-bit add on the count), the table
//           is swept into the workgroup's count-of-counts bins (LDS too, flushed once at the end).  Twelve waves a CU
//           hide each other's latency.  (kmer_bucket_count_kernel)  the few buckets with too many records or too many
//           distinct keys for that: a workgroup each, 4096 slots.
//
// Exact by construction -- and by a fall-back for everything that does not fit: a bucket that overflows its room in
// HBM (the sample misjudged it), a bucket whose distinct keys do not fit the LDS tables: all of THAT bucket's k-mers
// (records in place and records in the overflow list alike -- a key must be counted in one place only) go to the
// open-addressing table of kmer_count.hip, whose histogram is added at the end.
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

constexpr int kTile = 256;            // bytes (threads) of a tile of pass 0/1
constexpr int kTilesPerBlock = 8;     // consecutive tiles a workgroup works through (the unit of the sample)
constexpr unsigned kNoBucket = ~0u;
#ifndef KMER_TILES_AT_ONCE
#define KMER_TILES_AT_ONCE 1
#endif
// ... side by side (divides kTilesPerBlock).  Measured at 10 Gbp: 1 and 2 the same (18.5-18.9 ms a launch), 4 slower
// (20.6-21.4: 80 VGPRs) -- more atomics in flight per wave buy nothing, the memory side's rate is the bound.
constexpr int kTilesAtOnce = KMER_TILES_AT_ONCE;

// ---- the bucket of a k-mer ---------------------------------------------------------------------------------------
// m-mers as little-endian codes (base j at bits 2j).  The reverse complement of one: complement, reverse the pairs.
__device__ __forceinline__ unsigned mmer_canonical(unsigned x, int m, int canonical)
{
    if (!canonical)
        return x;
    unsigned r = __brev(~x);
    r = ((r & 0xAAAAAAAAu) >> 1) | ((r & 0x55555555u) << 1);
    r >>= 32 - 2 * m;
    return r < x ? r : x;
}

__device__ __forceinline__ unsigned mmer_hash(unsigned c)
{
    const unsigned h = (c + 1u) * 0x9E3779B1u; // (a bijection: different m-mers never tie)
    return h ^ (h >> 15);
}

__device__ __forceinline__ unsigned bucket_of_min(unsigned mn, int log2_buckets)
{
    unsigned h = mn * 0x85EBCA77u; // (the minimum of w hashes is small: spread it again)
    h ^= h >> 13;
    h *= 0xC2B2AE3Du;
    return h >> (32 - log2_buckets);
}

// The bucket of ONE window from its little-endian code (the slow way: pass 1's tiles share the m-mer hashes between
// windows).  The rc of the k-mer holds the rc's of its m-mers: with canonical m-mers the function is one of the KEY.
__device__ __forceinline__ unsigned bucket_of_le(u64 le, const KmerBulk &p)
{
    const unsigned mm = (1u << (2 * p.m)) - 1u;
    unsigned best = ~0u;
    for (int j = 0; j < p.w; ++j) {
        const unsigned hv = mmer_hash(mmer_canonical((unsigned)(le >> (2 * j)) & mm, p.m, p.canonical));
        best = hv < best ? hv : best;
    }
    return bucket_of_min(best, p.log2_buckets);
}

// four ASCII bases -> one byte of 2-bit codes (base j at bits 2j)
__device__ __forceinline__ unsigned pack4(unsigned w)
{
    unsigned x = (w >> 1) & 0x03030303u;
    x ^= (x >> 1) & 0x01010101u;
    return (x * 0x01041040u) >> 24;
}

// Up to 32 bases of a read from seq[s] on as 2-bit codes, little-endian: base i at bits 2i.
__device__ __forceinline__ u64 pack_bases(const unsigned char *__restrict__ seq, int64_t s, int64_t len, int n)
{
    u64 le = 0;
    if (s + 32 <= len) {
        for (int j = 0; j < 8; ++j) { // (whole words: the bases beyond n are masked off below)
            unsigned w;
            __builtin_memcpy(&w, seq + s + 4 * j, 4);
            le |= (u64)pack4(w) << (8 * j);
        }
    } else {
        for (int i = 0; i < n; ++i)
            le |= (u64)base_code(seq[s + i]) << (2 * i);
    }
    return n < 32 ? le & ((1ull << (2 * n)) - 1ull) : le;
}

// One record behind its bucket's cursor (COUNT: only counted).  ctl[b] = {first, end}: places in recs; fill[b]: the cursor.
template <bool COUNT>
__device__ __forceinline__ void append_record(unsigned b, u64 code, int n_bases, const KmerBulk &p)
{
    if (COUNT) {
        atomicAdd(&p.sampled[b], 1u);
        return;
    }
    // (the cursors are an array of their own, touched by atomics only: atomics on lines that plain loads of the same
    // kernel keep in the caches run at a tenth of the rate -- measured, DESIGN.md 6b.  Also measured: {cursor, end} in
    // one element with the end read by an agent-scope atomic load -- one line a record instead of two -- is SLOWER,
    // 137-141 against 102-103 ms at 10 Gbp)
    const ulonglong2 place = p.ctl[b];
    const u64 pos = place.x + (u64)atomicAdd(&p.fill[b], (KmerBulk::fill_t)1);
    if (pos < place.y) {
        p.recs[pos] = make_ulonglong2(code, (u64)n_bases);
    } else { // the bucket is full: the record goes to the list, and the WHOLE bucket to the table later (pass 2)
        const unsigned shard = blockIdx.x % kOvfShards;
        const u64 at = atomicAdd(&p.ovf_count[shard * kOvfStride], 1ull);
        if (at < p.overflow_cap)
            p.overflow[shard * p.overflow_cap + at] = make_ulonglong2(code, (u64)n_bases);
    }
}

// ---- pass 0 / 1, reads of one length ----------------------------------------------------------------------------
// The read array as one run of bytes (`len` bytes a read, nothing between them): tile t covers bytes
// [t * n_win, t * n_win + 256), its first n_win = 256 - max(16, w - 1) bytes are the window starts it answers for.
// `positions`: bytes of this launch (whole reads, < 2^32); `avail`: bytes that may be read from `bases` on.
// RAGGED: the reads are of different lengths -- offsets[r] .. offsets[r + 1] of the byte run are read r -- and a thread
// has to learn which read its byte belongs to: the tile's first read comes from a table made beforehand
// (kmer_tile_first_read_kernel), the starts of the up to 32 reads behind it are brought into LDS, and the thread
// searches those (a tile of shorter reads than that: the thread searches the offsets themselves).  `positions` and
// the tiles then count from byte `pos0` of the run, which need not be a read's start.
struct RaggedReads {
    const int64_t *offsets; // [n_reads + 1], absolute in the caller's byte run
    int64_t n_reads;
    int64_t base0;          // offsets[0]
    int64_t pos0;           // this launch's first byte, relative to base0
    const unsigned *first_read; // [tiles of this launch] read of the tile's first byte
};
constexpr int kTileReads = 32;

// largest r in [lo, n_reads) with offsets[r] - base0 <= pos (reads of no bases share their start with the next one)
__device__ __forceinline__ int64_t read_of(const int64_t *__restrict__ offsets, int64_t n_reads, int64_t base0, int64_t pos,
                                           int64_t lo)
{
    int64_t hi = n_reads; // offsets[hi] - base0 > pos (or hi == n_reads)
    while (hi - lo > 1) {
        const int64_t mid = lo + ((hi - lo) >> 1);
        if (offsets[mid] - base0 <= pos)
            lo = mid;
        else
            hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(256) void kmer_tile_first_read_kernel(RaggedReads src, unsigned n_tiles, int n_win,
                                                                   unsigned *__restrict__ first_read)
{
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n_tiles)
        first_read[t] = (unsigned)read_of(src.offsets, src.n_reads, src.base0, src.pos0 + (int64_t)t * n_win, 0);
}

template <bool COUNT, int U, bool RAGGED>
__global__ __launch_bounds__(kTile) void kmer_tile_kernel(const unsigned char *__restrict__ bases, unsigned positions,
                                                          u64 avail, unsigned len, unsigned n_tiles, const KmerBulk p,
                                                          const RaggedReads src)
{
    __shared__ int64_t roff[RAGGED ? U : 1][RAGGED ? kTileReads + 2 : 1]; // starts of the tile's reads, relative to pos0
    // U tiles side by side (a wave that waits for the places of one tile's records would keep U times as many atomics
    // in flight: see kTilesAtOnce for what that bought)
    __shared__ unsigned hs[U][kTile]; // hash of the m-mer that starts at the byte
    __shared__ unsigned cs[U][kTile]; // the 16 bases from the byte on
    __shared__ unsigned bk[U][kTile]; // bucket of the window that starts at the byte
    __shared__ u64 hmask[U][kTile / kWave], vmask[U][kTile / kWave];
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    const int n_win = kTile - max(16, p.w - 1);
    const unsigned mm = (1u << (2 * p.m)) - 1u;
    const unsigned block = COUNT ? blockIdx.x * (unsigned)p.sample : blockIdx.x;
    for (int tt = 0; tt < kTilesPerBlock; tt += U) {
        const unsigned tile0 = block * kTilesPerBlock + tt;
        if (tile0 >= n_tiles)
            break; // (workgroup-uniform)
        bool valid[U];
        unsigned b[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned at = (tile0 + u) * (unsigned)n_win + (unsigned)tid;
            unsigned c16 = 0;
            if ((u64)at + 16ull <= avail) {
                uint4 v;
                __builtin_memcpy(&v, bases + at, 16);
                c16 = pack4(v.x) | (pack4(v.y) << 8) | (pack4(v.z) << 16) | (pack4(v.w) << 24);
            } else {
                for (int j = 0; j < 16; ++j)
                    if ((u64)at + (u64)j < avail)
                        c16 |= base_code(bases[at + j]) << (2 * j);
            }
            if (!RAGGED) {
                const unsigned i = at % len; // place in its read
                const bool live = tile0 + u < n_tiles && at < positions;
                hs[u][tid] = (live && i + (unsigned)p.m <= len) ? mmer_hash(mmer_canonical(c16 & mm, p.m, p.canonical)) : ~0u;
                valid[u] = live && tid < n_win && i + (unsigned)p.k <= len;
            } else {
                // the starts of the tile's first reads (and the end of the last of them), relative to pos0
                const int64_t r0 = tile0 + u < n_tiles ? (int64_t)src.first_read[tile0 + u] : 0;
                if (tid <= kTileReads) {
                    const int64_t r = r0 + tid < src.n_reads ? r0 + tid : src.n_reads;
                    roff[u][tid] = src.offsets[r] - src.base0 - src.pos0;
                }
                __syncthreads();
                const int64_t here = (int64_t)at;
                int j = 0; // the read of this byte: the last of the tile's reads that starts at or before it
                for (int step = kTileReads / 2; step >= 1; step >>= 1)
                    if (roff[u][j + step] <= here)
                        j += step;
                int64_t start = roff[u][j], end = roff[u][j + 1];
                if (j == kTileReads - 1 && end <= here && r0 + kTileReads < src.n_reads) { // more reads than the table holds
                    const int64_t r = read_of(src.offsets, src.n_reads, src.base0, src.pos0 + here, r0 + kTileReads - 1);
                    start = src.offsets[r] - src.base0 - src.pos0;
                    end = src.offsets[r + 1] - src.base0 - src.pos0;
                }
                // (the halo's bytes belong to reads too: only the WINDOWS are this launch's or not)
                const bool in_run = tile0 + u < n_tiles && (u64)at < avail && here < end;
                hs[u][tid] = (in_run && here + p.m <= end) ? mmer_hash(mmer_canonical(c16 & mm, p.m, p.canonical)) : ~0u;
                valid[u] = in_run && at < positions && tid < n_win && here + p.k <= end && here >= start;
            }
            cs[u][tid] = c16;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < U; ++u) {
            b[u] = kNoBucket;
            if (valid[u]) {
                unsigned mn = hs[u][tid];
                for (int j = 1; j < p.w; ++j) {
                    const unsigned o = hs[u][tid + j];
                    mn = o < mn ? o : mn;
                }
                b[u] = bucket_of_min(mn, p.log2_buckets);
            }
            bk[u][tid] = b[u];
        }
        __syncthreads();
        // a run starts where the window before is none (a read begins, the tile does) or belongs elsewhere
        bool head[U];
        u64 hm[U], vm[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            head[u] = valid[u] && (tid == 0 || bk[u][tid - 1] != b[u]);
            hm[u] = __ballot(head[u]);
            vm[u] = __ballot(valid[u]);
            if (lane == 0) {
                hmask[u][wave] = hm[u];
                vmask[u][wave] = vm[u];
            }
        }
        __syncthreads();
        // ... and ends before the next head or the first byte that starts no window (the tile's halo counts as such).
        // Pieces of at most max_run windows: 32 bases a record.  The FIRST piece of every run first, all tiles': their
        // cursors' atomics are in flight together.
        int run[U];
        u64 le[U], pos[U], end[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            run[u] = 0;
            if (head[u]) {
                const u64 stop = (hm[u] | ~vm[u]) & (lane == kWave - 1 ? 0ull : (~0ull << (lane + 1)));
                if (stop) {
                    run[u] = __ffsll((long long)stop) - 1 - lane;
                } else {
                    run[u] = kWave - lane;
                    for (int wv = wave + 1; wv < kTile / kWave; ++wv) {
                        const u64 s2 = hmask[u][wv] | ~vmask[u][wv];
                        if (s2) {
                            run[u] += __ffsll((long long)s2) - 1;
                            break;
                        }
                        run[u] += kWave;
                    }
                }
                if (COUNT) {
                    atomicAdd(&p.sampled[b[u]], (unsigned)((run[u] + p.max_run - 1) / p.max_run));
                } else {
                    const int n_bases = min(p.max_run, run[u]) + p.k - 1;
                    le[u] = (u64)cs[u][tid] | ((u64)cs[u][tid + 16] << 32);
                    if (n_bases < 32)
                        le[u] &= (1ull << (2 * n_bases)) - 1ull;
                    const ulonglong2 place = p.ctl[b[u]];
                    end[u] = place.y;
                    pos[u] = place.x + (u64)atomicAdd(&p.fill[b[u]], (KmerBulk::fill_t)1);
                }
            }
        }
        if (!COUNT) {
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (head[u]) {
                    const ulonglong2 rec = make_ulonglong2(le[u], (u64)(min(p.max_run, run[u]) + p.k - 1));
                    if (pos[u] < end[u]) {
                        p.recs[pos[u]] = rec;
                    } else { // the bucket is full: the record goes to the list, and the WHOLE bucket to the table later
                        const unsigned shard = blockIdx.x % kOvfShards;
                        const u64 o = atomicAdd(&p.ovf_count[shard * kOvfStride], 1ull);
                        if (o < p.overflow_cap)
                            p.overflow[shard * p.overflow_cap + o] = rec;
                    }
                    for (int off = p.max_run; off < run[u]; off += p.max_run) { // (a run of more than max_run windows)
                        const int a = tid + off;
                        const int n_bases = min(p.max_run, run[u] - off) + p.k - 1;
                        u64 more = (u64)cs[u][a] | ((u64)cs[u][a + 16] << 32);
                        if (n_bases < 32)
                            more &= (1ull << (2 * n_bases)) - 1ull;
                        append_record<false>(b[u], more, n_bases, p);
                    }
                }
        }
        __syncthreads(); // (the arrays are the next tiles')
    }
}

// ---- pass 0 / 1, reads of any length: the tiles above with RAGGED, and the reads SHORTER than k here ------------------
// hash_kmer of what there is (bin/kmer_hist.py:36-37): an integer below 4^len -- the very key of the k-mer
// "a" * (k - len) + read, which a window elsewhere may spell out (a read of k - 1 bases behind an 'a' in the genome:
// one in four).  So it is counted AS that k-mer, through its bucket: a key must live in one place.  A thread a read.
template <bool COUNT>
__global__ __launch_bounds__(256) void kmer_short_reads_kernel(const unsigned char *__restrict__ bases,
                                                               const int64_t *__restrict__ offsets, int64_t n_reads,
                                                               const KmerBulk p)
{
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (COUNT)
        r *= p.sample;
    if (r >= n_reads)
        return;
    const int64_t p0 = offsets[r];
    const int len = (int)min(offsets[r + 1] - p0, (int64_t)p.k);
    if (len >= p.k)
        return;
    u64 le = 0;
    for (int i = 0; i < len; ++i)
        le |= (u64)base_code(bases[p0 + i]) << (2 * (p.k - len + i));
    append_record<COUNT>(bucket_of_le(le, p), le, p.k, p);
}

// ---- pass 0: room per bucket and the buckets' places (a prefix sum over the buckets) -------------------------------
constexpr int kScanItems = 4;
constexpr int kScanBlock = 256 * kScanItems;

// sampled records of a bucket -> records it gets room for: all of them when everything was counted, else the estimate
// plus three standard deviations of it (the count in a 1-in-s sample of n records scatters by sqrt(n / s))
__device__ __forceinline__ u64 room_for(unsigned sampled, int sample)
{
    if (sample <= 1)
        return sampled;
    const float c = (float)sampled;
    return (u64)((float)sample * (c + 3.0f * sqrtf(c) + 2.0f));
}

// exclusive prefix of `v` over the 256 threads of the workgroup (lds: 4 words); `total` = the sum over all of them
__device__ __forceinline__ u64 block_exclusive_scan(u64 v, u64 *lds, u64 &total)
{
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    u64 inc = v;
    for (int off = 1; off < kWave; off <<= 1) {
        const u64 o = __shfl_up(inc, off, kWave);
        if (lane >= off)
            inc += o;
    }
    __syncthreads(); // (lds may still be read from the call before)
    if (lane == kWave - 1)
        lds[wave] = inc;
    __syncthreads();
    u64 before = 0;
    total = 0;
    for (int wv = 0; wv < 4; ++wv) {
        const u64 s = lds[wv];
        if (wv < wave)
            before += s;
        total += s;
    }
    return before + inc - v;
}

__global__ __launch_bounds__(256) void kmer_caps_partial_kernel(const KmerBulk p, u64 *__restrict__ partial)
{
    __shared__ u64 lds[4];
    const unsigned first = blockIdx.x * kScanBlock + threadIdx.x * kScanItems;
    u64 mine = 0;
    for (int j = 0; j < kScanItems; ++j)
        mine += room_for(p.sampled[first + j], p.sample);
    u64 total;
    block_exclusive_scan(mine, lds, total);
    if (threadIdx.x == 0)
        partial[blockIdx.x] = total;
}

// one workgroup: partial[] -> its exclusive prefix, in place; out[0] = the total
__global__ __launch_bounds__(256) void kmer_caps_scan_kernel(u64 *__restrict__ partial, unsigned n, u64 *__restrict__ out)
{
    __shared__ u64 lds[4];
    const unsigned per = (n + 255u) / 256u;
    const unsigned first = threadIdx.x * per;
    u64 mine = 0;
    for (unsigned j = first; j < first + per && j < n; ++j)
        mine += partial[j];
    u64 total;
    u64 run = block_exclusive_scan(mine, lds, total);
    for (unsigned j = first; j < first + per && j < n; ++j) {
        const u64 v = partial[j];
        partial[j] = run;
        run += v;
    }
    if (threadIdx.x == 0)
        out[0] = total;
}

__global__ __launch_bounds__(256) void kmer_caps_place_kernel(const KmerBulk p, const u64 *__restrict__ partial)
{
    __shared__ u64 lds[4];
    const unsigned first = blockIdx.x * kScanBlock + threadIdx.x * kScanItems;
    u64 room[kScanItems], mine = 0;
    for (int j = 0; j < kScanItems; ++j) {
        room[j] = room_for(p.sampled[first + j], p.sample);
        mine += room[j];
    }
    u64 total;
    u64 at = partial[blockIdx.x] + block_exclusive_scan(mine, lds, total);
    for (int j = 0; j < kScanItems; ++j) {
        p.ctl[first + j] = make_ulonglong2(at, at + room[j]);
        at += room[j];
    }
}

// ---- pass 2 --------------------------------------------------------------------------------------------------------
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

constexpr int kWaveSlots = 1024;     // LDS hash table of a wave: 8 KB of keys + 4 KB of counts (or half of it, see the launch)
constexpr int kWaveRecs = 768;       // a bucket with more records than this goes to a workgroup instead
constexpr int kLdsSlots = 4096;      // LDS hash table of a workgroup: 32 KB of keys + 16 KB of counts
constexpr int kLdsHistBins = 1024;   // count-of-counts bins kept in LDS per workgroup
constexpr u64 kLdsEmpty = ~0ull;

// The key a k-mer is counted under in the LDS tables: any one-to-one function of the reference's key does -- the
// smaller of the window's little-endian code and its reverse complement's.  The reverse complement of the RECORD once
// (complement, reverse the pairs), its windows are then shifts of it.
__device__ __forceinline__ u64 record_revcomp(ulonglong2 rec)
{
    if (rec.y == 0)
        return 0ull;
    u64 r = __brevll(~rec.x);
    r = ((r & 0xAAAAAAAAAAAAAAAAull) >> 1) | ((r & 0x5555555555555555ull) << 1);
    return r >> (64 - 2 * (int)rec.y);
}

__device__ __forceinline__ unsigned lds_slot(u64 key, unsigned smask)
{
    const unsigned lo = (unsigned)key, hi = (unsigned)(key >> 32);
    const unsigned x = (lo ^ (hi << 15) ^ (hi >> 5) ^ (hi << 27)) * 0x9E3779B1u;
    return (x >> 12) & smask;
}

// Every k-mer of the lane's record (none: rec.y == 0) into the table, false: the table is full.  The lanes of a wave
// go through their records side by side but each at its own pace -- ONE flat loop, a probe per turn: a lane whose probe
// hit counts and moves on to its next k-mer while its neighbours still probe (nested loops -- k-mers outside, probes
// inside -- cost three scalar instructions of mask bookkeeping per vector instruction).  (Measured and not kept, round 5:
// TWO windows of a record per turn, the even and the odd ones with a probe state each, so that a wave has two
// compare-and-swaps in flight -- pass 2 43.8 -> 55.2 ms at 10 Gbp, profiles/r05_c5_ab_two_windows_per_turn_not_kept.txt:
// the LDS applies a wave's atomics one after the other either way, and the second state costs registers and selects.)
__device__ __forceinline__ bool insert_record(ulonglong2 rec, u64 *keys, unsigned *cnts, unsigned slots, u64 kmask,
                                              const KmerBulk &p)
{
    const unsigned smask = slots - 1u;
    int left = rec.y ? (int)rec.y - p.k + 1 : 0;                 // k-mers to go (< 0: gave up); window j of the record is
    const u64 rc = p.canonical ? record_revcomp(rec) : 0ull;     // window n_k - 1 - j of its reverse complement
    u64 f = rec.x;
    u64 key = f & kmask;
    if (p.canonical) {
        const u64 r = (rc >> (2 * max(left - 1, 0))) & kmask;
        key = r < key ? r : key;
    }
    unsigned at = lds_slot(key, smask), probes = 0;
    while (__ballot(left > 0) != 0ull) {
        if (left > 0) {
            // (no look before the compare-and-swap: one LDS round trip a probe instead of two -- measured, 10 % of pass 2)
            const u64 cur = atomicCAS(&keys[at], kLdsEmpty, key);
            if (cur == kLdsEmpty || cur == key) {
                atomicAdd(&cnts[at], 1u);
                f >>= 2;
                --left;
                key = f & kmask;
                if (p.canonical) {
                    const u64 r = (rc >> (2 * max(left - 1, 0))) & kmask;
                    key = r < key ? r : key;
                }
                at = lds_slot(key, smask);
                probes = 0;
            } else {
                at = (at + 1u) & smask;
                if (++probes >= slots) // every slot holds another key
                    left = -1;
            }
        }
    }
    return left == 0;
}

// (Measured and not kept, round 3: two records a lane with both probe sequences in flight per turn and no branch inside
// it -- lanes without a k-mer left swap "empty" for "empty" -- 65 against 44 ms at 10 Gbp: the idle lanes' swaps and
// the adds of 0 are LDS traffic too, and that, not the round trip alone, is what the kernel waits for.)
struct SweepAcc {
    u64 distinct = 0, mx = 0;
};

// one slot's count into the bins
__device__ __forceinline__ void count_into_bins(u64 c, SweepAcc &acc, unsigned *bins, u64 *__restrict__ hist, u64 hist_len,
                                                u64 *__restrict__ stats, u64 *__restrict__ big, u64 big_cap)
{
    ++acc.distinct;
    acc.mx = c > acc.mx ? c : acc.mx;
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

__device__ __forceinline__ void flush_stats(SweepAcc acc, const unsigned *bins, u64 *__restrict__ hist, u64 hist_len,
                                            u64 *__restrict__ stats)
{
    for (int i = threadIdx.x; i < kLdsHistBins; i += blockDim.x)
        if (bins[i] != 0u && (u64)i < hist_len)
            atomicAdd(&hist[i], (u64)bins[i]);
    for (int off = 32; off >= 1; off >>= 1) {
        acc.distinct += __shfl_xor(acc.distinct, off, kWave);
        const u64 o = __shfl_xor(acc.mx, off, kWave);
        acc.mx = o > acc.mx ? o : acc.mx;
    }
    if ((threadIdx.x & (kWave - 1)) == 0) {
        atomicMax(&stats[0], acc.mx);
        atomicAdd(&stats[1], acc.distinct);
    }
}

// Pass 2, a wave per bucket.  stats: [0] max count, [1] distinct keys, [2] entries of `big` (counts >= hist_len), [3]
// records pass 1 sent (to the buckets and to the overflow list).
// hist: dense count-of-counts for counts < hist_len.  later[0]: buckets left to
// kmer_bucket_count_kernel, listed in later_list.
template <int SLOTS>
__global__ __launch_bounds__(256) void kmer_wave_count_kernel(const KmerBulk p, u64 *__restrict__ hist, u64 hist_len,
                                                              u64 *__restrict__ stats, u64 *__restrict__ big, u64 big_cap,
                                                              unsigned *__restrict__ later, unsigned *__restrict__ later_list)
{
    __shared__ u64 all_keys[256 / kWave][SLOTS];
    __shared__ unsigned all_cnts[256 / kWave][SLOTS];
    __shared__ unsigned bins[kLdsHistBins];
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    u64 *keys = all_keys[wave];
    unsigned *cnts = all_cnts[wave];
    for (int i = tid; i < kLdsHistBins; i += blockDim.x)
        bins[i] = 0u;
    for (int i = lane; i < SLOTS; i += kWave) {
        keys[i] = kLdsEmpty;
        cnts[i] = 0u;
    }
    __syncthreads();
    SweepAcc acc;
    u64 n_records = 0;
    const unsigned n_groups = (1u << p.log2_buckets) / kWave; // (at least 2^10 buckets)
    const unsigned n_waves = gridDim.x * (blockDim.x / kWave);
    const u64 kmask = p.k < 32 ? (1ull << (2 * p.k)) - 1ull : ~0ull;
    for (unsigned g = blockIdx.x * (blockDim.x / kWave) + wave; g < n_groups; g += n_waves) {
        // 64 buckets' cursors at once, a lane each
        const unsigned b_lane = g * kWave + lane;
        const ulonglong2 c = p.ctl[b_lane];
        const u64 first = c.x, filled = p.fill[b_lane], room = c.y - c.x;
        n_records += filled;
        const bool skip = filled > room || filled > (u64)(kWaveRecs * SLOTS / kWaveSlots);
        const u64 skipmask = __ballot(skip);
        if (skipmask) {
            unsigned at = 0;
            if (lane == 0)
                at = atomicAdd(later, (unsigned)__popcll(skipmask));
            at = __builtin_amdgcn_readfirstlane(at);
            if (skip)
                later_list[at + __popcll(skipmask & ((1ull << lane) - 1ull))] = b_lane;
        }
        u64 todo = __ballot(!skip && filled > 0);
        u64 s_next = 0;
        unsigned n_next = 0;
        int l_next = 0;
        ulonglong2 rec = make_ulonglong2(0ull, 0ull);
        if (todo) {
            l_next = __ffsll((long long)todo) - 1;
            s_next = __shfl(first, l_next, kWave);
            n_next = (unsigned)__shfl((unsigned)filled, l_next, kWave);
            if ((unsigned)lane < n_next)
                rec = p.recs[s_next + lane];
        }
        while (todo) {
            const int l = l_next;
            const u64 s_b = s_next;
            const unsigned n_b = n_next;
            todo &= todo - 1ull;
            n_next = 0;
            if (todo) {
                l_next = __ffsll((long long)todo) - 1;
                s_next = __shfl(first, l_next, kWave);
                n_next = (unsigned)__shfl((unsigned)filled, l_next, kWave);
            }
            unsigned slots = 64;
            while (slots < 8u * n_b && slots < (unsigned)SLOTS)
                slots <<= 1;
            bool ok = true;
            for (unsigned i0 = 0; i0 < n_b; i0 += kWave) {
                // the records after these -- the bucket's next 64, or the first of the next bucket -- are on their way
                // while these are counted
                const bool last = i0 + kWave >= n_b;
                const u64 from = last ? s_next : s_b + i0 + kWave;
                const unsigned have = last ? n_next : n_b - (i0 + kWave);
                ulonglong2 rec_next = make_ulonglong2(0ull, 0ull);
                if ((unsigned)lane < have)
                    rec_next = p.recs[from + lane];
                ok = insert_record(rec, keys, cnts, slots, kmask, p) && ok;
                rec = rec_next;
            }
            const bool failed = __ballot(!ok) != 0ull; // more distinct keys than slots: a workgroup takes the bucket
            for (unsigned i = lane; i < slots; i += kWave) {
                if (keys[i] != kLdsEmpty) {
                    const u64 cnt = cnts[i];
                    keys[i] = kLdsEmpty; // (the table is the next bucket's)
                    cnts[i] = 0u;
                    if (!failed)
                        count_into_bins(cnt, acc, bins, hist, hist_len, stats, big, big_cap);
                }
            }
            if (failed && lane == 0)
                later_list[atomicAdd(later, 1u)] = g * kWave + (unsigned)l;
        }
    }
    __syncthreads();
    flush_stats(acc, bins, hist, hist_len, stats);
    for (int off = 32; off >= 1; off >>= 1)
        n_records += __shfl_xor(n_records, off, kWave);
    if (lane == 0 && n_records)
        atomicAdd(&stats[3], n_records);
}

// Pass 2, a workgroup per bucket: those of later_list.  What does not fit here either is listed for the table in HBM
// (to_table[0] buckets, to_table[1] their k-mer occurrences -- the table is sized AFTER this is known --, to_table_list).
__global__ __launch_bounds__(256) void kmer_bucket_count_kernel(const KmerBulk p, u64 *__restrict__ hist, u64 hist_len,
                                                                u64 *__restrict__ stats, u64 *__restrict__ big, u64 big_cap,
                                                                const unsigned *__restrict__ later,
                                                                const unsigned *__restrict__ later_list,
                                                                u64 *__restrict__ to_table, unsigned *__restrict__ to_table_list)
{
    __shared__ u64 keys[kLdsSlots];
    __shared__ unsigned cnts[kLdsSlots];
    __shared__ unsigned bins[kLdsHistBins];
    const int tid = threadIdx.x;
    for (int i = tid; i < kLdsHistBins; i += blockDim.x)
        bins[i] = 0u;
    for (int i = tid; i < kLdsSlots; i += blockDim.x) {
        keys[i] = kLdsEmpty;
        cnts[i] = 0u;
    }
    __syncthreads();
    SweepAcc acc;
    const unsigned n_later = later[0];
    const u64 kmask = p.k < 32 ? (1ull << (2 * p.k)) - 1ull : ~0ull;
    for (unsigned at = blockIdx.x; at < n_later; at += gridDim.x) {
        const unsigned b = later_list[at];
        const ulonglong2 c = p.ctl[b];
        const u64 first = c.x, filled = p.fill[b], room = c.y - c.x;
        const u64 n = filled < room ? filled : room;
        const ulonglong2 *recs = p.recs + first;
        // records of this bucket in the overflow list too, or more distinct keys than the table holds: to the table
        // ... or so many records that one key's count could pass the 32 bits of the LDS counters (a low-complexity input
        // with 2^32 windows and more in one bucket): the table's counters are 64-bit
        bool fall_back = filled > room || filled * (u64)p.max_run >= (1ull << 32);
        u64 n_kmers = 0;
        if (!fall_back) {
            unsigned slots = 1024;
            while ((u64)slots < 8ull * n && slots < (unsigned)kLdsSlots)
                slots <<= 1;
            bool ok = true;
            for (u64 i = tid; i < n; i += blockDim.x) {
                const ulonglong2 rec = recs[i];
                const int n_k = (int)rec.y - p.k + 1;
                n_kmers += (u64)n_k;
                ok = insert_record(rec, keys, cnts, slots, kmask, p) && ok;
            }
            fall_back = __syncthreads_or(!ok) != 0;
            for (unsigned i = tid; i < slots; i += blockDim.x)
                if (keys[i] != kLdsEmpty) {
                    const u64 cnt = cnts[i];
                    keys[i] = kLdsEmpty;
                    cnts[i] = 0u;
                    if (!fall_back)
                        count_into_bins(cnt, acc, bins, hist, hist_len, stats, big, big_cap);
                }
            __syncthreads();
        }
        if (fall_back) {
            if (n_kmers == 0) // (not read yet)
                for (u64 i = tid; i < n; i += blockDim.x)
                    n_kmers += (u64)((int)recs[i].y - p.k + 1);
            for (int off = 32; off >= 1; off >>= 1)
                n_kmers += __shfl_xor(n_kmers, off, kWave);
            if ((tid & (kWave - 1)) == 0)
                atomicAdd(&to_table[1], n_kmers);
            if (tid == 0)
                to_table_list[atomicAdd(&to_table[0], 1ull)] = b;
        }
    }
    __syncthreads();
    flush_stats(acc, bins, hist, hist_len, stats);
}

// The buckets no LDS table could hold: their records' k-mers into the table in HBM.
__global__ __launch_bounds__(256) void kmer_buckets_to_table_kernel(const KmerBulk p, const KmerTable t, int *overflow,
                                                                    const u64 *__restrict__ to_table,
                                                                    const unsigned *__restrict__ to_table_list)
{
    const unsigned n_listed = (unsigned)to_table[0];
    for (unsigned at = blockIdx.x; at < n_listed; at += gridDim.x) {
        const unsigned b = to_table_list[at];
        const ulonglong2 c = p.ctl[b];
        const u64 filled = p.fill[b], room = c.y - c.x;
        const u64 n = filled < room ? filled : room;
        for (u64 i = threadIdx.x; i < n; i += blockDim.x)
            record_to_table(p.recs[c.x + i], p, t, overflow);
    }
}

// The records that found their bucket full: their k-mers into the table (their buckets' other records follow in pass 2).
__global__ __launch_bounds__(256) void kmer_overflow_to_table_kernel(const KmerBulk p, const KmerTable t, int *overflow)
{
    const unsigned shard = blockIdx.y; // (a part of the list per grid row)
    const u64 n = min(p.ovf_count[shard * kOvfStride], p.overflow_cap);
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x)
        record_to_table(p.overflow[shard * p.overflow_cap + i], p, t, overflow);
}

// What bounds pass 1, measured: returning 64-bit atomic adds at pseudo-random places of `slots` words -- a wave
// instruction's 64 lanes all on different lines, as the records of a tile are.
__global__ __launch_bounds__(256) void kmer_scatter_rate_kernel(u64 *__restrict__ words, u64 slots, int per_thread,
                                                                u64 *__restrict__ sink)
{
    u64 x = ((u64)blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull;
    u64 acc = 0;
    for (int i = 0; i < per_thread; ++i) {
        x ^= x >> 29;
        x *= 0xBF58476D1CE4E5B9ull;
        x ^= x >> 32;
        acc += atomicAdd(&words[x % slots], 1ull);
    }
    if (acc == ~0ull)
        sink[0] = acc; // (never: keeps the returned values alive)
}

// out[0] = 0 if some read is not offsets[1] - offsets[0] bases long (out comes in as 1)
__global__ __launch_bounds__(256) void kmer_one_length_kernel(const int64_t *__restrict__ offsets, int64_t n_reads,
                                                              unsigned long long *__restrict__ out)
{
    const int64_t len0 = offsets[1] - offsets[0];
    bool same = true;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_reads; r += (int64_t)gridDim.x * blockDim.x)
        same = same && offsets[r + 1] - offsets[r] == len0;
    if (__ballot(!same) != 0ull && (threadIdx.x & (kWave - 1)) == 0)
        out[0] = 0ull;
}

template <bool COUNT>
hipError_t launch_tiles(const unsigned char *bases, int64_t n_reads, int64_t len, const KmerBulk &p, hipStream_t stream)
{
    const int n_win = kTile - std::max(16, p.w - 1);
    // whole reads per launch, their bytes (and the tiles' halo) below 2^32
    const int64_t reads_per_launch = std::max<int64_t>(1, (((int64_t)1 << 31)) / len);
    for (int64_t first = 0; first < n_reads; first += reads_per_launch) {
        const int64_t n = std::min(n_reads - first, reads_per_launch);
        const uint64_t positions = (uint64_t)(n * len);
        const uint64_t avail = (uint64_t)((n_reads - first) * len);
        const uint64_t n_tiles = (positions + n_win - 1) / n_win;
        uint64_t blocks = (n_tiles + kTilesPerBlock - 1) / kTilesPerBlock;
        if (COUNT)
            blocks = (blocks + p.sample - 1) / p.sample;
        hipLaunchKernelGGL((kmer_tile_kernel<COUNT, kTilesAtOnce, false>), dim3((unsigned)blocks), dim3(kTile), 0, stream,
                           bases + first * len, (unsigned)positions, (u64)avail, (unsigned)len, (unsigned)n_tiles, p, RaggedReads{});
    }
    return hipGetLastError();
}

// Reads of any length: the byte run offsets[0] .. offsets[n_reads] in launches of whole blocks of tiles (below 2^31
// bytes each; a launch need not end with a read: a window belongs to the launch its first byte is in).  first_read:
// room for a word per tile of the largest launch (kmer_bulk_ragged_tiles).
template <bool COUNT>
hipError_t launch_ragged(const unsigned char *bases, const int64_t *offsets, int64_t n_reads, int64_t base0, int64_t total,
                         unsigned *first_read, const KmerBulk &p, hipStream_t stream)
{
    const int n_win = kTile - std::max(16, p.w - 1);
    const int64_t block_bytes = (int64_t)kTilesPerBlock * n_win;
    const int64_t per_launch = (((int64_t)1 << 31) / block_bytes) * block_bytes;
    for (int64_t pos0 = 0; pos0 < total; pos0 += per_launch) {
        const int64_t positions = std::min(total - pos0, per_launch);
        const uint64_t n_tiles = (uint64_t)((positions + n_win - 1) / n_win);
        uint64_t blocks = (n_tiles + kTilesPerBlock - 1) / kTilesPerBlock;
        RaggedReads src{offsets, n_reads, base0, pos0, first_read};
        // (both passes make the table again: a launch's table is gone once the next launch has made its own)
        hipLaunchKernelGGL(kmer_tile_first_read_kernel, dim3((unsigned)((n_tiles + 255) / 256)), dim3(256), 0, stream, src,
                           (unsigned)n_tiles, n_win, first_read);
        if (COUNT)
            blocks = (blocks + p.sample - 1) / p.sample;
        hipLaunchKernelGGL((kmer_tile_kernel<COUNT, kTilesAtOnce, true>), dim3((unsigned)blocks), dim3(kTile), 0, stream,
                           bases + base0 + pos0, (unsigned)positions, (u64)(total - pos0), 0u, (unsigned)n_tiles, p, src);
    }
    const int64_t step = COUNT ? p.sample : 1;
    const int64_t threads = (n_reads + step - 1) / step;
    const int64_t per = (int64_t)1 << 30; // threads a launch
    for (int64_t first = 0; first < threads; first += per) {
        const int64_t n = std::min(threads - first, per);
        hipLaunchKernelGGL(kmer_short_reads_kernel<COUNT>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, bases,
                           offsets + first * step, n_reads - first * step, p);
    }
    return hipGetLastError();
}

} // namespace

int kmer_bulk_block_bytes(const KmerBulk &p)
{
    return kTilesPerBlock * (kTile - std::max(16, p.w - 1));
}

int64_t kmer_bulk_ragged_tiles(const KmerBulk &p, int64_t total_bytes)
{
    const int n_win = kTile - std::max(16, p.w - 1);
    const int64_t block_bytes = (int64_t)kTilesPerBlock * n_win;
    const int64_t per_launch = (((int64_t)1 << 31) / block_bytes) * block_bytes;
    return (std::min(total_bytes, per_launch) + n_win - 1) / n_win + 1;
}

hipError_t launch_kmer_scatter(const unsigned char *bases, const int64_t *offsets, int64_t n_reads, int64_t fixed_len,
                               int64_t base0, int64_t total_bytes, unsigned *first_read, const KmerBulk &p, bool count_only,
                               hipStream_t stream)
{
    if (n_reads <= 0)
        return hipSuccess;
    if (!offsets) // (the host sends reads shorter than k to the table path)
        return count_only ? launch_tiles<true>(bases, n_reads, fixed_len, p, stream)
                          : launch_tiles<false>(bases, n_reads, fixed_len, p, stream);
    return count_only ? launch_ragged<true>(bases, offsets, n_reads, base0, total_bytes, first_read, p, stream)
                      : launch_ragged<false>(bases, offsets, n_reads, base0, total_bytes, first_read, p, stream);
}

// sampled[] -> ctl[] = {first place, end} of every bucket; total[0] = the records there is room for.  partial: a word
// per 1024 buckets.
hipError_t launch_kmer_place_buckets(const KmerBulk &p, unsigned long long *partial, unsigned long long *total,
                                     hipStream_t stream)
{
    const unsigned n_blocks = (1u << p.log2_buckets) / kScanBlock; // (at least 2^10 buckets)
    hipLaunchKernelGGL(kmer_caps_partial_kernel, dim3(n_blocks), dim3(256), 0, stream, p, partial);
    hipLaunchKernelGGL(kmer_caps_scan_kernel, dim3(1), dim3(256), 0, stream, partial, n_blocks, total);
    hipLaunchKernelGGL(kmer_caps_place_kernel, dim3(n_blocks), dim3(256), 0, stream, p, partial);
    return hipGetLastError();
}

hipError_t launch_kmer_one_length(const int64_t *offsets, int64_t n_reads, unsigned long long *out, hipStream_t stream)
{
    hipLaunchKernelGGL(kmer_one_length_kernel, dim3(1024), dim3(256), 0, stream, offsets, n_reads, out);
    return hipGetLastError();
}

hipError_t launch_kmer_scatter_rate(unsigned long long *words, unsigned long long slots, long long ops, unsigned long long *sink,
                                    hipStream_t stream)
{
    const int per_thread = 64;
    const long long threads = (ops + per_thread - 1) / per_thread;
    hipLaunchKernelGGL(kmer_scatter_rate_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, words, slots,
                       per_thread, sink);
    return hipGetLastError();
}

hipError_t launch_kmer_bucket_count(const KmerBulk &p, unsigned long long *hist, unsigned long long hist_len,
                                    unsigned long long *stats, unsigned long long *big, unsigned long long big_cap,
                                    unsigned *later, unsigned *later_list, unsigned long long *to_table, unsigned *to_table_list,
                                    bool small_buckets, int n_cu, hipStream_t stream)
{
    // A wave per bucket: three workgroups of four per CU with tables of 1024 slots (52 KB of LDS each) -- or five with
    // 512 slots where the buckets are small enough for them (1 Gbp: 7.2 -> 5.8 ms; the 200 records of a 10 Gbp bucket do
    // not fit: 44 -> 92 ms with half a million buckets left to the workgroups).  Then a workgroup per bucket left over.
    if (small_buckets)
        hipLaunchKernelGGL(kmer_wave_count_kernel<kWaveSlots / 2>, dim3((unsigned)(5 * n_cu)), dim3(256), 0, stream, p, hist,
                           hist_len, stats, big, big_cap, later, later_list);
    else
        hipLaunchKernelGGL(kmer_wave_count_kernel<kWaveSlots>, dim3((unsigned)(3 * n_cu)), dim3(256), 0, stream, p, hist, hist_len,
                           stats, big, big_cap, later, later_list);
    hipLaunchKernelGGL(kmer_bucket_count_kernel, dim3((unsigned)(3 * n_cu)), dim3(256), 0, stream, p, hist, hist_len, stats, big,
                       big_cap, later, later_list, to_table, to_table_list);
    return hipGetLastError();
}

hipError_t launch_kmer_to_table(const KmerBulk &p, bool any_overflowed, const KmerTable &t, int *overflow,
                                const unsigned long long *to_table, const unsigned *to_table_list, hipStream_t stream)
{
    if (any_overflowed)
        hipLaunchKernelGGL(kmer_overflow_to_table_kernel, dim3(64, kOvfShards), dim3(256), 0, stream, p, t, overflow);
    hipLaunchKernelGGL(kmer_buckets_to_table_kernel, dim3(2048), dim3(256), 0, stream, p, t, overflow, to_table, to_table_list);
    return hipGetLastError();
}

} // namespace covest
