// kmer_wide.hip -- K-kmer for k > 31: keys of W 64-bit words.
//
// The reference hashes a k-mer into a Python integer of 2k bits, whatever k is (bin/kmer_hist.py:18-31): nothing
// there stops at 64 bits.  kmer_count.hip packs key and empty marker into one word (k <= 31, the case every
// benchmark runs); this file holds the general case -- W = 2, 4 or 8 words, k <= 63, 127, 255 -- with the same
// semantics (exact counts keyed by the integer; a read shorter than k counts the hash of what there is; `canonical`
// takes the smaller of the code and its reverse complement's as integers).
//
// Table: open addressing, linear probing, slots of `stride` = 4, 8 or 16 words (32 / 64 / 128 bytes, a power of two
// so that a slot never straddles a 128-byte line):
//     word 0      state: 0 = empty, bit 63 = a writer holds the slot, else the COUNT of the key (>= 1)
//     words 1..W  the key, most significant word first
// A key cannot be claimed by one compare-and-swap, so a slot is taken in two steps: CAS state 0 -> LOCKED, write
// the key with write-through stores, drain them (s_waitcnt vmcnt(0): they are at the L2, where every reader looks),
// then publish state = the first count.  A lane that meets a LOCKED slot looks again -- the key being written may
// be its own -- and never waits on a lane of its own wave: the writer publishes within the loop iteration in which
// it took the lock, before any lane of the wave comes round again.  Readers load state and key past their L1
// (agent-scope atomic loads): a line cached while the slot was still empty would otherwise hide the key.
// Capability path: 1 + W loads and one atomic per occurrence, a CAS and W + 1 stores per new key; K-kmer's line
// sharing and four-bases-per-load tricks are not repeated here.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "kernels.h"
#include "wave.h"

namespace covest {

namespace {

typedef unsigned long long u64;
constexpr u64 kLocked = 1ull << 63;
constexpr int kWideMaxProbe = 1 << 16;
constexpr int kWideMaxSpin = 1 << 20;

__device__ __forceinline__ unsigned wide_base_code(unsigned char ch)
{
    const unsigned x = (ch >> 1) & 3u; // a=0 c=1 t=2 g=3; x ^ (x >> 1) swaps g and t
    return x ^ (x >> 1);
}

template <int W>
struct WideKey {
    u64 w[W]; // w[0] most significant
    __device__ __forceinline__ void clear()
    {
#pragma unroll
        for (int i = 0; i < W; ++i)
            w[i] = 0;
    }
    // h <<= 2; h |= c   (hash_kmer, bin/kmer_hist.py:18-23)
    __device__ __forceinline__ void push(u64 c)
    {
#pragma unroll
        for (int i = 0; i < W - 1; ++i)
            w[i] = (w[i] << 2) | (w[i + 1] >> 62);
        w[W - 1] = (w[W - 1] << 2) | c;
    }
    // h |= c << (2 pos)
    __device__ __forceinline__ void set(int pos, u64 c)
    {
        const int word = W - 1 - (pos >> 5), sh = 2 * (pos & 31);
#pragma unroll
        for (int i = 0; i < W; ++i)
            if (i == word)
                w[i] |= c << sh;
    }
    __device__ __forceinline__ u64 get(int pos) const
    {
        const int word = W - 1 - (pos >> 5), sh = 2 * (pos & 31);
        u64 v = 0;
#pragma unroll
        for (int i = 0; i < W; ++i)
            if (i == word)
                v = (w[i] >> sh) & 3ull;
        return v;
    }
    // keep the low 2k bits
    __device__ __forceinline__ void mask_to(int k)
    {
        const int bits = 2 * k;
#pragma unroll
        for (int i = 0; i < W; ++i) {
            const int lo = 64 * (W - 1 - i); // bit position of this word's bit 0
            if (bits <= lo)
                w[i] = 0;
            else if (bits < lo + 64)
                w[i] &= (1ull << (bits - lo)) - 1ull;
        }
    }
    __device__ __forceinline__ bool less_than(const WideKey &o) const
    {
#pragma unroll
        for (int i = 0; i < W; ++i)
            if (w[i] != o.w[i])
                return w[i] < o.w[i];
        return false;
    }
    __device__ __forceinline__ u64 hash() const
    {
        u64 h = 0x9E3779B97F4A7C15ull;
#pragma unroll
        for (int i = 0; i < W; ++i) {
            h ^= w[i];
            h *= 0xD6E8FEB86659FD93ull;
            h ^= h >> 32;
        }
        return h * 0x9E3779B97F4A7C15ull;
    }
};

// the reverse complement of the k-mer coded by `x` (complement = 3 - base), over k positions
template <int W>
__device__ __forceinline__ WideKey<W> wide_revcomp(const WideKey<W> &x, int k)
{
    WideKey<W> rc;
    rc.clear();
    for (int i = 0; i < k; ++i)
        rc.set(k - 1 - i, 3ull - x.get(i));
    return rc;
}

template <int W>
__device__ __forceinline__ void wide_add(const KmerWideTable t, const WideKey<W> &key, u64 add, int *overflow)
{
    u64 h = key.hash() >> (64 - t.log2_slots);
    int spins = 0;
    for (int probe = 0; probe < kWideMaxProbe;) {
        u64 *slot = t.words + h * (u64)t.stride;
        u64 s = __hip_atomic_load(&slot[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (s == 0) {
            const u64 prev = atomicCAS(&slot[0], 0ull, kLocked);
            if (prev == 0) { // ours: key first, drained to the L2, then the count publishes it
#pragma unroll
                for (int i = 0; i < W; ++i)
                    __hip_atomic_store(&slot[1 + i], key.w[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(&slot[0], add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
            s = prev;
        }
        if (s & kLocked) { // somebody is writing this slot's key -- it may be ours: look again
            if (++spins > kWideMaxSpin) {
                *overflow = 1;
                return;
            }
            continue;
        }
        bool same = true;
#pragma unroll
        for (int i = 0; i < W; ++i)
            same = same && __hip_atomic_load(&slot[1 + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == key.w[i];
        if (same) {
            atomicAdd(&slot[0], add);
            return;
        }
        h = (h + 1) & t.mask;
        ++probe;
    }
    *overflow = 1;
}

// One wave per read, lane = window start (then + 64, ...): compute_counts, bin/kmer_hist.py:34-41.
template <int W>
__global__ __launch_bounds__(256) void kmer_wide_count_kernel(const unsigned char *__restrict__ bases,
                                                              const int64_t *__restrict__ offsets, int64_t n_reads,
                                                              int64_t fixed_len, int canonical, const KmerWideTable t,
                                                              int *overflow)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t r = (int64_t)blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave;
    if (r >= n_reads)
        return;
    const int k = t.k;
    const int64_t p0 = offsets ? offsets[r] : r * fixed_len;
    const int64_t len = offsets ? offsets[r + 1] - p0 : fixed_len;
    const unsigned char *seq = bases + p0;
    if (len < k) {
        // hash_kmer(seq[:k]) of a read shorter than k: the hash of what there is, counted once; an empty read
        // counts k-mer 0 (bin/kmer_hist.py:36-37)
        if (lane == 0) {
            WideKey<W> h;
            h.clear();
            for (int i = 0; i < (int)len; ++i)
                h.push(wide_base_code(seq[i]));
            if (canonical) {
                const WideKey<W> rc = wide_revcomp(h, k);
                if (rc.less_than(h))
                    h = rc;
            }
            wide_add(t, h, 1ull, overflow);
        }
        return;
    }
    const int64_t n_windows = len - k + 1;
    for (int64_t s = lane; s < n_windows; s += kWave) {
        WideKey<W> h, rc;
        h.clear();
        rc.clear();
        for (int i = 0; i < k; ++i) {
            const u64 c = wide_base_code(seq[s + i]);
            h.push(c);              // hash_kmer, :18-23 (rehash :26-31 yields the same window code)
            rc.set(i, 3ull - c);    // reverse complement, built back to front
        }
        h.mask_to(k);
        if (canonical && rc.less_than(h))
            h = rc;
        wide_add(t, h, 1ull, overflow);
    }
}

template <int W>
__global__ __launch_bounds__(256) void kmer_wide_rehash_kernel(const KmerWideTable src, const KmerWideTable dst,
                                                               int *overflow)
{
    const u64 n = src.mask + 1;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        const u64 *slot = src.words + i * (u64)src.stride;
        const u64 s = slot[0];
        if (s != 0 && !(s & kLocked)) {
            WideKey<W> key;
#pragma unroll
            for (int j = 0; j < W; ++j)
                key.w[j] = slot[1 + j];
            wide_add(dst, key, s, overflow);
        }
    }
}

__global__ __launch_bounds__(256) void kmer_wide_clear_kernel(u64 *words, u64 n_words)
{
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (u64)gridDim.x * blockDim.x)
        words[i] = 0ull;
}

// stats[0] = max count, stats[1] = distinct keys
__global__ __launch_bounds__(256) void kmer_wide_stats_kernel(const KmerWideTable t, u64 *stats)
{
    const u64 n = t.mask + 1;
    u64 distinct = 0, mx = 0;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        const u64 s = t.words[i * (u64)t.stride];
        if (s != 0) {
            ++distinct;
            mx = s > mx ? s : mx;
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        distinct += __shfl_xor(distinct, off, kWave);
        const u64 o = __shfl_xor(mx, off, kWave);
        mx = o > mx ? o : mx;
    }
    if ((threadIdx.x & (kWave - 1)) == 0) {
        atomicMax(&stats[0], mx);
        atomicAdd(&stats[1], distinct);
    }
}

// compute_histogram (bin/kmer_hist.py:57-64): hist[c] = number of keys with count c
__global__ __launch_bounds__(256) void kmer_wide_histogram_kernel(const KmerWideTable t, u64 *hist, u64 hist_len)
{
    const u64 n = t.mask + 1;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        const u64 s = t.words[i * (u64)t.stride];
        if (s != 0 && s < hist_len)
            atomicAdd(&hist[s], 1ull);
    }
}

unsigned wide_grid_for(u64 n, unsigned cap = 256 * 16)
{
    const u64 blocks = (n + 255) / 256;
    return (unsigned)(blocks < cap ? (blocks ? blocks : 1) : cap);
}

template <int W>
hipError_t count_w(const unsigned char *bases, const int64_t *offsets, int64_t n_reads, int64_t fixed_len, int canonical,
                   const KmerWideTable &t, int *overflow, hipStream_t stream)
{
    // HIP wraps a grid of more than 2^32 threads silently: at most 2^23 workgroups per launch
    const int reads_per_block = 4;
    const int64_t reads_per_launch = (int64_t)reads_per_block << 23;
    for (int64_t first = 0; first < n_reads; first += reads_per_launch) {
        const int64_t n = n_reads - first < reads_per_launch ? n_reads - first : reads_per_launch;
        const dim3 grid((unsigned)((n + reads_per_block - 1) / reads_per_block));
        hipLaunchKernelGGL((kmer_wide_count_kernel<W>), grid, dim3(reads_per_block * kWave), 0, stream,
                           offsets ? bases : bases + first * fixed_len, offsets ? offsets + first : nullptr, n, fixed_len,
                           canonical, t, overflow);
    }
    return hipGetLastError();
}

} // namespace

hipError_t launch_kmer_wide_clear(const KmerWideTable &t, hipStream_t stream)
{
    const u64 n = (t.mask + 1) * (u64)t.stride;
    hipLaunchKernelGGL(kmer_wide_clear_kernel, dim3(wide_grid_for(n)), dim3(256), 0, stream, t.words, n);
    return hipGetLastError();
}

hipError_t launch_kmer_wide_count(const unsigned char *bases, const int64_t *offsets, int64_t n_reads, int64_t fixed_len,
                                  int canonical, const KmerWideTable &t, int *overflow, hipStream_t stream)
{
    if (n_reads <= 0)
        return hipSuccess;
    if (t.w == 2)
        return count_w<2>(bases, offsets, n_reads, fixed_len, canonical, t, overflow, stream);
    if (t.w == 4)
        return count_w<4>(bases, offsets, n_reads, fixed_len, canonical, t, overflow, stream);
    if (t.w == 8)
        return count_w<8>(bases, offsets, n_reads, fixed_len, canonical, t, overflow, stream);
    return hipErrorInvalidValue;
}

hipError_t launch_kmer_wide_rehash(const KmerWideTable &src, const KmerWideTable &dst, int *overflow, hipStream_t stream)
{
    const dim3 grid(wide_grid_for(src.mask + 1));
    if (src.w == 2)
        hipLaunchKernelGGL((kmer_wide_rehash_kernel<2>), grid, dim3(256), 0, stream, src, dst, overflow);
    else if (src.w == 4)
        hipLaunchKernelGGL((kmer_wide_rehash_kernel<4>), grid, dim3(256), 0, stream, src, dst, overflow);
    else if (src.w == 8)
        hipLaunchKernelGGL((kmer_wide_rehash_kernel<8>), grid, dim3(256), 0, stream, src, dst, overflow);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_kmer_wide_stats(const KmerWideTable &t, unsigned long long *stats, hipStream_t stream)
{
    hipLaunchKernelGGL(kmer_wide_stats_kernel, dim3(wide_grid_for(t.mask + 1)), dim3(256), 0, stream, t, stats);
    return hipGetLastError();
}

hipError_t launch_kmer_wide_histogram(const KmerWideTable &t, unsigned long long *hist, unsigned long long hist_len,
                                      hipStream_t stream)
{
    hipLaunchKernelGGL(kmer_wide_histogram_kernel, dim3(wide_grid_for(t.mask + 1)), dim3(256), 0, stream, t, hist, hist_len);
    return hipGetLastError();
}

} // namespace covest
