// kmer_count.hip -- K-kmer: the k-mer abundance histogram of bin/kmer_hist.py on gfx950
// (BASELINE.json config 5).
//
// Reference restated (paths in the reference checkout):
//   single_hash / hash_kmer / rehash  bin/kmer_hist.py:14-31   a=0 c=1 g=2 t=3, 2 bits per base
//   compute_counts                    :34-41   exact counts keyed by the 2k-bit integer
//   compute_histogram                 :57-64   count-of-counts
// `canonical` (min of the code and its reverse complement's) is the jellyfish -C convention that
// config 5 asks for; the reference itself is forward-strand only.
//
// Shape: one wave64 per read; lane s owns the window starting at base s (then s+64, ...), builds
// its 2k-bit code from k byte loads (neighbouring lanes overlap: L1-resident) and inserts it into
// an open-addressing table in HBM: 16-byte slots {key u64 (CAS on first touch), count u64 (atomic add)}.
// Bound: HBM random atomics -- one 8-byte read/CAS and one 8-byte add per k-mer, both in the same line, 64
// different cache lines per wave instruction (MI355X_MICROARCH.md, Global atomics: the scattered shape runs
// ~17x below the 1.3 TB/s contiguous-atomic rate); the arithmetic is noise.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "kernels.h"
#include "kmer_table.h"
#include "wave.h"

namespace covest {

namespace {

using namespace kmer;

// One window of a read: its 2k-bit code (hash_kmer / rehash, bin/kmer_hist.py:18-31), the reverse complement's, the
// canonical choice, the table.
__device__ __forceinline__ void count_window(const unsigned char *__restrict__ seq, int64_t s, int64_t len, int k,
                                             int canonical, const KmerTable &t, int *overflow)
{
    unsigned long long h, rc;
    window_codes(seq, s, len, k, h, rc);
    if (canonical && rc < h) {
        const unsigned long long x = h;
        h = rc;
        rc = x;
    }
    table_add(t, h, rc, 1ull, overflow);
}

__global__ __launch_bounds__(256) void kmer_count_kernel(const unsigned char *__restrict__ bases,
                                                         const int64_t *__restrict__ offsets,
                                                         int64_t n_reads, int64_t fixed_len, int k,
                                                         int canonical, const KmerTable t, int *overflow)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t r = (int64_t)blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave;
    if (r >= n_reads)
        return;
    const int64_t p0 = offsets ? offsets[r] : r * fixed_len;
    const int64_t len = offsets ? offsets[r + 1] - p0 : fixed_len;
    const unsigned char *seq = bases + p0;
    if (len < k) {
        // hash_kmer(seq[:k]) of a read shorter than k: the hash of what there is, counted once;
        // an empty read counts k-mer 0 (bin/kmer_hist.py:36-37)
        if (lane == 0) {
            unsigned long long h = 0, rc = 0;
            for (int i = 0; i < (int)len; ++i) {
                const unsigned long long c = base_code(seq[i]);
                h = (h << 2) | c;
            }
            rc = revcomp_code(h, k);
            if (canonical && rc < h) {
                const unsigned long long x = h;
                h = rc;
                rc = x;
            }
            table_add(t, h, rc, 1ull, overflow);
        }
        return;
    }
    const int64_t n_windows = len - k + 1;
    for (int64_t s = lane; s < n_windows; s += kWave)
        count_window(seq, s, len, k, canonical, t, overflow);
}

// Reads of ONE length (what a sequencing run's FASTQ holds, and config 5's synthetic reads): the windows of all reads
// are numbered through -- read = index / windows per read -- and a wave takes 64 consecutive ones, so that no lane
// idles in a read's last 64-window round (100-base reads have 80 windows: 64 + 16 lanes, 37 % of the slots empty).
__global__ __launch_bounds__(256) void kmer_count_fixed_kernel(const unsigned char *__restrict__ bases, int64_t n_reads,
                                                               int64_t len, int k, int canonical, const KmerTable t,
                                                               int *overflow)
{
    const int64_t n_windows = len - k + 1; // (the host sends reads shorter than k to kmer_count_kernel)
    const int64_t total = n_reads * n_windows;
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= total)
        return;
    const int64_t r = w / n_windows;
    count_window(bases + r * len, w - r * n_windows, len, k, canonical, t, overflow);
}

// Re-insert every entry of `src` into the (larger) `dst` table.
__global__ __launch_bounds__(256) void kmer_rehash_kernel(const KmerTable src, const KmerTable dst, int *overflow)
{
    const unsigned long long n = src.mask + 1;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        const KmerSlot e = src.slots[i];
        if (e.key != kEmptyKey)
            table_add(dst, e.key, revcomp_code(e.key, dst.k), e.count, overflow);
    }
}

__global__ __launch_bounds__(256) void kmer_fill_empty_kernel(KmerSlot *slots, unsigned long long n)
{
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (unsigned long long)gridDim.x * blockDim.x)
        slots[i] = KmerSlot{kEmptyKey, 0ull}; // one 16-byte store per lane: coalesced streaming write
}

// stats[0] = max count, stats[1] = distinct keys
__global__ __launch_bounds__(256) void kmer_stats_kernel(const KmerTable t, unsigned long long *stats)
{
    const unsigned long long n = t.mask + 1;
    unsigned long long distinct = 0, mx = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        const KmerSlot e = t.slots[i];
        if (e.key != kEmptyKey) {
            ++distinct;
            mx = e.count > mx ? e.count : mx;
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        distinct += __shfl_xor(distinct, off, kWave);
        const unsigned long long o = __shfl_xor(mx, off, kWave);
        mx = o > mx ? o : mx;
    }
    if ((threadIdx.x & (kWave - 1)) == 0) {
        atomicMax(&stats[0], mx);
        atomicAdd(&stats[1], distinct);
    }
}

// compute_histogram: hist[c] = number of keys with count c, c < hist_len.  Low counts go through
// LDS-private bins (one flush of atomics per workgroup), the rare high ones straight to HBM.
constexpr int kLdsBins = 4096;
__global__ __launch_bounds__(256) void kmer_histogram_kernel(const KmerTable t, unsigned long long *hist,
                                                             unsigned long long hist_len)
{
    __shared__ unsigned bins[kLdsBins];
    for (int i = threadIdx.x; i < kLdsBins; i += blockDim.x)
        bins[i] = 0u;
    __syncthreads();
    const unsigned long long n = t.mask + 1;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        const KmerSlot e = t.slots[i];
        if (e.key != kEmptyKey) {
            const unsigned long long c = e.count;
            if (c < (unsigned long long)kLdsBins)
                atomicAdd(&bins[c], 1u);
            else if (c < hist_len)
                atomicAdd(&hist[c], 1ull);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kLdsBins; i += blockDim.x)
        if (bins[i] != 0u && (unsigned long long)i < hist_len)
            atomicAdd(&hist[i], (unsigned long long)bins[i]);
}

unsigned grid_for(unsigned long long n, unsigned cap = 256 * 16)
{
    const unsigned long long blocks = (n + 255) / 256;
    return (unsigned)(blocks < cap ? (blocks ? blocks : 1) : cap);
}

} // namespace

hipError_t launch_kmer_fill_empty(const KmerTable &t, hipStream_t stream)
{
    const unsigned long long n = t.mask + 1;
    hipLaunchKernelGGL(kmer_fill_empty_kernel, dim3(grid_for(n)), dim3(256), 0, stream, t.slots, n);
    return hipGetLastError();
}

hipError_t launch_kmer_count(const unsigned char *bases, const int64_t *offsets, int64_t n_reads,
                             int64_t fixed_len, int k, int canonical, const KmerTable &t, int *overflow,
                             hipStream_t stream)
{
    if (n_reads <= 0)
        return hipSuccess;
    if (!offsets && fixed_len >= k) {
        // reads of one length: 64 consecutive windows per wave (kmer_count_fixed_kernel), at most 2^23 workgroups
        // of 256 windows per launch, cut at whole reads
        const int64_t n_windows = fixed_len - k + 1;
        const int64_t reads_per_launch = std::max<int64_t>(1, (((int64_t)1 << 31) - 256) / n_windows);
        for (int64_t first = 0; first < n_reads; first += reads_per_launch) {
            const int64_t n = std::min(n_reads - first, reads_per_launch);
            const int64_t total = n * n_windows;
            hipLaunchKernelGGL(kmer_count_fixed_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream,
                               bases + first * fixed_len, n, fixed_len, k, canonical, t, overflow);
        }
        return hipGetLastError();
    }
    // HIP wraps a grid of more than 2^32 threads silently: at most 2^23 workgroups per launch
    const int reads_per_block = 4;
    const int64_t reads_per_launch = (int64_t)reads_per_block << 23;
    for (int64_t first = 0; first < n_reads; first += reads_per_launch) {
        const int64_t n = n_reads - first < reads_per_launch ? n_reads - first : reads_per_launch;
        const dim3 grid((unsigned)((n + reads_per_block - 1) / reads_per_block));
        hipLaunchKernelGGL(kmer_count_kernel, grid, dim3(reads_per_block * kWave), 0, stream,
                           offsets ? bases : bases + first * fixed_len, offsets ? offsets + first : nullptr, n,
                           fixed_len, k, canonical, t, overflow);
    }
    return hipGetLastError();
}

hipError_t launch_kmer_rehash(const KmerTable &src, const KmerTable &dst, int *overflow, hipStream_t stream)
{
    hipLaunchKernelGGL(kmer_rehash_kernel, dim3(grid_for(src.mask + 1)), dim3(256), 0, stream, src, dst, overflow);
    return hipGetLastError();
}

hipError_t launch_kmer_stats(const KmerTable &t, unsigned long long *stats, hipStream_t stream)
{
    hipLaunchKernelGGL(kmer_stats_kernel, dim3(grid_for(t.mask + 1)), dim3(256), 0, stream, t, stats);
    return hipGetLastError();
}

hipError_t launch_kmer_histogram(const KmerTable &t, unsigned long long *hist, unsigned long long hist_len,
                                 hipStream_t stream)
{
    hipLaunchKernelGGL(kmer_histogram_kernel, dim3(grid_for(t.mask + 1)), dim3(256), 0, stream, t, hist, hist_len);
    return hipGetLastError();
}

} // namespace covest
