// thin_hist.hip -- K-thin: expected k-mer histogram after down-sampling the reads by `factor`
// (the step before the likelihood path: covest/histogram.py:47-75 sample_histogram).
//
// A k-mer seen i times survives j times with the thinning pmf the reference uses,
//     i < 100 : binomial(i, 1/factor).pmf(j)          (scipy.stats.binom, histogram.py:60-62)
//     i >= 100: Poisson(i / factor).pmf(j), j <= i     (covest_poisson.poisson_dist, :64)
// and the expected sampled histogram is  h'[j] = sum_i h[i] pmf_i(j):  an O(B^2) sum the reference
// evaluates with O(i) long-double products per (i, j).  Here: one lane per target count j and chunk of
// source bins, every pmf one exp of a log-domain expression (ln n! and ln(i/factor) from host tables),
// the chunks' partial sums added in order -- deterministic, 1 exp per (i, j) pair with j <= i.
//
// The Poisson branch is the textbook pmf.  For i / factor > 200 the reference's poisson_dist differs
// from it: it rescales by e^200 once, KEEPS the reduced rate for all later j and still divides by
// e^(original rate) (c_src/covest_poissonmodule.c:88-99) -- a bug, documented in DESIGN.md and pinned by
// the oracle's faithful mode, not replicated here.
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace covest {

namespace {

constexpr int kChunks = 64; // source bins are cut into this many contiguous chunks (grid.y): 10^4 targets alone
                            // are 40 workgroups; partial sums are added in chunk order, so the result is
                            // deterministic

// partial[chunk][j-1] = sum over the chunk's source bins i >= j of counts_i * pmf_i(j)
__global__ __launch_bounds__(256) void thin_partial_kernel(const ThinSource *__restrict__ src, int64_t n,
                                                           const double *__restrict__ lgam, // lgam[m] = ln m!
                                                           double log_p, double log_1mp, int64_t out_len,
                                                           double *__restrict__ partial)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x + 1; // target count
    const int64_t per = (n + kChunks - 1) / kChunks;
    const int64_t s0 = (int64_t)blockIdx.y * per, s1 = s0 + per < n ? s0 + per : n;
    if (j > out_len)
        return;
    const double lj = lgam[j], dj = (double)j;
    double acc = 0.0;
    for (int64_t s = s0; s < s1; ++s) {
        const ThinSource e = src[s]; // wave-uniform: scalar loads
        if (e.i < j)
            continue;
        // binomial: ln C(i, j) + j ln p + (i - j) ln(1 - p);   Poisson(l = i p): j ln l - ln j! - l
        const double lp = e.i < 100 ? e.a - lj - lgam[e.i - j] + dj * log_p + (double)(e.i - j) * log_1mp
                                    : fma(dj, e.a, -lj) - e.b;
        acc = fma(e.count, exp(lp), acc);
    }
    partial[(int64_t)blockIdx.y * out_len + (j - 1)] = acc;
}

__global__ __launch_bounds__(256) void thin_sum_kernel(const double *__restrict__ partial, int64_t out_len,
                                                       double *__restrict__ out)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= out_len)
        return;
    double acc = 0.0;
    for (int c = 0; c < kChunks; ++c)
        acc += partial[(int64_t)c * out_len + j];
    out[j] = acc;
}

} // namespace

int thin_hist_chunks() { return kChunks; }

hipError_t launch_thin_hist(const ThinSource *src, int64_t n, const double *lgam, double factor, int64_t out_len,
                            double *partial, double *out, hipStream_t stream)
{
    if (out_len <= 0)
        return hipSuccess;
    const double prob = 1.0 / factor;
    const dim3 block(256), grid((unsigned)((out_len + 255) / 256), kChunks);
    hipLaunchKernelGGL(thin_partial_kernel, grid, block, 0, stream, src, n, lgam, log(prob), log1p(-prob), out_len,
                       partial);
    hipLaunchKernelGGL(thin_sum_kernel, dim3(grid.x), block, 0, stream, partial, out_len, out);
    return hipGetLastError();
}

} // namespace covest
