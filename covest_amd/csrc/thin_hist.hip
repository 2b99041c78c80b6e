// thin_hist.hip -- K-thin: expected k-mer histogram after down-sampling the reads by `factor`
// (the step before the likelihood path: covest/histogram.py:47-75 sample_histogram).
//
// A k-mer seen i times survives j times with the thinning pmf the reference uses,
//     i < 100 : binomial(i, 1/factor).pmf(j)          (scipy.stats.binom, histogram.py:60-62)
//     i >= 100: Poisson(i / factor).pmf(j), j <= i     (covest_poisson.poisson_dist, :64)
// and the expected sampled histogram is  h'[j] = sum_i h[i] pmf_i(j):  an O(B^2) sum the reference
// evaluates with O(i) long-double products per (i, j).  Here: one lane per target count j, a loop over
// the source bins, every pmf one exp of a log-domain expression with ln n! from a host table --
// deterministic (a lane adds its terms in source order), 1 exp per (i, j) pair with j <= i.
//
// The Poisson branch is the textbook pmf.  For i / factor > 200 the reference's poisson_dist differs
// from it: it rescales by e^200 once, KEEPS the reduced rate for all later j and still divides by
// e^(original rate) (c_src/covest_poissonmodule.c:88-99) -- a bug, documented in DESIGN.md and pinned by
// the oracle's faithful mode, not replicated here.
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace covest {

namespace {

__global__ __launch_bounds__(256) void thin_hist_kernel(const int32_t *__restrict__ keys,
                                                        const double *__restrict__ counts, int64_t n,
                                                        const double *__restrict__ lgam, // lgam[m] = ln m!
                                                        double log_p, double log_1mp, double prob, int64_t out_len,
                                                        double *__restrict__ out)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x + 1; // target count
    if (j > out_len)
        return;
    const double lj = lgam[j];
    double acc = 0.0;
    for (int64_t s = 0; s < n; ++s) {
        const int i = keys[s]; // wave-uniform: scalar loads
        if (i < j)
            continue;
        double lp;
        if (i < 100) {
            lp = lgam[i] - lj - lgam[i - j] + (double)j * log_p + (double)(i - j) * log_1mp;
        } else {
            const double l = (double)i * prob;
            lp = (double)j * log(l) - lj - l;
        }
        acc += counts[s] * exp(lp);
    }
    out[j - 1] = acc;
}

} // namespace

hipError_t launch_thin_hist(const int32_t *keys, const double *counts, int64_t n, const double *lgam,
                            double factor, int64_t out_len, double *out, hipStream_t stream)
{
    if (out_len <= 0)
        return hipSuccess;
    const double prob = 1.0 / factor;
    const dim3 block(256), grid((unsigned)((out_len + 255) / 256));
    hipLaunchKernelGGL(thin_hist_kernel, grid, block, 0, stream, keys, counts, n, lgam, log(prob), log1p(-prob), prob,
                       out_len, out);
    return hipGetLastError();
}

} // namespace covest
