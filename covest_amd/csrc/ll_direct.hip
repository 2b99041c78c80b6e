// ll_direct.hip -- K-direct: the general likelihood kernel for gfx950.
//
// One wave64 per grid point (4 points per 256-thread workgroup); the evaluation of a
// point is direct_point.h.  It is the
// kernel BASELINE.json's north_star sketches: every lane owns histogram bins,
// the (copy number o, error class s) mixture components are prepared lane-
// parallel and broadcast through the scalar unit, every pmf term costs one
// fp64 exp, and the per-bin log terms are reduced with wavefront shuffles.
// It accepts ANY point list (each point its own threshold_o), any key set
// (sparse keys, zero counts), both models, and is the cross-check for the
// faster kernels.
//
// Reference restated (paths relative to the reference checkout):
//   BasicModel.compute_probabilities    covest/models.py:81-98
//   RepeatsModel.compute_probabilities  covest/models.py:211-242
//   BasicModel.compute_loglikelihood    covest/models.py:100-107
//   truncated_poisson                   c_src/covest_poissonmodule.c:7-35
//
// pmf term in the log domain:  TP(x, j) = exp(j*ln x - lgamma(j+1) - D(x)), D the log
// of the normaliser the reference divides by (log_trunc_norm in point_fetch.h).
// The reference's O(j) long-double product is replaced by this O(1) form; the
// difference is bounded by the rounding of lgamma(j+1) (<= 1 ulp of ~8e4 at
// j = 10^4, i.e. ~1e-11 relative on a term).
//
// Roofline: fp64 VALU.  Algorithmic work per point = n_bins * S * (T-1) pmf
// terms; this kernel spends ~30 fp64 instructions per term (the exp), so it sits
// far below the 4 flop/term recurrence bound by construction -- it is the
// correctness baseline, not the fast path.
#include <hip/hip_runtime.h>

#include "direct_point.h"
#include "kernels.h"

namespace covest {

namespace {

template <int P, bool WRITE_P, bool REF_OVF = false>
__global__ __launch_bounds__(256) void ll_direct_kernel(const DevModel m, const PointSource src,
                                                        const int64_t n, double *__restrict__ out_ll,
                                                        double *__restrict__ out_p)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t pt = (int64_t)blockIdx.x * (blockDim.x / kWave) + (threadIdx.x / kWave);
    if (pt >= n)
        return; // wave-uniform
    const double ll = direct_point_ll<P, WRITE_P, REF_OVF>(m, src, pt, out_p);
    if (lane == 0)
        out_ll[pt] = ll;
}

} // namespace

hipError_t launch_ll_direct(const DevModel &m, const PointSource &src, int64_t n, double *out_ll,
                            double *out_p, hipStream_t stream, bool ref_overflow)
{
    if (n <= 0)
        return hipSuccess;
    // HIP wraps a grid of more than 2^32 threads silently: at most 2^23 workgroups per launch
    const int waves_per_block = 4;
    const dim3 block(waves_per_block * kWave);
    const int64_t per_launch = (int64_t)waves_per_block << 23;
    for (int64_t first = 0; first < n; first += per_launch) {
        const int64_t cnt = n - first < per_launch ? n - first : per_launch;
        const dim3 grid((unsigned)((cnt + waves_per_block - 1) / waves_per_block));
        PointSource part = src;
        if (src.is_grid) {
            part.flat_begin = src.flat_begin + first;
        } else {
            part.params = src.params + first * (m.kind == 0 ? 2 : 5);
            part.t_list = src.t_list ? src.t_list + first : nullptr;
        }
        double *out = out_ll + first;
        if (ref_overflow && !out_p) { // COVEST_KERNEL_DIRECT_REF (direct_point.h REF_OVF)
            if (m.kind == 0)
                hipLaunchKernelGGL((ll_direct_kernel<2, false, true>), grid, block, 0, stream, m, part, cnt, out, out_p);
            else
                hipLaunchKernelGGL((ll_direct_kernel<5, false, true>), grid, block, 0, stream, m, part, cnt, out, out_p);
        } else if (m.kind == 0) {
            if (out_p)
                hipLaunchKernelGGL((ll_direct_kernel<2, true>), grid, block, 0, stream, m, part, cnt, out, out_p);
            else
                hipLaunchKernelGGL((ll_direct_kernel<2, false>), grid, block, 0, stream, m, part, cnt, out, out_p);
        } else {
            if (out_p)
                hipLaunchKernelGGL((ll_direct_kernel<5, true>), grid, block, 0, stream, m, part, cnt, out, out_p);
            else
                hipLaunchKernelGGL((ll_direct_kernel<5, false>), grid, block, 0, stream, m, part, cnt, out, out_p);
        }
    }
    return hipGetLastError();
}

} // namespace covest
