// ll_direct.hip -- K-direct: the general likelihood kernel for gfx950.
//
// One wave64 per grid point (4 points per 256-thread workgroup).  It is the
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

#include "kernels.h"
#include "point_fetch.h"
#include "wave.h"

namespace covest {

namespace {

constexpr int kBinsPerLane = 4; // bins held in registers per lane per pass

template <int P, bool WRITE_P>
__global__ __launch_bounds__(256) void ll_direct_kernel(const DevModel m, const PointSource src,
                                                        const int64_t n, double *__restrict__ out_ll,
                                                        double *__restrict__ out_p)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t pt = (int64_t)blockIdx.x * (blockDim.x / kWave) + (threadIdx.x / kWave);
    if (pt >= n)
        return; // wave-uniform

    double par[kMaxParams];
    int T;
    fetch_point<P>(src, pt, par, T);
    clamp_point<P>(m, par);

    const int S = m.n_err;
    const int OT = kWave / S; // copy-number classes prepared per tile
    const int s = lane % S;
    const int og = lane / S;
    const bool lane_in_tile = og < OT;
    const double lam = error_class_rate(m, par[0], par[1], s);
    const double comb_s = m.comb[s];

    double acc_ll = 0.0;
    CompSum acc_sp = {0.0, 0.0};
    const int64_t n_bins = m.bins.n;

    for (int64_t base = 0; base < n_bins; base += (int64_t)kWave * kBinsPerLane) {
        double key[kBinsPerLane], nlg[kBinsPerLane], p[kBinsPerLane], inner[kBinsPerLane];
#pragma unroll
        for (int b = 0; b < kBinsPerLane; ++b) {
            const int64_t idx = base + (int64_t)b * kWave + lane;
            const bool ok = idx < n_bins;
            key[b] = ok ? m.bins.key[idx] : 0.0;
            nlg[b] = ok ? -m.bins.lgam[idx] : 0.0;
            p[b] = 0.0;
            inner[b] = 0.0;
        }

        for (int o0 = 1; o0 < T; o0 += OT) {
            // ---- lane-parallel preparation of up to OT*S mixture components ----
            const int o = o0 + og;
            const bool live = lane_in_tile && o < T;
            const double x = (double)o * lam;         // o * l_s[s]            models.py:238
            const double ex = exp_neg_rn(x);          // exp(o * -l_s[s])      models.py:221
            const double n_os = comb_s * (1.0 - ex);  // NOT expm1, as the reference
            double tot = 0.0;                         // naive sum in s order  models.py:225
            for (int t = 0; t < S; ++t)
                tot += __shfl(n_os, og * S + t, kWave);
            if (tot == 0.0)
                tot = 1.0;                            // fix_zero
            double a_os = n_os / tot;
            const double b_o = (P == 5) ? copy_number_weight(par[2], par[3], par[4], o) : 1.0;
            double lx = 0.0, nd = -INFINITY; // exp(key*0 - inf) = 0: component contributes a_os*0
            if (live && x > 0.0) {
                lx = log(x);
                nd = -log_trunc_norm(x, lx);
            }
            if (!live)
                a_os = 0.0;

            // ---- every lane accumulates all components for its own bins ----
            const int n_comp = min(OT, T - o0) * S;
            for (int i = 0; i < n_comp; ++i) {
                const double a_i = wave_bcast(a_os, i);
                if (a_i != 0.0) { // wave-uniform; NaN falls through and poisons p_j as in the reference
                    const double l_i = wave_bcast(lx, i);
                    const double d_i = wave_bcast(nd, i);
#pragma unroll
                    for (int b = 0; b < kBinsPerLane; ++b)
                        inner[b] += a_i * exp(fma(key[b], l_i, d_i + nlg[b]));
                }
                if ((i + 1) % S == 0) { // end of one copy-number class: p_j += b_o * inner  models.py:237
                    const double b_i = wave_bcast(b_o, i);
#pragma unroll
                    for (int b = 0; b < kBinsPerLane; ++b) {
                        p[b] += b_i * inner[b];
                        inner[b] = 0.0;
                    }
                }
            }
        }

        // ---- bin epilogue: sp_j contribution and h_j * safe_log(p_j)  models.py:103-106 ----
#pragma unroll
        for (int b = 0; b < kBinsPerLane; ++b) {
            const int64_t idx = base + (int64_t)b * kWave + lane;
            if (idx < n_bins) {
                const double h = m.bins.cnt[idx];
                acc_sp.add(p[b]);
                if (h != 0.0)
                    acc_ll += h * ((p[b] <= 0.0) ? -INFINITY : log(p[b]));
                if (WRITE_P)
                    out_p[idx] = p[b];
            }
        }
    }

    acc_ll = wave_sum(acc_ll);
    double tail_term = 0.0;
    if (m.tail != 0.0) { // tail == 0: the term is 0 * finite = 0 in the reference
        double sp = wave_comp_sum(acc_sp);
        if (!(sp < 1.0))
            sp = 1.0; // min(1, fsum(...)), NaN -> 1
        if (sp < 1.0)
            tail_term = m.tail * log(1.0 - sp);
    }
    if (lane == 0)
        out_ll[pt] = acc_ll + tail_term;
}

} // namespace

hipError_t launch_ll_direct(const DevModel &m, const PointSource &src, int64_t n, double *out_ll,
                            double *out_p, hipStream_t stream)
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
        if (m.kind == 0) {
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
