// direct_point.h -- the log-likelihood of ONE grid point computed by ONE wave64, the body of
// K-direct (ll_direct.hip).  Also called from the arg-min pass (argmin.hip) for the points a fast
// kernel hands back (kRedoBits): there every wave re-evaluates the flagged points of its 64.
//
// Every lane owns histogram bins, the (copy number o, error class s) mixture components are
// prepared lane-parallel and broadcast through the scalar unit, every pmf term costs one fp64
// exp, and the per-bin log terms are reduced with wavefront shuffles.  The terms are formed and
// rounded one by one, in the reference's order -- which is what makes this the strict kernel
// where p_j is a subnormal double.
//
// Reference restated (paths relative to the reference checkout):
//   BasicModel.compute_probabilities    covest/models.py:81-98
//   RepeatsModel.compute_probabilities  covest/models.py:211-242
//   BasicModel.compute_loglikelihood    covest/models.py:100-107
//   truncated_poisson                   c_src/covest_poissonmodule.c:7-35
#pragma once
#include <hip/hip_runtime.h>

#include "device_model.h"
#include "point_fetch.h"
#include "wave.h"

namespace covest {

// "Evaluate this point again with the strict kernel": the value a recurrence kernel (K-basic,
// K-factored) writes for a point at which some key with h_j != 0 has a SUBNORMAL p_j -- there the
// reference's result depends on the rounding of every single term onto the 4.9e-324 grid, which
// only the term-by-term evaluation reproduces.  A quiet NaN with a payload no arithmetic produces;
// it never leaves the library (argmin.hip / capi.cpp replace it by K-direct's value).
constexpr unsigned long long kRedoBits = 0x7FF8C0DE5B0A0001ull;

__device__ __forceinline__ double redo_marker() { return __longlong_as_double((long long)kRedoBits); }
__device__ __forceinline__ bool is_redo_marker(double v)
{
    return (unsigned long long)__double_as_longlong(v) == kRedoBits;
}

constexpr double kMinNormal = 2.2250738585072014e-308; // DBL_MIN: below it p_j is subnormal (or 0)

constexpr int kDirectBinsPerLane = 4; // bins held in registers per lane per pass

// All 64 lanes of a wave call this with the same point; every lane returns the point's LL.
template <int P, bool WRITE_P>
__device__ __forceinline__ double direct_point_ll(const DevModel &m, const PointSource &src, int64_t pt,
                                                  double *__restrict__ out_p)
{
    constexpr int kBinsPerLane = kDirectBinsPerLane;
    const int lane = threadIdx.x & (kWave - 1);
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
    return acc_ll + tail_term;
}

} // namespace covest
