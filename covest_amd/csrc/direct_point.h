// direct_point.h -- the log-likelihood of ONE grid point computed by ONE wave64, the body of
// K-direct (ll_direct.hip); and, with the same arithmetic, the strict evaluation of single keys for
// the points a recurrence kernel hands back (strict_pj_wave here; ll_fix_list_kernel in argmin.hip).
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
#include "tiles.h"
#include "wave.h"

namespace covest {


// ---- the hand-back of the recurrence kernels: a RANGE of tile rows per point ----
// Where keys with h_j != 0 have a subnormal p_j -- typically a run of a few dozen keys at one end of the counted
// keys -- the reference's value hangs on the rounding of every single term onto the 4.9e-324 grid (DESIGN.md
// section 2), which only the term-by-term evaluation reproduces.  How far the recurrence kernels' own p_j can be
// off there is bounded.  With g = 4.94e-324: the reference rounds each of the S terms of a copy number twice
// (<= g each, weighted by b_o, and the b_o sum to <= 1) and each copy number's share once (g / 2); the recurrence
// kernels round each G[o][j] once and each accumulation step once.  So |p_j - p_j(reference)| <= (S + T) g, which
// moves the log-likelihood by at most h_j (S + T) g / p_j -- and |LL| >= 708 h_j because of that very key.  So above
//     p_clamp = (S + T_max) * 7e-317     (T_max: the largest threshold_o of the launch)
// the recurrence kernels' value is within 1e-10 of the reference's whatever the roundings were, and nothing needs
// doing.  Below it they take log(max(p_j, p_clamp)) (one v_max per log, everywhere), so what such a key
// contributed is KNOWN -- h_j log(p_clamp) -- and they name, in a 64-bit side word per point, the first and last
// tile row (row = 32 * tile + position, tiles.h) at which they met one: single rows (K-basic) or units of 16 rows
// (K-factored: the half tile of a weight vector).  ll_fix_list_kernel (argmin.hip) then evaluates the counted rows
// of that range strictly (K-direct's arithmetic) and, where the strict p_j is below p_clamp, replaces the known
// contribution by h_j safe_log(p_j).
constexpr unsigned long long kSubFieldMask = 0xFFFFFull; // 20 bits: rows < 2^20 (16384 keys, one tile each, at worst)
constexpr double kClampPerTerm = 7e-317; // see above
// WHEN IS p_j ZERO IN THE REFERENCE?  (round 4: found by comparing K-basic with K-direct on ALL 10^6 points of C2 -- one
// point, p_j = 0.9987 x 2^-1075, was -inf here and finite there and in the reference.)  The reference rounds every
// term onto the 4.94e-324 grid on its way: the extension's cast to double (c_src/covest_poissonmodule.c:32, ties to even:
// anything above half a grid step survives), then a_os * TP (covest/models.py:93,238: survives if a_os > 1/2), then, repeats
// model, b_o * (sum over s) (:237-241: survives if b_o > 1/2).  So the reference's p_j can be ONE GRID STEP when the exact
// value is as small as 1/4 of a step (basic model) or 1/8 (repeats model) -- and h_j log(4.94e-324) is finite where
// h_j log(0) is -inf.  A recurrence kernel's own product, rounded once, is 0 below 1/2 step: it must not decide.  Below
// kZeroSteps grid steps (EXACT value, tested before the product underflows) a p_j is zero in the reference whatever the
// roundings were; between that and p_clamp the row is handed back to the term-by-term evaluation.
constexpr double kZeroSteps = 0.12;
constexpr double kGridStep = 4.94065645841246544e-324; // 2^-1074
// kZeroSteps grid steps TIMES 2^sh, for values carried with that factor (K-basic's p_j 2^64, the chunks' shares
// 2^128).  Multiplied in THIS order: kZeroSteps * kGridStep alone rounds to 0 (round 4, first cut: every row below
// the clamp went to the strict kernel, 78 instead of 29 us on C2)
constexpr double zero_steps_scaled(double two_to_sh) { return kZeroSteps * (kGridStep * two_to_sh); }

__host__ __device__ inline unsigned long long sub_word(unsigned first, unsigned last, bool units16)
{
    return (1ull << 62) | ((unsigned long long)(units16 ? 1 : 0) << 60) | ((unsigned long long)last << 20) |
           (unsigned long long)first;
}
// The queue of handed-back points of one launch: (local point index, side word) pairs appended with one atomic
// each by the thread that writes the point's value.  Capacity = the number of points, so it cannot overflow.
// ll_fix_list_kernel (argmin.hip) drains it, one wave per entry.
struct SubList {
    double p_clamp;           // (S + T_max) * kClampPerTerm of this launch, and its log (libm, host)
    double log_p_clamp;
    unsigned *count;          // device counter; zero before the launch
    int64_t *index;           // [capacity] index into the launch's LL buffer
    unsigned long long *word; // [capacity]
    int64_t index_offset;     // added by the launcher when it cuts a launch into parts
#ifdef COVEST_DIAG
    // DIAGNOSTIC builds only (tiles.h): K-basic writes, INSTEAD of the log-likelihood, 1: the class of the point's
    // route through its closed form (ll_basic.hip kClass*), 2: the smaller of log p_j at the first and at the last
    // counted key the closed form was asked about (NaN where it was not asked).  tools/dump_c2_classes.py uses it to
    // choose WHERE the reference is asked (tests/golden/make_golden.py section c2classes); env COVEST_DIAG_BASIC_CLASS.
    int diag_class;
#endif
#ifdef __HIPCC__
    __device__ __forceinline__ void push(int64_t idx, unsigned long long w) const
    {
        const unsigned at = atomicAdd(count, 1u);
        index[at] = idx;
        word[at] = w;
    }
#endif
};

__host__ __device__ inline unsigned sub_first(unsigned long long w) { return (unsigned)(w & kSubFieldMask); }
__host__ __device__ inline unsigned sub_last(unsigned long long w) { return (unsigned)((w >> 20) & kSubFieldMask); }
__host__ __device__ inline bool sub_units16(unsigned long long w) { return ((w >> 60) & 1) != 0; }

constexpr int kDirectBinsPerLane = 4; // bins held in registers per lane per pass

// p_j of ONE bin at one point, by the whole wave, with K-direct's arithmetic: the same expressions, the same
// order of the two sums (error classes inside, copy numbers outside: covest/models.py:92-97, :235-241), every
// term rounded to a double on its own.  All lanes call it with the same arguments and get the same value.
// par: the point's parameters AFTER clamp_point; key / neg_lgam: the bin's j and -lgamma(j + 1).
template <int P>
__device__ __forceinline__ double strict_pj_wave(const DevModel &m, const double *par, int T, double key, double neg_lgam)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int S = m.n_err;
    const int OT = kWave / S;
    const int s = lane % S;
    const int og = lane / S;
    const bool lane_in_tile = og < OT;
    const double lam = error_class_rate(m, par[0], par[1], s);
    const double comb_s = m.comb[s];
    double p = 0.0;
    for (int o0 = 1; o0 < T; o0 += OT) {
        const int o = o0 + og;
        const bool live = lane_in_tile && o < T;
        const double x = (double)o * lam;
        const double n_os = comb_s * (1.0 - exp_neg_rn(x));
        double tot = 0.0;
        for (int t = 0; t < S; ++t)
            tot += __shfl(n_os, og * S + t, kWave);
        if (tot == 0.0)
            tot = 1.0;
        double a_os = n_os / tot;
        const double b_o = (P == 5) ? copy_number_weight(par[2], par[3], par[4], o) : 1.0;
        double lx = 0.0, nd = -INFINITY;
        if (live && x > 0.0) {
            lx = log(x);
            nd = -log_trunc_norm(x, lx);
        }
        if (!live)
            a_os = 0.0;
        const double term = a_os != 0.0 ? a_os * exp(fma(key, lx, nd + neg_lgam)) : 0.0;
        const int n_o = min(OT, T - o0);
        for (int g = 0; g < n_o; ++g) { // wave-uniform; copy numbers in ascending order
            double inner = 0.0;
            for (int t = 0; t < S; ++t)
                inner += __shfl(term, g * S + t, kWave);
            p += __shfl(b_o, g * S, kWave) * inner;
        }
    }
    return p;
}

// ln(LDBL_MAX) of the x87 long double the reference's product lives in (c_src/covest_poissonmodule.c:19-24)
constexpr double kLnLdblMax = 11356.523406294143949492;

// All 64 lanes of a wave call this with the same point; every lane returns the point's LL.
// REF_OVF (COVEST_KERNEL_DIRECT_REF): the reference's OVERFLOW reproduced.  truncated_poisson forms the whole product
// prod_{i <= j} (x / i) in long double before any scaling (c_src/covest_poissonmodule.c:22-24) and returns +inf once the
// running product passes LDBL_MAX -- it grows until i = floor(x), so the term of (x, j) is +inf iff
// m ln x - ln m! > ln LDBL_MAX for m = min(j, floor(x)).  Then p_j = +inf, log p_j = +inf, and the likelihood is +inf,
// or NaN where another counted key has p_j = 0 (inf - inf); sp_j = min(1, fsum(...)) = 1 and the tail term is dropped
// (covest/models.py:103-105).  optimize_grid WOULD select such a point (covest/grid.py:65-70: -(+inf) < anything).
template <int P, bool WRITE_P, bool REF_OVF = false>
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
    bool saw_special = false; // REF_OVF: a p_j that is +inf (or NaN) -- sp_j is then not < 1

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
            // REF_OVF: can this component's product overflow at all (its largest value, at i = floor(x)), and from
            // which key on is that value reached
            double fx = INFINITY;
            if (REF_OVF && live && x >= 1.0) {
                const double top = floor(x);
                if (fma(top, lx, -lgamma(top + 1.0)) > kLnLdblMax)
                    fx = top;
            }

            // ---- every lane accumulates all components for its own bins ----
            const int n_comp = min(OT, T - o0) * S;
            for (int i = 0; i < n_comp; ++i) {
                const double a_i = wave_bcast(a_os, i);
                if (a_i != 0.0) { // wave-uniform; NaN falls through and poisons p_j as in the reference
                    const double l_i = wave_bcast(lx, i);
                    const double d_i = wave_bcast(nd, i);
                    const double fx_i = REF_OVF ? wave_bcast(fx, i) : INFINITY;
                    if (REF_OVF && fx_i < INFINITY) { // (wave-uniform, rare) this component overflows somewhere
#pragma unroll
                        for (int b = 0; b < kBinsPerLane; ++b) {
                            // the running product grows up to i = floor(x): past LDBL_MAX at this key?
                            const bool ovf = key[b] >= fx_i || fma(key[b], l_i, nlg[b]) > kLnLdblMax;
                            inner[b] += a_i * (ovf ? INFINITY : exp(fma(key[b], l_i, d_i + nlg[b])));
                        }
                    } else {
#pragma unroll
                        for (int b = 0; b < kBinsPerLane; ++b)
                            inner[b] += a_i * exp(fma(key[b], l_i, d_i + nlg[b]));
                    }
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
                if (REF_OVF && !(p[b] < INFINITY))
                    saw_special = true; // (+inf or NaN: kept out of the compensated sum, whose error terms it would poison)
                else
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
        if (REF_OVF && __any(saw_special))
            tail_term = 0.0; // min(1, fsum(...)) of a sum with +inf (or NaN) in it is 1: covest/models.py:103-104
    }
    return acc_ll + tail_term;
}

} // namespace covest
