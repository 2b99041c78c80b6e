// streams.h -- the pmf recurrence shared by the fast kernels (ll_basic.hip,
// ll_factored.hip).
//
// A "stream" is one mixture component (copy number o, error class s): the
// weighted truncated-Poisson terms  u_j = a_os * TP(x, j),  x = o * lambda_s,
// along the histogram keys j.  TP is what the reference's C extension computes
// (c_src/covest_poissonmodule.c:7-35), i.e. x^j / j! divided by the normaliser of
// log_trunc_norm() in point_fetch.h.  Instead of the extension's O(j) product
// per term, a lane walks the keys in ascending order and keeps
//
//     v_b = u_{k0+b} * 2^SC * (k0+b)! / (k0-1)!          (k0 = first key of the tile)
//
// for which ONE multiply advances one key:  v_b = v_{b-1} * x.  The per-key
// factor 2^-SC (k0-1)!/(k0+b)! is the same for every stream, so it is applied
// once to the sum over the lane's S streams (a wave-uniform scalar from the tile
// table): 2 fp64 instructions per pmf term (multiply + add) -- SURVEY 8(d)'s
// 4-flop bound counts the same work as mul, mul, fma.
//
// Range: keys <= 16384 and tiles of <= 32 keys bound the growth inside a tile by
// (k0+b)!/(k0-1)! <= 1e140; with SC = 540 every term that is >= e^-760 (anything
// smaller is exactly 0 in the reference's double result) stays a normal double,
// and nothing overflows.  A stream is (re)anchored with one exp only when it
// enters that window; after the mode it decays to 0 on its own.
#pragma once
#include <hip/hip_runtime.h>

#include "device_model.h"
#include "fastmath.h"
#include "point_fetch.h"
#include "tiles.h"

namespace covest {

// Where a lane keeps the two per-stream constants that only enter_tile needs:
//   lx = ln x,   c = ln a_os - D(x)  (log of the weight over the normaliser).
// Registers (K-factored, whose LDS holds G), or a [2S][workgroup] array in LDS (K-basic: 32
// VGPRs less per lane is the difference between 3 and 4 waves per SIMD there).
template <int S>
struct RegAnchors {
    double lx_[S], c_[S];
    __device__ __forceinline__ void set(int s, double lx, double c) { lx_[s] = lx; c_[s] = c; }
    __device__ __forceinline__ double lx(int s) const { return lx_[s]; }
    __device__ __forceinline__ double c(int s) const { return c_[s]; }
};

template <int S>
struct LdsAnchors {
    double *mine; // this lane's column
    int stride;   // lanes per workgroup
    __device__ __forceinline__ void set(int s, double lx, double c)
    {
        mine[(2 * s) * stride] = lx;
        mine[(2 * s + 1) * stride] = c;
    }
    __device__ __forceinline__ double lx(int s) const { return mine[(2 * s) * stride]; }
    __device__ __forceinline__ double c(int s) const { return mine[(2 * s + 1) * stride]; }
};

template <int S, class Anchors = RegAnchors<S>>
struct StreamSet {
    double v[S];   // scaled running term (see header)
    double x[S];   // o * lambda_s
    Anchors an;
    unsigned gone; // (wave-uniform) streams that are off in every lane and past their mode: they never come back

    // Mixture weights and constants of one (copy number o) over the S error
    // classes: covest/models.py:85-90 (o = 1) and :217-233.
    // `log_tab`: the workgroup's LDS copy of the fast_log table (fastmath.h).  ln x and ln a need an ABSOLUTE
    // accuracy of a few 1e-16 (they are exponents of the anchors), which fast_log delivers at a quarter of the
    // device library's cost -- the prologue is 8 streams x (2 logs + exp + the normaliser) per lane.
    // `norm_tab`: the same table for the two logs inside the normaliser, or nullptr for the device library's
    // (K-factored: the shorter code there costs it more in register allocation than it saves -- measured).
    // `s0`, `total`: this set holds the error classes s0 .. s0 + S - 1 of a model with MORE than S of them (the
    // classes of one copy number are then dealt to several lanes, ll_factored.hip); `total` is the sum of n_os over
    // ALL the model's classes, in s order -- a negative value means: the S classes here are all there are.
    template <bool CANCEL = true>
    __device__ __forceinline__ void init(const DevModel &m, const double *lam, int o, bool live, const double *log_tab,
                                         const double *norm_tab = nullptr, int s0 = 0, double total = -1.0)
    {
        double n_os[S];
        double tot = 0.0;
        gone = 0u;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            x[s] = (double)o * lam[s];
            n_os[s] = m.comb[s0 + s] * (1.0 - exp_neg_rn(x[s]));
            tot += n_os[s]; // naive sum in s order
        }
        if (total >= 0.0)
            tot = total;
        if (tot == 0.0)
            tot = 1.0; // fix_zero
        // Round 5: the stream's constant  c = ln a_os - D(x)  WITHOUT the division, the log of the quotient and the
        // normaliser's exp and log, wherever the reference's roundings leave room for it.  a_os = comb_s (1 - e^-x) / tot
        // and, for x <= 200, D = ln(e^x - 1) = x + ln(1 - e^-x)  (c_src/covest_poissonmodule.c:29-31): the factor
        // (1 - e^-x) CANCELS,
        //     c = ln comb_s - ln tot - x                                   (2^-6 <= x <= 200),
        // one log a LANE (ln tot) and a host constant a class instead of a division, two logs and an exp a STREAM.
        // Beyond 200 the extension's normaliser is the 200-chunk one (log_trunc_norm: not ln(e^x - 1)) and 1 - e^-x is
        // exactly 1 in a double: c = ln comb_s - ln tot - log_trunc_norm(x).  Below 2^-6 the reference's own roundings
        // matter -- `1.0 - exp(-x)` carries a relative 1.1e-16 / x, and what the tail term makes of it (point_fetch.h
        // exp_neg_rn, log_trunc_norm) -- so there the quotient is formed and logged as the reference forms it.  Between,
        // those roundings are below 7e-15 relative, the size of the device's and glibc's difference in exp(-x) before.
        // CANCEL = false: every stream by the reference's route (K-factored: the lanes of a wave are copy numbers, their
        // rates o x lambda_s span all three regimes, and a wave that takes all three pays more than the one route costs).
        const double ln_tot = CANCEL ? fast_log(tot, log_tab) : 0.0;
        // (measured and not kept, round 5: the classes' ln x four at a time through fast_log_n, their table reads in flight
        // together -- 4 spilled registers at 128, C2 0.167 against 0.166 ms: profiles/r05_c2_ab_staged_prologue_logs_not_kept.txt)
#pragma unroll
        for (int s = 0; s < S; ++s) {
            v[s] = 0.0;
            if (live && x[s] > 0.0 && n_os[s] > 0.0) { // (a_os > 0 iff n_os > 0: tot is a finite positive number)
                const double lx = fast_log(x[s], log_tab);
                double c;
                if (CANCEL ? x[s] < 0.015625 : true) {
                    const double a = n_os[s] / tot;
                    c = a > 0.0 ? fast_log(a, log_tab) - log_trunc_norm(x[s], lx, norm_tab) : -INFINITY;
                } else if (x[s] <= 200.0) {
                    c = (m.ln_comb[s0 + s] - ln_tot) - x[s];
                } else {
                    c = (m.ln_comb[s0 + s] - ln_tot) - log_trunc_norm(x[s], lx, norm_tab);
                }
                if (c == -INFINITY)
                    x[s] = 0.0; // (a_os underflowed to 0: the stream contributes exactly 0)
                an.set(s, c == -INFINITY ? 0.0 : lx, c);
            } else { // contributes exactly 0 (x == 0: TP returns 0, c_src/covest_poissonmodule.c:15)
                x[s] = 0.0;
                an.set(s, 0.0, -INFINITY);
            }
        }
    }

    // Entering a tile: streams that are off (v == 0) and whose log-term reaches
    // the window inside this tile, or all streams at a run start, get
    // v = u_{k0-1} * 2^SC from one exp.  km1 = k0 - 1, klast = last key of the tile.
    // Returns 1 + the highest stream that is on in ANY lane of the wave (0: none): the streams above it hold exact
    // zeros in every lane, and a caller may leave them out of the tile's steps (stepN below) -- adding their zeros
    // changes no bit.  The error classes' rates fall geometrically with s (covest/models.py:74-79), so along the
    // keys the streams go out from the top: beyond the first few hundred keys one or two of the eight are left.
    __device__ __forceinline__ int enter_tile(double km1, double klast, double lgam_prev,
                                              double lgam_last, bool run_start)
    {
        return enter_tile_n<S>(km1, klast, lgam_prev, lgam_last, run_start);
    }

    // Entering a tile whose keys only enter a SUM (sp_j, covest/models.py:103: the tiles without a count of a histogram
    // with a tail): a term below e^-60 = 9e-27 adds nothing a double of that sum can hold (p_j <= 2.5; ten thousand
    // such terms are 1e-22), so a stream is anchored only once its log-term passes kSumWindowLn inside the tile, and one
    // that is on, has fallen below it and is past its mode is switched off -- a stream is walked 11 standard deviations
    // either side of its mode instead of 39.  `gone` is still decided on the real window: a later tile WITH counts takes
    // logs of p_j down to e^-745 and anchors the stream again if it reaches that far.
    static constexpr double kSumWindowLn = -60.0;
    __device__ __forceinline__ int enter_sum_tile(double km1, double klast, double lgam_prev, double lgam_last, bool run_start)
    {
        return enter_tile_n<S, true>(km1, klast, lgam_prev, lgam_last, run_start);
    }

    // ... looking at the streams 0 .. N-1 only: for a caller that has seen the others go for good (`gone`)
    // (Measured and not kept, round 5: RELATIVE retirement -- a stream switched off, lane by lane, once its term is below
    // 2^-60 of a lower class's, which it then stays for every later key (the ratio of two classes' terms falls like
    // (x_s / x_s')^j).  It takes the classes 2 .. 7 out of the first four tiles instead of the first two, and costs a
    // compare and a select per live class and tile, 26 more spilled scalars and, with a tail, 13 spilled vector
    // registers: C3 0.695 against 0.677 ms, the trimmed histogram 0.432 against 0.414 --
    // profiles/r05_c3_ab_relative_retirement_not_kept.txt.)
    template <int N, bool SUM_ONLY = false>
    __device__ __forceinline__ int enter_tile_n(double km1, double klast, double lgam_prev,
                                                double lgam_last, bool run_start)
    {
        // Round 4: the questions of all streams are asked FIRST (compares into scalar masks, no branch), then ONE branch
        // for the rare case that some stream must be anchored, then the masks of what is live / gone.  (Until then every
        // stream took three compare-then-branch round trips through the scalar unit, one after the other: a tenth of a
        // builder wave's time in K-factored.)  Same decisions, same arithmetic, stream by stream.
        uint64_t m_window[N], m_need[N], m_real[SUM_ONLY ? N : 1];
        double a0s[N];
        uint64_t any_need = 0;
#pragma unroll
        for (int s = 0; s < N; ++s) {
            m_window[s] = m_need[s] = 0;
            a0s[s] = 0.0;
            if ((gone >> s) & 1u)
                continue; // wave-uniform: nothing to test, v[s] is 0 in every lane and stays 0
            const double lx = an.lx(s), c = an.c(s);
            const double a0 = fma(km1, lx, c - lgam_prev);
            const double a1 = fma(klast, lx, c - lgam_last);
            // A stream is on only inside the window.  (At a run start every stream used to be anchored whatever a0
            // was: for a0 + ln 2^SC in (-745, -708) -- o * lambda_s between about 1083 and 1119 at key 0 -- the
            // anchor was a SUBNORMAL double, a handful of significant bits carried along by every later multiply;
            // below -745 it was 0 and the stream re-entered correctly.  Found by the C3 fixture with a tail:
            // sp_j off by 2e-10 where those streams have their mass.)
            const double top = fmax(a0, a1);
            m_window[s] = __ballot(top > (SUM_ONLY ? kSumWindowLn : kWindowLn));
            if (SUM_ONLY) {
                m_real[s] = __ballot(top > kWindowLn);
                // on, below what a sum can hold at both ends of the tile, past its mode (it only falls from here): off
                v[s] = (!(top > kSumWindowLn) && km1 >= x[s]) ? 0.0 : v[s];
            }
            m_need[s] = run_start ? m_window[s] : (__ballot(v[s] == 0.0) & m_window[s]);
            if (run_start)
                v[s] = 0.0; // what is left of the run before means nothing here
            a0s[s] = a0;
            any_need |= m_need[s];
        }
        if (any_need != 0) { // (wave-uniform; the first tiles of a run, and now and then a stream that enters the window)
            const uint64_t me = 1ull << (threadIdx.x & 63);
#pragma unroll
            for (int s = 0; s < N; ++s) {
                if (m_need[s] == 0)
                    continue;
                // 2^SC is applied exactly, by the v_ldexp_f64 at the end of exp_scaled (fastmath.h): folding
                // ln 2^SC = 374.3 into the argument would cost its ulp (5.7e-14) in every term, which
                // tail*log(1 - sp_j) amplifies by 1/(1 - sp_j).
                const double anchored = exp_scaled(a0s[s], kScaleBits);
                v[s] = (m_need[s] & me) ? anchored : v[s];
            }
        }
        int n_live = 0;
#pragma unroll
        for (int s = 0; s < N; ++s) {
            if ((gone >> s) & 1u)
                continue;
            const uint64_t m_on = __ballot(v[s] != 0.0);
            // off, outside the window and past the mode (the log-term is concave in the key, its top near x): in every
            // lane, for every later key
            const uint64_t m_stay = (SUM_ONLY ? m_real[s] : m_window[s]) | __ballot(!(km1 >= x[s]));
            if (m_on != 0)
                n_live = s + 1; // wave-uniform
            else if (m_stay == 0)
                gone |= 1u << s;
        }
        return n_live;
    }

    // At a RUN START (every stream is re-anchored there: what is left of the run before means nothing) -- the test of
    // enter_tile for "gone", on its own: a stream whose log-term is outside the window over the whole tile, in every
    // lane, and past its mode cannot come on again.  Lets a caller see BEFORE it walks the tile that only the first
    // stream is left (ll_basic.hip: the closed form starts there).  Changes nothing but `gone`.
    __device__ __forceinline__ void retire_at_run_start(double km1, double klast, double lgam_prev, double lgam_last)
    {
#pragma unroll
        for (int s = 0; s < S; ++s) {
            if ((gone >> s) & 1u)
                continue;
            const double lx = an.lx(s), c = an.c(s);
            const double a0 = fma(km1, lx, c - lgam_prev);
            const double a1 = fma(klast, lx, c - lgam_last);
            const bool in_window = fmax(a0, a1) > kWindowLn;
            if (!__any(in_window || !(km1 >= x[s]))) {
                gone |= 1u << s;
                v[s] = 0.0; // (what enter_tile does at a run start before it looks: a gone stream is skipped there)
            }
        }
    }

    // Advance every stream by one key and return the sum of the scaled terms.  (Splitting
    // the sum into two interleaved chains was measured: no gain on gfx950.)
    __device__ __forceinline__ double step()
    {
        v[0] *= x[0];
        double g = v[0];
#pragma unroll
        for (int s = 1; s < S; ++s) {
            v[s] *= x[s];
            g += v[s];
        }
        return g;
    }

    // Two keys at once: the sum at key+1 is a dot product of the CURRENT terms with x (one FMA
    // per term), the terms themselves jump two keys with x^2 (one multiply) and are summed for
    // key+2 (one add): 3 instructions per stream for two keys instead of 4.
    __device__ __forceinline__ void step2(const double (&xx)[S], double &g1, double &g2)
    {
        g1 = v[0] * x[0];
#pragma unroll
        for (int s = 1; s < S; ++s)
            g1 = fma(v[s], x[s], g1);
        v[0] *= xx[0];
        g2 = v[0];
#pragma unroll
        for (int s = 1; s < S; ++s) {
            v[s] *= xx[s];
            g2 += v[s];
        }
    }

    // step2 / squares / leave_tile over the streams 0 .. N-1 only (the others are zero in every lane: enter_tile)
    template <int N>
    __device__ __forceinline__ void step2n(const double (&xx)[S], double &g1, double &g2)
    {
        static_assert(N >= 1 && N <= S, "live streams");
        g1 = v[0] * x[0];
#pragma unroll
        for (int s = 1; s < N; ++s)
            g1 = fma(v[s], x[s], g1);
        v[0] *= xx[0];
        g2 = v[0];
#pragma unroll
        for (int s = 1; s < N; ++s) {
            v[s] *= xx[s];
            g2 += v[s];
        }
    }

    template <int N>
    __device__ __forceinline__ double step_n()
    {
        v[0] *= x[0];
        double g = v[0];
#pragma unroll
        for (int s = 1; s < N; ++s) {
            v[s] *= x[s];
            g += v[s];
        }
        return g;
    }

    // true once every stream but the first has gone for good (wave-uniform)
    __device__ __forceinline__ bool only_first_left() const { return (gone | 1u) == (~0u >> (32 - S)); }

    template <int N>
    __device__ __forceinline__ void leave_tile_n(double renorm)
    {
#pragma unroll
        for (int s = 0; s < N; ++s)
            v[s] *= renorm;
    }

    __device__ __forceinline__ void squares(double (&xx)[S]) const
    {
#pragma unroll
        for (int s = 0; s < S; ++s)
            xx[s] = x[s] * x[s];
    }

    __device__ __forceinline__ void leave_tile(double renorm)
    {
#pragma unroll
        for (int s = 0; s < S; ++s)
            v[s] *= renorm;
    }
};

} // namespace covest
