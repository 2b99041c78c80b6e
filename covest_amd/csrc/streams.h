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
#include "point_fetch.h"
#include "tiles.h"

namespace covest {

template <int S>
struct StreamSet {
    double v[S];   // scaled running term (see header)
    double x[S];   // o * lambda_s
    double lx[S];  // ln x
    double c[S];   // ln a_os - D(x): log of the weight over the normaliser

    // Mixture weights and constants of one (copy number o) over the S error
    // classes: covest/models.py:85-90 (o = 1) and :217-233.
    __device__ __forceinline__ void init(const DevModel &m, const double *lam, int o, bool live)
    {
        double n_os[S];
        double tot = 0.0;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            x[s] = (double)o * lam[s];
            n_os[s] = m.comb[s] * (1.0 - exp_neg_rn(x[s]));
            tot += n_os[s]; // naive sum in s order
        }
        if (tot == 0.0)
            tot = 1.0; // fix_zero
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const double a = n_os[s] / tot;
            v[s] = 0.0;
            if (live && x[s] > 0.0 && a > 0.0) {
                lx[s] = log(x[s]);
                c[s] = log(a) - log_trunc_norm(x[s], lx[s]);
            } else { // contributes exactly 0 (x == 0: TP returns 0, c_src/covest_poissonmodule.c:15)
                x[s] = 0.0;
                lx[s] = 0.0;
                c[s] = -INFINITY;
            }
        }
    }

    // Entering a tile: streams that are off (v == 0) and whose log-term reaches
    // the window inside this tile, or all streams at a run start, get
    // v = u_{k0-1} * 2^SC from one exp.  km1 = k0 - 1, klast = last key of the tile.
    __device__ __forceinline__ void enter_tile(double km1, double klast, double lgam_prev,
                                               double lgam_last, bool run_start)
    {
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const double a0 = fma(km1, lx[s], c[s] - lgam_prev);
            const double a1 = fma(klast, lx[s], c[s] - lgam_last);
            const bool need = run_start ? true : (v[s] == 0.0 && fmax(a0, a1) > kWindowLn);
            if (__any(need)) {
                // 2^SC is applied exactly (v_ldexp_f64) wherever exp(a0) itself is a normal
                // double: folding ln 2^SC = 374.3 into the argument would cost its ulp
                // (5.7e-14) in every term, which tail*log(1 - sp_j) amplifies by 1/(1 - sp_j).
                const bool deep = a0 < -700.0;
                const double e0 = exp(deep ? a0 + kScaleLn : a0);
                const double anchored = deep ? e0 : ldexp(e0, kScaleBits);
                v[s] = need ? anchored : v[s];
            }
        }
    }

    // Advance every stream by one key and return the sum of the scaled terms.
    // Advance every stream by one key and return the sum of the scaled terms.  (Splitting
    // the sum into two interleaved chains was measured: no gain on gfx950.)
    __device__ __forceinline__ double step()
    {
        double g = 0.0;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            v[s] *= x[s];
            g += v[s];
        }
        return g;
    }

    __device__ __forceinline__ void leave_tile(double renorm)
    {
#pragma unroll
        for (int s = 0; s < S; ++s)
            v[s] *= renorm;
    }
};

} // namespace covest
