// point_fetch.h -- device helpers shared by the likelihood kernels: locate a grid
// point, clamp it to the model bounds, derive the per-error-class Poisson rates.
#pragma once
#include <hip/hip_runtime.h>

#include "device_model.h"
#include "fastmath.h"

namespace covest {

// Parameters of point `i` of the launch and its threshold_o.
// Grid mode decomposes the flat itertools.product index, last axis fastest
// (covest/grid.py:43).  Everything here is wave-uniform.
template <int P>
__device__ __forceinline__ void fetch_point(const PointSource &src, int64_t i, double *par, int &T)
{
    T = 2; // basic model: the single copy-number class o = 1
    if (src.is_grid) {
        int64_t idx = src.flat_begin + i;
        int64_t coord[kMaxParams];
#pragma unroll
        for (int d = P - 1; d >= 0; --d) {
            const int64_t len = src.len[d];
            const int64_t q = idx / len;
            coord[d] = idx - q * len;
            idx = q;
            par[d] = src.axis[d][coord[d]];
        }
        if (P == 5)
            T = src.t_table[(coord[2] * src.len[3] + coord[3]) * src.len[4] + coord[4]];
    } else {
#pragma unroll
        for (int d = 0; d < P; ++d)
            par[d] = src.params[i * P + d];
        if (P == 5)
            T = src.t_list[i];
    }
}

// BasicModel.fit_to_bounds, covest/models.py:60-69 (NaN bound = None).
template <int P>
__device__ __forceinline__ void clamp_point(const DevModel &m, double *par)
{
#pragma unroll
    for (int d = 0; d < P; ++d) {
        const double lo = m.lo[d], hi = m.hi[d];
        double v = par[d];
        if (lo == lo && v < lo)
            v = lo;
        else if (hi == hi && v > hi)
            v = hi;
        par[d] = v;
    }
}

// correct_c + _get_lambda_s, covest/models.py:71-79, same evaluation order:
// ((ck * 3**-s) * (1-e)**(k-s)) * e**s with ck = c*(r-k+1)/r.
__device__ __forceinline__ double error_class_rate(const DevModel &m, double c, double err, int s)
{
    const double ck = c * (double)(m.r - m.k + 1) / (double)m.r;
    double v = ck * m.pow3neg[s];
    v = v * pow(1.0 - err, (double)(m.k - s));
    v = v * pow(err, (double)s);
    return v;
}

// ONE class's rate by multiplication, for a kernel whose lanes are classes (ll_fix_basic_packed_kernel): the products of
// error_class_rates<S> below in the same order -- (1 - e)^max(k - S + 1, 0) by squaring, one more factor a class down to
// s, e^s factor by factor -- so the strict re-evaluation of a K-basic point works with the very rates K-basic had.
__device__ __forceinline__ double error_class_rate_mul(const DevModel &m, double c, double err, int s, int S)
{
    const double ck = c * (double)(m.r - m.k + 1) / (double)m.r;
    const double q = 1.0 - err;
    const int n_low = m.k - (S - 1) > 0 ? m.k - (S - 1) : 0;
    double pw = 1.0, sq = q;
    for (int n = n_low; n > 0; n >>= 1) { // q^n_low by squaring (wave-uniform)
        if (n & 1)
            pw *= sq;
        sq *= sq;
    }
    for (int t = S - 1; t > s; --t) // q^max(k - s, 0): class t - 1 has one more factor than class t, where k - t >= 0
        if (m.k - t >= 0)
            pw *= q;
    double pe = 1.0; // e^s (0^0 = 1: the error-free class at e = 0)
    for (int t = 0; t < s; ++t)
        pe *= err;
    return ((ck * m.pow3neg[s]) * pw) * pe; // the reference's order of evaluation, covest/models.py:76-79
}

// ALL S error classes' rates of one point at once, for a kernel whose every lane is a point of its own (K-basic):
// (1 - e)^(k - s) and e^s by multiplication -- (1 - e)^(k - S + 1) by squaring, then one multiply a class -- instead of
// two calls of pow a class.  Round 4: the device library's pow is 210 instructions, and sixteen of them were HALF of
// K-basic's instruction count on C2.  The products differ from pow's by a few 1e-16 relative (the reference's own pow and
// the device's differ by as much); the log-likelihood moves by less than 1e-12 of itself.  Classes beyond k (padding of
// the class count to a multiple of 8: comb = 0, they weigh nothing) get the exponent 0.
template <int S>
__device__ __forceinline__ void error_class_rates(const DevModel &m, double c, double err, double (&lam)[S])
{
    const double ck = c * (double)(m.r - m.k + 1) / (double)m.r;
    const double q = 1.0 - err;
    const int n_low = m.k - (S - 1) > 0 ? m.k - (S - 1) : 0; // the smallest exponent of q among the classes (wave-uniform)
    double pw = 1.0, sq = q;
    for (int n = n_low; n > 0; n >>= 1) { // q^n_low by squaring
        if (n & 1)
            pw *= sq;
        sq *= sq;
    }
    double qp[S]; // q^(k - s)
#pragma unroll
    for (int s = S - 1; s >= 0; --s) {
        qp[s] = pw;        // q^max(k - s, 0)
        if (m.k - s >= 0)  // (wave-uniform) the class before has one more factor
            pw *= q;
    }
    double pe = 1.0; // e^s (0^0 = 1: the error-free class at e = 0)
#pragma unroll
    for (int s = 0; s < S; ++s) {
        lam[s] = ((ck * m.pow3neg[s]) * qp[s]) * pe; // the reference's order of evaluation, covest/models.py:76-79
        pe *= err;
    }
}

// RepeatsModel.get_b_o, covest/models.py:193-208 (o >= 1).
__device__ __forceinline__ double copy_number_weight(double q1, double q2, double q, int o)
{
    if (o == 1)
        return q1;
    if (o == 2)
        return (1.0 - q1) * q2;
    return (1.0 - q1) * (1.0 - q2) * q * pow(1.0 - q, (double)(o - 3));
}

// The same weight with (1 - q)^(o - 3) by squaring (at most 14 squarings for o <= 16384: about 1e-15 relative) instead
// of the device library's pow (210 instructions): for the strict re-evaluation of handed-back rows (argmin.hip), where
// every lane of every lot wants one and the terms are rounded onto the 4.9e-324 grid anyway.
__device__ __forceinline__ double copy_number_weight_by_squaring(double q1, double q2, double q, int o)
{
    if (o == 1)
        return q1;
    if (o == 2)
        return (1.0 - q1) * q2;
    double pw = 1.0, sq = 1.0 - q;
    for (unsigned n = (unsigned)(o - 3); n != 0; n >>= 1) {
        pw = (n & 1u) ? pw * sq : pw;
        sq *= sq;
    }
    return (1.0 - q1) * (1.0 - q2) * q * pw;
}

// exp(-x), x >= 0, as the reference's libm returns it.  The mixture weights are
// n_os = comb[s] * (1.0 - exp(o * -l_s)) (covest/models.py:87,221) -- deliberately
// NOT expm1 -- so for small x the last bit of exp(-x) decides n_os to a relative
// 1.1e-16 / x (1e-8 at x = 1e-8).  glibc's exp is correctly rounded there to
// within a 1e-8 ulp margin, so the correctly rounded value is computed here from
// the Taylor series in double-double instead of trusting the device library's
// last bit.  Above 2^-6 that sensitivity is < 1e-14 and the device exp is used.
__device__ __forceinline__ double exp_neg_rn(double x)
{
    if (!(x < 0.015625))
        return exp_fast(-x);
    const double p = x * x;
    const double pe = fma(x, x, -p); // x^2 = p + pe exactly
    const double tail = (p * x) * (-1.0 / 6 + x * (1.0 / 24 + x * (-1.0 / 120 + x * (1.0 / 720 +
                        x * (-1.0 / 5040 + x * (1.0 / 40320 + x * (-1.0 / 362880)))))));
    double s, e, r, e2;
    const double mx = -x, hp = 0.5 * p;
    s = mx + hp; // two-sum of -x and x^2/2
    {
        const double bb = s - mx;
        e = (mx - (s - bb)) + (hp - bb);
    }
    const double lo = e + (0.5 * pe + tail);
    r = 1.0 + s; // two-sum of 1 and s
    {
        const double bb = r - 1.0;
        e2 = (1.0 - (r - bb)) + (s - bb);
    }
    return r + (e2 + lo);
}

// log of the normaliser the reference divides the pmf product by
// (c_src/covest_poissonmodule.c:20,25-31).  It is NOT log(e^x - 1) for x > 200:
// the extension divides by e^200 once per 200 taken off x and then by
// expl(x_res) - 1 of the residual x_res in (0, 200], i.e. by
// e^(200 n) * (e^x_res - 1), which is SMALLER than e^x - 1 by the factor
// (1 - e^-x_res): the reference's pmf is too large by up to 1/(1 - e^-x_res) when
// x is just above a multiple of 200.  Parity is with what the reference computes,
// so this is reproduced, together with its two small-argument branches: x <= 1e-8 divides by
// x itself (:20,29), and a residual <= 1e-8 divides by the ORIGINAL x (:29-31).
// `log_tab`: optional LDS copy of the fast_log table (fastmath.h) -- the two logs below need absolute accuracy.
__device__ __forceinline__ double log_trunc_norm(double x, double log_x, const double *log_tab = nullptr)
{
    if (x <= 1e-8)
        return log_x;
    double base = 0.0, xr = x;
    if (x > 200.0) {
        double n = ceil(x / 200.0) - 1.0;
        xr = fma(-200.0, n, x); // exact: x and 200 n are multiples of ulp(x)
        if (xr > 200.0) {
            n += 1.0;
            xr -= 200.0;
        } else if (xr <= 0.0) {
            n -= 1.0;
            xr += 200.0;
        }
        base = 200.0 * n;
        if (xr <= 1e-8)
            return base + log_x;
    }
    if (xr < 1.0) {
        // expl(x) - 1 in x87 long double: e^x is rounded to a 2^-63 grid BEFORE the
        // subtraction (:30), a relative noise of up to 2^-64 / x (5e-12 at 1e-8) that
        // the tail term tail*log(1 - sp_j) amplifies when sp_j is close to 1.
        double m = expm1(xr);
        if (xr < 0x1p-10)
            m = rint(m * 0x1p63) * 0x1p-63;
        return base + (log_tab ? fast_log(m, log_tab) : log(m));
    }
    // ln(e^xr - 1) = xr + ln(1 - e^-xr), e^-xr <= 0.37: the rounding of 1 - e^-xr is an absolute 1.1e-16
    return base + (xr + (log_tab ? fast_log(1.0 - exp_fast(-xr), log_tab) : log1p(-exp_fast(-xr))));
}

} // namespace covest
