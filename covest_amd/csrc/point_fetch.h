// point_fetch.h -- device helpers shared by the likelihood kernels: locate a grid
// point, clamp it to the model bounds, derive the per-error-class Poisson rates.
#pragma once
#include <hip/hip_runtime.h>

#include "device_model.h"

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

// RepeatsModel.get_b_o, covest/models.py:193-208 (o >= 1).
__device__ __forceinline__ double copy_number_weight(double q1, double q2, double q, int o)
{
    if (o == 1)
        return q1;
    if (o == 2)
        return (1.0 - q1) * q2;
    return (1.0 - q1) * (1.0 - q2) * q * pow(1.0 - q, (double)(o - 3));
}

// log of the truncated-Poisson normaliser e^x - 1, following the reference's two
// regimes (c_src/covest_poissonmodule.c:20,29-31): for x <= 1e-8 it divides by x
// itself, above by expl(x) - 1.
__device__ __forceinline__ double log_trunc_norm(double x, double log_x)
{
    if (x <= 1e-8)
        return log_x;
    if (x < 1.0)
        return log(expm1(x));
    return x + log1p(-exp(-x));
}

} // namespace covest
