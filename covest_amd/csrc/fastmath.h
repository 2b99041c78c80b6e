// fastmath.h -- fp64 log for the per-bin term h_j * log(p_j) of the fast kernels.
//
// The device library's log costs ~72 FMA-issue slots on gfx950 (tools/
// microbench_f64.hip) -- it became the dominant cost once a pmf term is down to 2
// instructions.  The log-likelihood needs log(p_j) to an ABSOLUTE accuracy of a few
// 1e-16 (it is summed with weights h_j into a total whose terms all have the same
// sign), so: exponent/mantissa split (v_frexp_*), a 32-entry table {1/c, log c'}
// in LDS indexed by the top 5 mantissa bits, r = fma(m, 1/c, -1) with |r| <= 2^-6,
// and log1p(r) to r^8.  ~18 instructions, absolute error < 2e-16 (relative
// < 2e-16 for |log x| > 1) for x in (0, 1].
#pragma once
#include <hip/hip_runtime.h>

#include "log_table.h"

namespace covest {

// Copy the table into this workgroup's LDS (call from all threads, then barrier).
__device__ __forceinline__ void load_log_table(double *tab_lds)
{
    if (threadIdx.x < 64)
        tab_lds[threadIdx.x] = kLogTable[threadIdx.x];
}

// log(x) for finite x > 0 (subnormals included).  NaN propagates.
__device__ __forceinline__ double fast_log(double x, const double *tab_lds)
{
    const double m = __builtin_amdgcn_frexp_mant(x); // [0.5, 1)
    const int e = __builtin_amdgcn_frexp_exp(x);
    const int idx = (__double2hiint(m) >> 15) & 31;
    const double2 ent = *reinterpret_cast<const double2 *>(tab_lds + 2 * idx);
    const double r = fma(m, ent.x, -1.0);
    double q = fma(r, -0.125, 1.0 / 7.0);
    q = fma(r, q, -1.0 / 6.0);
    q = fma(r, q, 0.2);
    q = fma(r, q, -0.25);
    q = fma(r, q, 1.0 / 3.0);
    q = fma(r, q, -0.5);
    const double lp = fma(r * r, q, r); // log1p(r)
    return fma((double)e, 0.693147180559945309417232121458, ent.y) + lp;
}

} // namespace covest
