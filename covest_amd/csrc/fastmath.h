// fastmath.h -- fp64 log for the per-bin term h_j * log(p_j) of the fast kernels.
//
// The device library's log costs ~72 FMA-issue slots on gfx950 (tools/
// microbench_f64.hip) -- it became the dominant cost once a pmf term is down to 2
// instructions.  The log-likelihood needs log(p_j) to an ABSOLUTE accuracy of a few
// 1e-16 (it is summed with weights h_j into a total whose terms all have the same
// sign), so: exponent/mantissa split (v_frexp_*), a 256-entry table {1/c, log c'}
// in LDS (4 KB) indexed by the top 8 mantissa bits, r = fma(m, 1/c, -1) with |r| <= 2^-9,
// and log1p(r) to r^5 (a 64-entry table needs r^7: two more FMAs per log).  ~16 instructions,
// absolute error < 2e-16 (relative < 2e-16 for |log x| > 1) for x in (0, 1].
#pragma once
#include <hip/hip_runtime.h>

#include "log_table.h"

namespace covest {

constexpr int kLogTableDoubles = 2 << kLogTableBits;

// Copy the table into this workgroup's LDS (call from all threads, then barrier).
__device__ __forceinline__ void load_log_table(double *tab_lds)
{
    for (int i = threadIdx.x; i < kLogTableDoubles; i += blockDim.x)
        tab_lds[i] = kLogTable[i];
}

// d = a * b + c with the coefficient c in an SGPR pair, as the three-address VOP3 instruction.
// Left to itself the compiler keeps the Horner coefficients in VGPRs, picks the two-address
// v_fmac_f64 and copies the coefficient into the destination first -- 7 extra full-rate slots
// per log (16% of K-basic).  A scalar operand costs nothing and frees the VGPRs.
__device__ __forceinline__ double fma_vvs(double a, double b, double c)
{
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c));
    return d;
}

// d = a * b + c as the three-address instruction with every operand in vector registers: where d is a LOOP-CARRIED value
// and c a freshly loaded one the compiler takes v_fmac_f64 into c's register and copies the result back (a v_mov_b64, one
// full-rate slot, per Horner step of K-factored's shared steps).
__device__ __forceinline__ double fma_vvv(double a, double b, double c)
{
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

// exp(x) 2^sc: the device library's algorithm (ocml expD: x = n ln 2 + t by Cody-Waite, a degree-11 polynomial in t,
// |t| <= 0.347, ldexp) with its OWN coefficients, written so that none of them lives in a vector register.  Inlined from
// the library the eleven coefficients are loop invariants that the compiler parks in 22 VGPRs for the whole key walk of
// a recurrence kernel (and copies into the v_fmac destination before every use, a full-rate v_mov_b64 each): a sixth
// of K-basic's register file -- round 5, found in the ISA of ll_basic_kernel<true>, which spilled 266 registers.  Here
// every coefficient is a scalar operand of a three-address v_fma_f64 (the scalar unit re-creates it with two s_mov).
// The power of two is applied by the final v_ldexp_f64 -- exactly, one rounding in all, where the result is a normal
// double -- so a caller that wants exp(x) 2^540 (streams.h) no longer adds ln 2^540 to the argument or scales twice.
// x = -inf gives 0, NaN propagates, results beyond the doubles' range are +inf / 0 as v_ldexp_f64 rounds them.
__device__ __forceinline__ double exp_scaled(double x, int sc)
{

    const double dn = __builtin_rint(x * 0x1.71547652b82fep+0);
    double t = fma(-dn, 0x1.62e42fefa39efp-1, x);
    t = fma(-dn, 0x1.abc9e3b39803fp-56, t);
    // (the first Horner step as two instructions with one scalar operand each -- gfx950's VOP3 reads ONE scalar pair --
    // behind asm: left to the compiler the pair is contracted into a v_fmac whose addend is hoisted into a VGPR pair)
    double p;
    asm("v_mul_f64 %0, %1, %2" : "=v"(p) : "v"(t), "s"(0x1.ade156a5dcb37p-26));
    asm("v_add_f64 %0, %1, %2" : "=v"(p) : "v"(p), "s"(0x1.28af3fca7ab0cp-22));
    p = fma_vvs(t, p, 0x1.71dee623fde64p-19);
    p = fma_vvs(t, p, 0x1.a01997c89e6b0p-16);
    p = fma_vvs(t, p, 0x1.a01a014761f6ep-13);
    p = fma_vvs(t, p, 0x1.6c16c1852b7b0p-10);
    p = fma_vvs(t, p, 0x1.1111111122322p-7);
    p = fma_vvs(t, p, 0x1.55555555502a1p-5);
    p = fma_vvs(t, p, 0x1.5555555555511p-3);
    p = fma_vvs(t, p, 0x1.000000000000bp-1);
    p = fma(t, p, 1.0);
    p = fma(t, p, 1.0);
    // (|x| beyond 1100: the result is 0 or +inf whatever the polynomial says; keeps n + sc inside the ints)
    const int n = (int)fmin(fmax(dn, -4096.0), 4096.0);
    const double z = ldexp(p, n + sc);
    return x == -INFINITY ? 0.0 : z;
}

__device__ __forceinline__ double exp_fast(double x) { return exp_scaled(x, 0); }

// log(x) for finite x > 0 (subnormals included).  NaN propagates.  x == 0 gives a finite
// value (callers handle p_j <= 0 themselves).
__device__ __forceinline__ double fast_log(double x, const double *tab_lds)
{
    const double m = __builtin_amdgcn_frexp_mant(x); // [0.5, 1)
    const int e = __builtin_amdgcn_frexp_exp(x);
    // byte offset of the 16-byte entry: the top kLogTableBits mantissa bits
    const unsigned off = __builtin_amdgcn_ubfe((unsigned)__double2hiint(m), 20 - kLogTableBits, kLogTableBits) << 4;
    const double2 ent = *reinterpret_cast<const double2 *>(reinterpret_cast<const char *>(tab_lds) + off);
    const double r = fma(m, ent.x, -1.0); // |r| <= 2^-(kLogTableBits + 1)
    double q;
    if (kLogTableBits >= 8) { // |r| <= 2^-9: log1p(r) to r^5, truncation r^6/6 < 2^-56
        q = fma_vvs(r, 0.2, -0.25);
    } else { // |r| <= 2^-7: to r^7, truncation r^8/8 < 2^-59
        q = fma_vvs(r, 1.0 / 7.0, -1.0 / 6.0);
        q = fma_vvs(r, q, 0.2);
        q = fma_vvs(r, q, -0.25);
    }
    q = fma_vvs(r, q, 1.0 / 3.0);
    q = fma(r, q, -0.5);
    const double lp = fma(r * r, q, r); // log1p(r)
    return fma((double)e, 0.693147180559945309417232121458, ent.y) + lp;
}

// The raw instruction: llvm's fmax quiets its operands first (a v_max_f64 x, x, x each) -- two extra full-rate
// slots per log for a NaN nobody feeds it (a NaN parameter poisons the result by other means).
__device__ __forceinline__ double max_raw(double a, double b)
{
    double d;
    asm("v_max_f64 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

// N logs at once, written stage by stage so that the N table reads are in flight together and the N dependent
// chains interleave: one log after the other exposes the LDS latency and the latency of every FMA of its chain,
// which two waves per SIMD do not hide (ll_factored.hip).
template <int N>
__device__ __forceinline__ void fast_log_n(const double (&x)[N], double (&out)[N], const double *tab_lds)
{
    double m[N], r[N], q[N], lp[N];
    int e[N];
    double2 ent[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        m[k] = __builtin_amdgcn_frexp_mant(x[k]);
        e[k] = __builtin_amdgcn_frexp_exp(x[k]);
    }
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const unsigned off = __builtin_amdgcn_ubfe((unsigned)__double2hiint(m[k]), 20 - kLogTableBits, kLogTableBits) << 4;
        ent[k] = *reinterpret_cast<const double2 *>(reinterpret_cast<const char *>(tab_lds) + off);
    }
#pragma unroll
    for (int k = 0; k < N; ++k)
        r[k] = fma(m[k], ent[k].x, -1.0);
    static_assert(kLogTableBits >= 8, "log1p to r^5 needs |r| <= 2^-9");
#pragma unroll
    for (int k = 0; k < N; ++k)
        q[k] = fma_vvs(r[k], 0.2, -0.25);
#pragma unroll
    for (int k = 0; k < N; ++k)
        q[k] = fma_vvs(r[k], q[k], 1.0 / 3.0);
#pragma unroll
    for (int k = 0; k < N; ++k)
        q[k] = fma(r[k], q[k], -0.5);
#pragma unroll
    for (int k = 0; k < N; ++k)
        lp[k] = fma(r[k] * r[k], q[k], r[k]);
#pragma unroll
    for (int k = 0; k < N; ++k)
        out[k] = fma((double)e[k], 0.693147180559945309417232121458, ent[k].y) + lp[k];
}

// N logs of POSITIVE doubles scaled by 2^SC, log(x 2^-SC): fast_log_n with the 2^-SC folded into the exponent
// arithmetic -- the exponent comes off the bits with a shift and an add (two half-rate integer instructions, the
// price of v_frexp_exp_i32_f64 alone), so the scale costs nothing; x must be a NORMAL double (the callers clamp).
// (Taking the mantissa off the bits too -- and, or -- was tried: the compiler copies the low word into a fresh
// register pair for it, v_frexp_mant_f64 is cheaper.)  DEG: log1p(r) to r^DEG, |r| <= 2^-9: 5 -> truncation 9e-18
// (as fast_log); 4 -> r^5 / 5 <= 5.7e-15, zero-mean over the mantissa (odd in r) -- one FMA less per log; 3 (round 5,
// dense grids: ll_factored.hip) -> another FMA less: r - (1/2 + d) r^2 + r^3 / 3 with d = 0.2071 x 2^-18 chosen so
// that d r^2 - r^4 / 4 equioscillates over |r| <= 2^-9 -- an absolute 6.2e-13 at most (3.6e-12 for the plain Taylor
// cubic), against sums whose terms are |log p_j| >= 1 and a parity bar of 1e-9.
// RAW (DEG 3 only): the exponent is not un-biased -- the result is log(x 2^-SC) + kLogRawBias(SC), one integer
// subtract less per log; the caller takes the constant off once per SUM (h_j kLogRawBias summed over the counted keys
// is a constant of the histogram).
template <int SC>
constexpr double kLogRawBias = (double)(1022 + SC) * 0.693147180559945309417232121458;
template <int N, int DEG, int SC, bool RAW = false>
__device__ __forceinline__ void fast_log_bits_n(const double (&x)[N], double (&out)[N], const double *tab_lds)
{
    static_assert(kLogTableBits == 8, "table offset below takes the top 8 mantissa bits");
    static_assert(DEG == 3 || DEG == 4 || DEG == 5, "log1p degree");
    double m[N], r[N], q[N], lp[N];
    int e[N];
    double2 ent[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const int hi = __double2hiint(x[k]);
        e[k] = RAW ? (hi >> 20) : (hi >> 20) - (1022 + SC);
        m[k] = __builtin_amdgcn_frexp_mant(x[k]); // [0.5, 1): the mantissa bits are x's own
        const unsigned off = ((unsigned)hi >> 8) & 0xFF0u;
        ent[k] = *reinterpret_cast<const double2 *>(reinterpret_cast<const char *>(tab_lds) + off);
    }
#pragma unroll
    for (int k = 0; k < N; ++k)
        r[k] = fma(m[k], ent[k].x, -1.0);
    if (DEG == 5) {
#pragma unroll
        for (int k = 0; k < N; ++k)
            q[k] = fma_vvs(r[k], 0.2, -0.25);
#pragma unroll
        for (int k = 0; k < N; ++k)
            q[k] = fma_vvs(r[k], q[k], 1.0 / 3.0);
    } else if (DEG == 4) {
#pragma unroll
        for (int k = 0; k < N; ++k)
            q[k] = fma_vvs(r[k], -0.25, 1.0 / 3.0);
    }
    if (DEG == 3) {
#pragma unroll
        for (int k = 0; k < N; ++k) // (1/3 in a vector register pair, as -1/4 was for degree 4: VOP3 reads ONE scalar pair)
            q[k] = fma_vvs(r[k], 1.0 / 3.0, -(0.5 + 0.20710678118654752 * 0x1p-18));
    } else {
#pragma unroll
        for (int k = 0; k < N; ++k)
            q[k] = fma(r[k], q[k], -0.5);
    }
#pragma unroll
    for (int k = 0; k < N; ++k)
        lp[k] = fma(r[k] * r[k], q[k], r[k]);
#pragma unroll
    for (int k = 0; k < N; ++k)
        out[k] = fma((double)e[k], 0.693147180559945309417232121458, ent[k].y) + lp[k];
}

} // namespace covest
