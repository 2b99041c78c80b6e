// wave.h -- wave64 helpers for gfx950 (CDNA4).  A wavefront is 64 lanes; nothing
// here is written for, or tested on, 32-wide hardware.
#pragma once
#include <hip/hip_runtime.h>

namespace covest {

constexpr int kWave = 64;

// Broadcast lane `src` (wave-uniform) of a double to every lane through the
// scalar unit: two v_readlane_b32, no LDS crossbar traffic.
__device__ __forceinline__ double wave_bcast(double v, int src)
{
    int lo = __double2loint(v);
    int hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, src);
    hi = __builtin_amdgcn_readlane(hi, src);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
        v += __shfl_xor(v, off, kWave);
    return v;
}

// Error-free transformation: a + b = s + e exactly (Knuth two-sum, 6 flops).
__device__ __forceinline__ void two_sum(double a, double b, double &s, double &e)
{
    s = a + b;
    const double bb = s - a;
    e = (a - (s - bb)) + (b - bb);
}

// Compensated accumulator (hi + lo): the stand-in for math.fsum at
// covest/models.py:103.  Adding n non-negative terms leaves an error of O(eps^2 n),
// i.e. hi + lo is the correctly rounded sum for every n that occurs here.
struct CompSum {
    double hi, lo;
    __device__ __forceinline__ void add(double x)
    {
        double e;
        two_sum(hi, x, hi, e);
        lo += e;
    }
};

__device__ __forceinline__ double wave_comp_sum(CompSum v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const double ohi = __shfl_xor(v.hi, off, kWave);
        const double olo = __shfl_xor(v.lo, off, kWave);
        double e;
        two_sum(v.hi, ohi, v.hi, e);
        v.lo += olo + e;
    }
    return v.hi + v.lo;
}

} // namespace covest
