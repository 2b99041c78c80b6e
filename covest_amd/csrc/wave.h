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

} // namespace covest
