// argmin.hip -- K-argmin: (min -LL, lowest flat index attaining it) over the LL
// buffer of one GPU's block of the grid.
//
// Restates the selection scan of covest/grid.py:65-70 (maximize=False) started
// from +inf:  `if val < min_val` is STRICT, so the lowest index wins ties, NaN
// never wins, +inf (LL = -inf) never wins, -inf (LL = +inf) does.
//
// HBM-bound streaming read of 8 bytes per grid point, two launches: a
// grid-stride pass of at most 256 workgroups (one per CU) that leaves one
// candidate per workgroup, then one workgroup over the candidates.  No
// atomics, so the result is deterministic.  The first pass also re-evaluates
// the points a recurrence kernel handed back (redo marker, direct_point.h).
#include <hip/hip_runtime.h>

#include <algorithm>

#include "direct_point.h"
#include "kernels.h"
#include "wave.h"

namespace covest {

namespace {

struct Cand {
    double v;
    int64_t i;
};

__device__ __forceinline__ Cand better(Cand a, Cand b)
{
    // b replaces a iff b is strictly smaller, or equal with a lower index
    const bool take = (b.v < a.v) || (b.v == a.v && b.i < a.i);
    return take ? b : a;
}

__device__ __forceinline__ Cand wave_best(Cand c)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        Cand o;
        o.v = __shfl_xor(c.v, off, kWave);
        o.i = __shfl_xor(c.i, off, kWave);
        c = better(c, o);
    }
    return c;
}

__device__ __forceinline__ Cand block_best(Cand c)
{
    __shared__ double sv[16];
    __shared__ int64_t si[16];
    c = wave_best(c);
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
    if (lane == 0) {
        sv[w] = c.v;
        si[w] = c.i;
    }
    __syncthreads();
    const int n_w = blockDim.x / kWave;
    Cand r;
    r.v = (threadIdx.x < n_w) ? sv[threadIdx.x] : INFINITY;
    r.i = (threadIdx.x < n_w) ? si[threadIdx.x] : INT64_MAX;
    return wave_best(r); // valid in wave 0
}

// First stage, fused with the hand-back of the recurrence kernels: a point whose value is the
// redo marker (direct_point.h: a key with h_j != 0 has a subnormal p_j there) is evaluated again,
// term by term, by the whole wave that meets it, and the LL buffer is patched in place.  Such points
// are rare (a model that gives probability 1e-310 to a key that was observed); without any, the
// pass costs one compare per point on top of the 8-byte read.
template <int P>
__global__ __launch_bounds__(256) void argmin_stage1(const DevModel m, const PointSource src, double *__restrict__ ll,
                                                     int64_t n, double *__restrict__ pv, int64_t *__restrict__ pi)
{
    Cand c;
    c.v = INFINITY;
    c.i = INT64_MAX;
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    // wave-uniform trip count: lanes past the end hold +inf
    for (int64_t base = (int64_t)blockIdx.x * blockDim.x + (threadIdx.x - lane); base < n; base += stride) {
        const int64_t i = base + lane;
        double val = i < n ? ll[i] : -INFINITY;
        uint64_t redo = __ballot(i < n && is_redo_marker(val));
        while (redo) { // wave-uniform
            const int who = __builtin_ctzll(redo);
            redo &= redo - 1;
            const double again = direct_point_ll<P, false>(m, src, base + who, nullptr);
            if (lane == who) {
                val = again;
                ll[i] = again;
            }
        }
        const double v = -val;
        // ascending i within a thread: strict < keeps the first occurrence
        if (v < c.v) {
            c.v = v;
            c.i = i;
        }
    }
    c = block_best(c);
    if (threadIdx.x == 0) {
        pv[blockIdx.x] = c.v;
        pi[blockIdx.x] = c.i;
    }
}

__global__ __launch_bounds__(256) void argmin_stage2(const double *__restrict__ pv,
                                                     const int64_t *__restrict__ pi, int n_part, int64_t flat_begin,
                                                     ArgminResult *__restrict__ result)
{
    Cand c;
    c.v = INFINITY;
    c.i = INT64_MAX;
    for (int i = threadIdx.x; i < n_part; i += blockDim.x) {
        Cand o;
        o.v = pv[i];
        o.i = pi[i];
        c = better(c, o);
    }
    c = block_best(c);
    if (threadIdx.x == 0) {
        result->min_negll = c.v;
        result->index = (c.i == INT64_MAX) ? -1 : c.i;
        // the same as a pair of doubles with the GLOBAL flat index, for the cross-GPU exchange (flat indices
        // stay below 2^53)
        result->pair[0] = c.v;
        result->pair[1] = (c.i == INT64_MAX) ? -1.0 : (double)(flat_begin + c.i);
    }
}

} // namespace

hipError_t launch_argmin(const DevModel &m, const PointSource &src, double *ll, int64_t n, int64_t flat_begin,
                         double *partial_val, int64_t *partial_idx, ArgminResult *result, hipStream_t stream)
{
    // (a small grid needs no more workgroups than it has waves of points)
    const int blocks = (int)std::min<int64_t>(kArgminBlocks, std::max<int64_t>(1, (n + 255) / 256));
    if (m.kind == 0)
        hipLaunchKernelGGL(argmin_stage1<2>, dim3(blocks), dim3(256), 0, stream, m, src, ll, n, partial_val,
                           partial_idx);
    else
        hipLaunchKernelGGL(argmin_stage1<5>, dim3(blocks), dim3(256), 0, stream, m, src, ll, n, partial_val,
                           partial_idx);
    hipLaunchKernelGGL(argmin_stage2, dim3(1), dim3(256), 0, stream, partial_val, partial_idx, blocks, flat_begin, result);
    return hipGetLastError();
}

} // namespace covest
