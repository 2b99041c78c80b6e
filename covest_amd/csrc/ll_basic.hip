// ll_basic.hip -- K-basic: the fast likelihood kernel of the BASIC model.
//
// One LANE per grid point: a wave64 evaluates 64 consecutive points of the flat
// grid order (or of a point list).  Every lane walks the histogram keys in
// ascending order with the pmf recurrence of streams.h (S error-class streams in
// registers, 1.5 fp64 instructions per pmf term on full tiles), so the key, its count h_j and
// the per-key scale are WAVE-UNIFORM: they come from the tile table through the
// scalar cache into SGPRs, LDS holds only the 1 KB log table and the per-lane anchor constants
// (read once per tile), there is no cross-lane operation and
// no divergence (neighbouring lanes differ only in (c, e)).  One log per
// (point, non-zero bin) -- the dominant cost of this kernel once the terms are
// down to 2 instructions.
//
// Reference restated: BasicModel.compute_probabilities / compute_loglikelihood,
// covest/models.py:81-107, over the grid of covest/grid.py:59-64.
//
// Roofline: fp64 VALU.  Per point and key: 12.5 instructions for the 8 streams + 18 for the
// log and its accumulation (30.75 measured in the ISA; tools/microbench_ops.hip prices each).
// Algorithmic HBM bytes: 16 in (or the two axes) + 8 out per point.
#include <hip/hip_runtime.h>

#include "direct_point.h"
#include "fastmath.h"
#include "kernels.h"
#include "point_fetch.h"
#include "streams.h"
#include "wave.h"

namespace covest {

namespace {

template <int S, bool TAIL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void ll_basic_kernel(const DevModel m, const int32_t n_tiles,
                                                       const double *__restrict__ tile_dbl,
                                                       const int32_t *__restrict__ tile_int,
                                                       const PointSource src, const int64_t n,
                                                       double *__restrict__ out_ll)
{
    const TileView tv = tile_view_from(n_tiles, tile_dbl, tile_int);
    __shared__ __attribute__((aligned(16))) double log_tab[kLogTableDoubles];
    load_log_table(log_tab);
    __syncthreads();
    const int64_t pt = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = pt < n;
    const int64_t ptc = live ? pt : n - 1; // idle lanes shadow the last point: every lane stays in the wave ops

    double par[kMaxParams];
    int T;
    fetch_point<2>(src, ptc, par, T);
    clamp_point<2>(m, par);
    const bool finite = isfinite(par[0]) && isfinite(par[1]);

    double lam[S];
#pragma unroll
    for (int s = 0; s < S; ++s)
        lam[s] = error_class_rate(m, par[0], par[1], s);
    __shared__ double anchors[2 * S * 256]; // [2S][lane of the workgroup]: conflict-free columns
    StreamSet<S, LdsAnchors<S>> st;
    st.an.mine = anchors + threadIdx.x;
    st.an.stride = 256;
    st.init(m, lam, 1, finite, log_tab, log_tab);

    double acc_ll = 0.0;
    uint64_t dead = 0; // lanes that met a p_j <= 0 with h_j != 0
    uint64_t tiny = 0; // lanes that met a p_j below the normal range with h_j != 0 (p_j <= 0 included)
    CompSum acc_sp = {0.0, 0.0};

    // one key's p_j (flushed like the reference's double): into sp_j, and its log (all branches
    // wave-uniform)
    auto account = [&](double p, double h) {
        if (TAIL)
            acc_sp.add(p); // (filler keys have scale 0: p == 0)
        if (h != 0.0) { // filler keys and zero counts: no log (`if h`, covest/models.py:106)
            // utils.safe_log: p_j <= 0 makes the whole sum -inf.  Remembered as a lane mask in
            // SGPRs (one compare) instead of a select per key; fast_log(0) is finite.
            dead |= __ballot(p <= 0.0);
            // a SUBNORMAL p_j: the reference's value hangs on the rounding of every single term onto the
            // 4.9e-324 grid -- the point is handed to the term-by-term kernel
            tiny |= __ballot(p < kMinNormal);
            acc_ll = fma(h, fast_log(p, log_tab), acc_ll);
        }
    };

    for (int t = 0; t < tv.n_tiles; ++t) {
        const double k0 = tv.first_key[t];
        const int nb = tv.n_bins[t];
        st.enter_tile(k0 - 1.0, k0 + (double)(nb - 1), tv.lgam_prev[t], tv.lgam_last[t],
                      tv.run_start[t] != 0);
        const double *scal = tv.scal + (int64_t)t * kTileBins;
        const double *cnt = tv.cnt + (int64_t)t * kTileBins;
        if (nb == kTileBins) {
            double xx[S]; // squared rates, recomputed per tile (S multiplies) rather than held in 2 S registers
            st.squares(xx);
            // full tile: two straight-line halves of 16 keys, their scales and counts fetched
            // into SGPRs up front (s_load_dwordx16) so no key waits on the scalar cache; the
            // streams advance two keys per step (streams.h step2)
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                double sc[16], hc[16];
#pragma unroll
                for (int b = 0; b < 16; ++b) {
                    sc[b] = scal[16 * half + b];
                    hc[b] = cnt[16 * half + b];
                }
#pragma unroll
                for (int b = 0; b < 16; b += 2) {
                    double g1, g2;
                    st.step2(xx, g1, g2);
                    account(g1 * sc[b], hc[b]);
                    account(g2 * sc[b + 1], hc[b + 1]);
                }
            }
        } else {
            for (int b = 0; b < nb; ++b)
                account(st.step() * scal[b], cnt[b]);
        }
        st.leave_tile(tv.renorm[t]);
    }

    double tail_term = 0.0;
    if (TAIL) {
        double sp = acc_sp.hi + acc_sp.lo;
        if (!(sp < 1.0))
            sp = 1.0;
        if (sp < 1.0)
            tail_term = m.tail * log(1.0 - sp);
    }
    if ((dead >> (threadIdx.x & (kWave - 1))) & 1)
        acc_ll = isnan(acc_ll) ? acc_ll : -INFINITY; // h * -inf summed with finite terms
    double ll = acc_ll + tail_term;
    if (((tiny & ~dead) >> (threadIdx.x & (kWave - 1))) & 1)
        ll = isfinite(ll) ? redo_marker() : ll; // replaced by K-direct's value before anyone sees it
    if (!finite)
        ll = NAN; // a NaN parameter poisons every p_j in the reference
    if (live)
        out_ll[pt] = ll;
}

} // namespace

hipError_t launch_ll_basic(const DevModel &m, const TileView &tv, const PointSource &src, int64_t n,
                           double *out_ll, hipStream_t stream)
{
    if (n <= 0)
        return hipSuccess;
    if (m.n_err != 8 || m.kind != 0)
        return hipErrorInvalidValue;
    // HIP wraps a grid of more than 2^32 threads silently: at most 2^23 workgroups per launch
    const dim3 block(256);
    const int64_t per_launch = (int64_t)256 << 23;
    for (int64_t first = 0; first < n; first += per_launch) {
        const int64_t cnt = n - first < per_launch ? n - first : per_launch;
        const dim3 grid((unsigned)((cnt + 255) / 256));
        PointSource part = src;
        if (src.is_grid)
            part.flat_begin = src.flat_begin + first;
        else
            part.params = src.params + first * 2;
        if (m.tail != 0.0)
            hipLaunchKernelGGL((ll_basic_kernel<8, true>), grid, block, 0, stream, m, tv.n_tiles, tv.dbl_base,
                               tv.int_base, part, cnt, out_ll + first);
        else
            hipLaunchKernelGGL((ll_basic_kernel<8, false>), grid, block, 0, stream, m, tv.n_tiles, tv.dbl_base,
                               tv.int_base, part, cnt, out_ll + first);
    }
    return hipGetLastError();
}

} // namespace covest
