// ll_scan.hip -- K-scan: the fast likelihood kernel of the REPEATS model on a dense grid
// (tail == 0).
//
// RepeatsModel.compute_probabilities (covest/models.py:211-242) is
//
//     p_j = sum_{o=1}^{T-1} b_o G[o][j],      G[o][j] = sum_s a_os TP(o l_s, j)   (depends on c, e only)
//     b_1 = q1,  b_2 = (1-q1) q2,  b_o = o_n rho^(o-3),  o_n = (1-q1)(1-q2) q,  rho = 1 - q   (:193-208)
//
// so every grid point with the same (c, e, q) needs the SAME running sum
//
//     S(T) = sum_{3 <= o < T} G[o][j] rho^(o-3)
//
// and differs only in where it stops (T = threshold_o depends on q1, q2 through o_n, :185-191) and
// in three scalars:   p_j = q1 G[1][j] + (1-q1) q2 G[2][j] + o_n S(T).
// A dense grid (covest/grid.py:39-43) has |q1| x |q2| such VARIANTS per q value.  K-factored
// (ll_factored.hip) contracts every variant separately on the fp64 matrix pipe -- 16 columns per MFMA,
// but on gfx950 an fp64 MFMA is no faster per flop than v_fma_f64 (tools/microbench_f64.hip), so the 16
// near-identical columns are 16x redundant work.  Here the sum is a plain per-lane scan, done once
// per (key, q) and read off at the variants' cut-offs:
//
//   phase A  lane = copy number o: the 8 error-class streams of streams.h walk the 32 keys of a tile
//            and store G[key][o] to LDS (as in K-factored), double-buffered.
//   phase B  a wave takes a UNIT (one q, <= 16 variants sorted by T; tiles.h ScanPlan); lane = (key,
//            half).  Half 0 sums o in [3, m), half 1 sums [m, T_max) and keeps its partial sum at each
//            variant's cut-off in a register array indexed by a scalar (s_set_gpr_idx); 2 fp64
//            instructions per (key, o): s += G w, w *= rho.  LDS reads are conflict-free (odd row
//            stride) and prefetched one block of 8 ahead.
//   phase C  the halves exchange sums (v_permlane32_swap / ds_bpermute) and each lane takes 8 of the
//            unit's 16 variants: p_j from three scalars of an LDS table, fast_log, h_j log p_j into
//            a per-(unit, variant) register.  All 64 lanes log together.
//
// Work per (c, e) and key tile, in fp64 issue slots: A 13.5 x 32 per builder wave; B 2 x sum_q
// (T_max(q) - m)/... ; C ~26 per (key, variant) / 64 -- on C3 about 6e3 against 10.5e3 for K-factored.
//
// Reference restated: covest/models.py:100-107 (LL), :211-242 (p_j), over covest/grid.py:59-64.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "fastmath.h"
#include "kernels.h"
#include "point_fetch.h"
#include "streams.h"
#include "wave.h"

namespace covest {

namespace {

typedef double d16 __attribute__((ext_vector_type(16)));
typedef double d8 __attribute__((ext_vector_type(8)));

constexpr int kNT = kScanWaves * kWave;
constexpr int kBlock = 8; // scan steps per prefetched block
constexpr bool kFastBlocks = false; // two-accumulator blocks where nothing is read off: measured, costs registers (spills)
constexpr int kLogGroup = 4; // logs in flight per lane (registers: two waves per SIMD leave 256 VGPRs)
constexpr double kTinyP = 0x1p-960; // below this, p_j is re-summed with the reference's per-term rounding

// v_permlane32_swap: lanes 32..63 of `x` trade places with lanes 0..31 of `y` (pure VALU, no LDS)
__device__ __forceinline__ void swap_halves(double &x, double &y)
{
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
    x = __hiloint2double((int)hi[0], (int)lo[0]);
    y = __hiloint2double((int)hi[1], (int)lo[1]);
}

// sum over the 32 lanes of this lane's half (every lane gets it)
__device__ __forceinline__ double half_sum(double x)
{
#pragma unroll
    for (int off = 1; off < 32; off <<= 1)
        x += __shfl_xor(x, off, kWave);
    return x;
}

__global__ __launch_bounds__(kNT) void ll_scan_kernel(const DevModel m, const int32_t n_tiles,
                                                      const double *__restrict__ tile_dbl,
                                                      const int32_t *__restrict__ tile_int, const ScanPlan plan,
                                                      const int32_t *__restrict__ unit_m,
                                                      const int32_t *__restrict__ unit_cut,
                                                      const double *__restrict__ unit_rho,
                                                      double *__restrict__ out_ll)
{
    const TileView tv = tile_view_from(n_tiles, tile_dbl, tile_int);
    constexpr int NU = kScanUnitsPerWave;
    constexpr int NV = kScanVariants;
    const int LD = plan.ld;
    extern __shared__ double Gs[]; // [n_buf][kTileBins][LD] + 64 slack + coefficient table
    __shared__ __attribute__((aligned(16))) double log_tab[kLogTableDoubles];
    load_log_table(log_tab);

    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int key = lane & 31; // row of the key tile
    const int half = lane >> 5;

    // ---- the (c, e) of this workgroup ----
    const int64_t ce = plan.ce_begin + blockIdx.x;
    const int64_t ic = ce / plan.n_e;
    const int64_t ie = ce - ic * plan.n_e;
    double par[kMaxParams] = {plan.c_axis[ic], plan.e_axis[ie], 0, 0, 0};
    clamp_point<2>(m, par);
    const bool finite = isfinite(par[0]) && isfinite(par[1]);
    if (tid < 8)
        Gs[tid] = error_class_rate(m, par[0], par[1], tid);
    __syncthreads();
    double lam[8];
#pragma unroll
    for (int s = 0; s < 8; ++s)
        lam[s] = Gs[s];
    __syncthreads();

    // ---- LDS layout ----
    const size_t g_doubles = (size_t)plan.n_buf * kTileBins * LD;
    double *coef = Gs + g_doubles + 64; // [16 slots of this workgroup][16 variants][4]
    const int slot0 = (int)blockIdx.y * (kScanWaves * NU);
    for (int i = tid; i < kScanWaves * NU * NV * 4; i += kNT)
        coef[i] = plan.var_coef[(int64_t)slot0 * NV * 4 + i];
    if (tid < 64)
        Gs[g_doubles + tid] = 0.0; // slack behind the buffers: prefetches run past a row's end
    if (tid < plan.n_buf * kTileBins) { // the pad columns of every row
        Gs[(size_t)tid * LD + LD - 2] = 0.0;
        Gs[(size_t)tid * LD + LD - 1] = 0.0;
    }

    // ---- phase-A state: lane = copy number o = tid + 1 ----
    const bool wave_builds = wave * kWave < plan.max_o; // wave-uniform
    StreamSet<8> st;
    st.init(m, lam, tid + 1, finite && wave_builds && (tid + 1) <= plan.max_o);
    double xx[8];
    st.squares(xx);
    const bool lane_in_row = tid < LD - 2;

    auto build_tile = [&](int t, double *dst) {
        if (plan.skip_phases & 1)
            return;
        const double k0 = tv.first_key[t];
        const int nb = tv.n_bins[t];
        st.enter_tile(k0 - 1.0, k0 + (double)(nb - 1), tv.lgam_prev[t], tv.lgam_last[t],
                      tv.run_start[t] != 0);
        const double *scal = tv.scal + (int64_t)t * kTileBins;
        double *colp = dst + (lane_in_row ? tid : 0);
        if (nb == kTileBins) { // the common case: straight-line code, scales in SGPRs
#pragma unroll
            for (int b = 0; b < kTileBins; b += 2) {
                double g1, g2;
                st.step2(xx, g1, g2);
                g1 *= scal[b];
                g2 *= scal[b + 1];
                if (lane_in_row) {
                    colp[b * LD] = g1;
                    colp[(b + 1) * LD] = g2;
                }
            }
        } else {
            for (int b = 0; b < kTileBins; ++b) {
                const double g = b < nb ? st.step() * scal[b] : 0.0;
                if (lane_in_row)
                    colp[b * LD] = g;
            }
        }
        st.leave_tile(tv.renorm[t]);
    };

    // ---- phase-B/C state: this wave's units ----
    int um[NU];        // first o of the upper half (-1: empty slot), wave-uniform
    int len0[NU];      // steps of the lower half: m - 3
    int cutv[NU];      // lane v < 16 holds cut v of the unit (read with v_readlane)
    double rho[NU], rho_m[NU];
    int col0[NU];      // this lane's first column inside a G row (column = o - 1)
    double ll[NU][NV / 2];
    uint64_t dead[NU][NV / 2]; // lanes that met p_j <= 0 with h_j != 0
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int slot = slot0 + wave * NU + u;
        um[u] = __builtin_amdgcn_readfirstlane(unit_m[slot]);
        len0[u] = um[u] - 3;
        cutv[u] = unit_cut[(int64_t)slot * NV + (lane & (NV - 1))];
        rho[u] = unit_rho[2 * slot];
        rho_m[u] = unit_rho[2 * slot + 1];
        col0[u] = half ? um[u] - 1 : 2;
#pragma unroll
        for (int k = 0; k < NV / 2; ++k) {
            ll[u][k] = 0.0;
            dead[u][k] = 0;
        }
    }

    // in-kernel stamps (diagnostic runs only): cycles per wave in build / scan / log / barrier
    long long dg_a = 0, dg_b = 0, dg_c = 0, dg_w = 0, dg_t0 = 0;
    const bool diag = plan.diag != nullptr;
#define STAMP(acc)                                    \
    if (diag) {                                       \
        const long long now__ = (long long)clock64(); \
        acc += now__ - dg_t0;                         \
        dg_t0 = now__;                                \
    }
    if (diag)
        dg_t0 = (long long)clock64();

    const bool dbuf = plan.n_buf == 2;
    if (dbuf) {
        if (wave_builds)
            build_tile(0, Gs);
        __syncthreads();
    }
    for (int t = 0; t < tv.n_tiles; ++t) {
        const double *cur = Gs + (dbuf ? (t & 1) * kTileBins * LD : 0);
        if (!dbuf) {
            if (wave_builds)
                build_tile(t, Gs);
            __syncthreads();
        } else if (wave_builds && t + 1 < tv.n_tiles) {
            build_tile(t + 1, Gs + ((t + 1) & 1) * kTileBins * LD);
        }
        STAMP(dg_a)
        const double hj = tv.cnt[(int64_t)t * kTileBins + key];
        const double *row = cur + key * LD;
        const double g1 = row[0], g2 = row[1]; // G[1][key], G[2][key]

#pragma unroll
        for (int u = 0; u < NU; ++u) {
            if (um[u] < 0) // wave-uniform
                continue;
            // ============ phase B: the running sum over o ============
            // Blocks of 8 steps, branch-free inside: the partial sum after every step of the block stays
            // in registers (pre[0..7], s), and the cut-offs that fall into the block pick theirs with a
            // scalar register index afterwards.  A taken scalar branch costs this pipeline about as much
            // as a dozen fp64 instructions, so there is one loop branch per block and one per cut-off.
            const double *p0 = row + col0[u];
            const double r = rho[u], r2 = r * r;
            double s = 0.0, w = 1.0;
            d16 snap;
#pragma unroll
            for (int v = 0; v < NV; ++v)
                snap[v] = 0.0;
            int nv = 0;
            int next_cut = __builtin_amdgcn_readlane(cutv[u], 0);
            const int len = (plan.skip_phases & 2) ? 0 : __builtin_amdgcn_readlane(cutv[u], NV - 1);
            double s0 = 0.0;
            int s0_at = len0[u]; // step at which the lower half's sum is complete (then INT_MAX)

            // one block: consumes g[0..7] (steps base .. base + 7) and refills each register with the
            // same step of the next block as soon as it is free (the loads fly during the rest of the
            // block and the read-offs)
            double g[kBlock];
#pragma unroll
            for (int k = 0; k < kBlock; ++k)
                g[k] = p0[k];
            for (int base = 0; base < len; base += kBlock) {
                const int stop = base + kBlock;
                const double *pn = p0 + stop;
                if (kFastBlocks && min(s0_at, next_cut) > stop) {
                    // nothing to read off inside this block: two independent partial sums
                    double sb = g[1] * (w * r);
                    s = fma(g[0], w, s);
                    g[0] = pn[0];
                    g[1] = pn[1];
                    w *= r2;
#pragma unroll
                    for (int k = 2; k < kBlock; k += 2) {
                        s = fma(g[k], w, s);
                        sb = fma(g[k + 1], w * r, sb);
                        g[k] = pn[k];
                        g[k + 1] = pn[k + 1];
                        w *= r2;
                    }
                    s += sb;
                    continue;
                }
                d8 pre;
#pragma unroll
                for (int k = 0; k < kBlock; ++k) {
                    pre[k] = s; // sum over the steps before base + k
                    s = fma(g[k], w, s);
                    g[k] = pn[k];
                    w *= r;
                }
                if (s0_at <= stop) {
                    const int at = s0_at - base;
                    const double inside = pre[at & (kBlock - 1)];
                    s0 = at == kBlock ? s : inside;
                    s0_at = 0x7fffffff;
                }
                while (next_cut <= stop) { // ends: the cut-off after the last one is INT_MAX
                    const int at = next_cut - base;
                    const double inside = pre[at & (kBlock - 1)];
                    snap[nv] = at == kBlock ? s : inside;
                    ++nv;
                    next_cut = nv < NV ? __builtin_amdgcn_readlane(cutv[u], nv) : 0x7fffffff;
                }
            }
            // (len == 0: every sum is the empty one, snap and s0 are already 0)

            STAMP(dg_b)
            // ============ phase C: exchange, p_j, log ============
            // half 0 holds s0 = S over [3, m); half 1 holds snap[v] = sum over [m, T_v) with weights
            // relative to m.  Lane (key, half) takes variants 8 half .. 8 half + 7.
            double base_sum = s0, junk = s0;
            swap_halves(base_sum, junk); // base_sum: lanes 32..63 now hold the lower half's s0 too
            const double *cf = coef + ((wave * NU + u) * NV + half * (NV / 2)) * 4;
            const double rm = rho_m[u];
            // Four variants at a time: p_j, then their (independent, interleaved) logs.  Near the bottom
            // of the double range the REFERENCE's p_j is a sum of products b_o * G that were each rounded
            // to a subnormal (covest/models.py:236-240); the factored sum rounds once and would differ
            // from it by whole subnormal ulps.  Those keys (a handful per grid, h_j of 1..3) are summed
            // again term by term.  The same test catches p_j <= 0 (utils.safe_log: the whole sum is -inf;
            // kept as a lane mask).  `if h` of covest/models.py:106: filler keys (h_j == 0) are skipped.
#pragma unroll
            for (int k0 = 0; k0 < NV / 2; k0 += kLogGroup) {
                double pk[kLogGroup];
                bool any_low = false;
#pragma unroll
                for (int i = 0; i < kLogGroup; ++i) {
                    const int k = k0 + i;
                    // after the swap `up` holds the upper half's snapshot of THIS lane's variant in both
                    // halves: lanes 0..31 receive snap[k] of lanes 32..63, lanes 32..63 keep their snap[k + 8]
                    double lo_part = snap[k], up = snap[k + NV / 2];
                    swap_halves(lo_part, up);
                    const double S = fma(rm, up, base_sum);
                    const double2 b12 = *reinterpret_cast<const double2 *>(cf + 4 * k);
                    const double on = cf[4 * k + 2];
                    pk[i] = fma(on, S, fma(b12.y, g2, b12.x * g1));
                    any_low |= pk[i] < kTinyP;
                }
                if (__ballot(any_low && hj != 0.0)) { // wave-uniform, rare
#pragma unroll
                    for (int i = 0; i < kLogGroup; ++i) {
                        const int k = k0 + i;
                        const bool low = pk[i] < kTinyP && hj != 0.0;
                        if (!__ballot(low))
                            continue;
                        // (all lanes take part in the shuffle: the cut-offs live in lanes 0..15)
                        const int t_end = __shfl(cutv[u], half * (NV / 2) + k, kWave) + um[u]; // max(T, 3)
                        if (low) {
                            const double2 b12 = *reinterpret_cast<const double2 *>(cf + 4 * k);
                            const double on = cf[4 * k + 2];
                            double acc = __dadd_rn(__dmul_rn(b12.x, g1), __dmul_rn(b12.y, g2));
                            double pw = 1.0;
                            for (int o = 3; o < t_end; ++o) {
                                acc = __dadd_rn(acc, __dmul_rn(__dmul_rn(on, pw), row[o - 1]));
                                pw = __dmul_rn(pw, r);
                            }
                            pk[i] = acc;
                        }
                        dead[u][k] |= __ballot(low && pk[i] <= 0.0);
                    }
                }
                if (!(plan.skip_phases & 4)) {
#pragma unroll
                    for (int i = 0; i < kLogGroup; ++i)
                        ll[u][k0 + i] = fma(hj, fast_log(pk[i], log_tab), ll[u][k0 + i]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            STAMP(dg_c)
        }
        __syncthreads(); // the tile just read may be overwritten, the one just built may be read
        STAMP(dg_w)
    }
    if (diag && lane == 0) {
        long long *d = plan.diag + ((int64_t)(blockIdx.x * gridDim.y + blockIdx.y) * kScanWaves + wave) * 8;
        d[0] = dg_a;
        d[1] = dg_b;
        d[2] = dg_c;
        d[3] = dg_w;
    }
#undef STAMP

    // ---- per-(unit, variant) results: sum over the 32 keys of the half ----
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        if (um[u] < 0)
            continue;
        const int slot = slot0 + wave * NU + u;
#pragma unroll
        for (int k = 0; k < NV / 2; ++k) {
            double v = ll[u][k];
            if ((dead[u][k] >> lane) & 1)
                v = isnan(v) ? v : -INFINITY; // h * -inf summed with finite terms
            v = half_sum(v);
            if (key == 0) {
                const int32_t qo = plan.var_orig[(int64_t)slot * NV + half * (NV / 2) + k];
                if (qo >= 0) {
                    const int64_t flat = ce * plan.n_q + qo;
                    if (flat >= plan.flat_begin && flat < plan.flat_end)
                        out_ll[flat - plan.flat_begin] = finite ? v : NAN;
                }
            }
        }
    }
}

} // namespace

hipError_t launch_ll_scan(const DevModel &m, const TileView &tv, const ScanPlan &plan, double *out_ll,
                          hipStream_t stream)
{
    if (plan.ce_end <= plan.ce_begin)
        return hipSuccess;
    if (m.n_err != 8 || m.kind != 1 || plan.max_o > kNT || m.tail != 0.0)
        return hipErrorInvalidValue;
    const size_t lds = ((size_t)plan.n_buf * kTileBins * plan.ld + 64 +
                        (size_t)kScanWaves * kScanUnitsPerWave * kScanVariants * 4) * sizeof(double);
    static size_t configured[64] = {0}; // the dynamic-LDS ceiling is a per-device attribute of the kernel
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64)
        dev = 0;
    if (lds > configured[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&ll_scan_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess)
            return e;
        configured[dev] = lds;
    }
    // HIP wraps a grid of more than 2^32 threads silently: at most 2^22 workgroups per launch
    const int64_t per_launch = std::max<int64_t>(1, ((int64_t)1 << 22) / plan.n_qblocks);
    for (int64_t first = plan.ce_begin; first < plan.ce_end; first += per_launch) {
        ScanPlan part = plan;
        part.ce_begin = first;
        part.ce_end = std::min(plan.ce_end, first + per_launch);
        const dim3 grid((unsigned)(part.ce_end - part.ce_begin), (unsigned)plan.n_qblocks);
        hipLaunchKernelGGL(ll_scan_kernel, grid, dim3(kNT), lds, stream, m, tv.n_tiles, tv.dbl_base, tv.int_base,
                           part, plan.unit_m, plan.unit_cut, plan.unit_rho, out_ll);
    }
    return hipGetLastError();
}

} // namespace covest
