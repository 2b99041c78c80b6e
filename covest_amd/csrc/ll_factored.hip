// ll_factored.hip -- K-factored: the fast likelihood kernel of the REPEATS model on
// a dense grid.
//
// In RepeatsModel.compute_probabilities (covest/models.py:211-242)
//
//     p_j = sum_{o=1}^{T-1} b_o(q1,q2,q) * G[o][j],   G[o][j] = sum_s a_os * TP(o*l_s, j)
//
// everything expensive -- G -- depends only on (c, e); the other three parameters
// enter through the weights b_o and the cut-off T (models.py:185-208).  A dense
// grid (covest/grid.py:39-43: itertools.product of the axes) evaluates every
// (c, e) with the same Q = |q1| x |q2| x |q| weight vectors, so one workgroup
// takes one (c, e) and all Q of them:
//
//   phase A  lane = copy number o: the 8 error-class streams of streams.h walk a
//            tile of 32 keys (2 fp64 instr per pmf term) and store G[key][o] to LDS
//   phase B  P[key][q] = sum_o G[key][o] * b_o(q) on the fp64 matrix pipe:
//            v_mfma_f64_16x16x4_f64, A = 16 keys x 4 o from LDS (conflict-free
//            ds_read_b64, row stride = 4 dwords mod 64), B = 4 o x 16 q generated
//            IN REGISTERS: b_{o+4} = b_o (1-q)^4 is one multiply per MFMA pair, the
//            first 8 weights and the cut-off T come from the host (libm pow, as
//            CPython) -- so the contraction reads no weights from memory at all
//   phase C  h_j * log P[key][q] straight from the accumulator registers: in the
//            f64 C/D layout a lane keeps ONE q column, so the running LL of a
//            q-tile is a single register per lane
//
// q points are sorted by T (descending) and dealt to the waves in tiles of 16, so
// a wave's o-loop stops at its own tile's T.  gfx950 measured (tools/
// microbench_f64.hip): v_fma_f64 62 TFLOP/s, v_mfma_f64_16x16x4 75 TFLOP/s, and
// the two do NOT overlap (same fp64 datapath), so the phases are simply
// sequential and the kernel is bound by the fp64 pipe:
//   flops per (c,e) = B*8*(Tmax-1)*2  +  B*sum_q(T_q-1)*2  +  B*Q*(one log)
// against B*8*sum_q(T_q-1)*4 for the per-point formulation of SURVEY 8(d).
//
// Reference restated: covest/models.py:100-107 (LL), :211-242 (p_j), over
// covest/grid.py:59-64 (the grid map).
#include <hip/hip_runtime.h>

#include "fastmath.h"
#include "kernels.h"
#include "point_fetch.h"
#include "streams.h"
#include "wave.h"

namespace covest {

namespace {

typedef double d4 __attribute__((ext_vector_type(4)));

template <int NT, bool TAIL>
__global__ __launch_bounds__(NT) void ll_factored_kernel(const DevModel m, const TileView tv,
                                                         const FactoredPlan plan,
                                                         double *__restrict__ out_ll)
{
    constexpr int NW = NT / kWave;
    constexpr int LD = NT + 2; // G row stride in doubles: 2*NT + 4 dwords = 4 (mod 64) -> conflict-free A reads
    extern __shared__ double Gs[]; // [kTileBins][LD]
    __shared__ __attribute__((aligned(16))) double log_tab[64];
    load_log_table(log_tab);

    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = tid / kWave;

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

    // ---- phase-A state: lane = copy number o = tid + 1 ----
    StreamSet<8> st;
    st.init(m, lam, tid + 1, finite && (tid + 1) <= plan.max_o);

    // ---- phase-B/C state: this wave's q-tiles (interleaved over waves and q-blocks) ----
    const int col = lane & 15; // q column inside a tile / key row of the A fragment
    const int kq = lane >> 4;  // which of the 4 o of an MFMA step
    int nsteps[kMaxQTiles], tq[kMaxQTiles], qslot[kMaxQTiles];
    double r4[kMaxQTiles], llacc[kMaxQTiles];
    CompSum spacc[kMaxQTiles];
    const int n_slots = plan.n_qtiles * 16;
#pragma unroll
    for (int i = 0; i < kMaxQTiles; ++i) {
        const int qt = __builtin_amdgcn_readfirstlane((i * NW + wave) * (int)gridDim.y + (int)blockIdx.y);
        const bool on = qt < plan.n_qtiles;
        const int slot = (on ? qt : 0) * 16 + col;
        qslot[i] = on ? slot : -1;
        nsteps[i] = __builtin_amdgcn_readfirstlane(on ? plan.qtile_nsteps[qt] : 0);
        tq[i] = on ? plan.q_T[slot] : 0;
        r4[i] = plan.q_r4[slot];
        llacc[i] = 0.0;
        spacc[i].hi = 0.0;
        spacc[i].lo = 0.0;
    }
    const int max_steps = nsteps[0]; // tiles are sorted by T: this wave's first tile is its longest

    for (int t = 0; t < tv.n_tiles; ++t) {
        // ================= phase A: G[key][o] for 32 keys =================
        const double k0 = tv.first_key[t];
        const int nb = tv.n_bins[t];
        st.enter_tile(k0 - 1.0, k0 + (double)(nb - 1), tv.lgam_prev[t], tv.lgam_last[t],
                      tv.run_start[t] != 0);
        const double *scal = tv.scal + (int64_t)t * kTileBins;
        for (int b = 0; b < nb; ++b)
            Gs[b * LD + tid] = st.step() * scal[b];
        for (int b = nb; b < kTileBins; ++b)
            Gs[b * LD + tid] = 0.0;
        st.leave_tile(tv.renorm[t]);
        __syncthreads();

        // ================= phase B: P = G x b on the matrix pipe =================
        d4 acc[kMaxQTiles][2];
        double wfirst[kMaxQTiles], wrun[kMaxQTiles]; // b_o for o = 1+kq and 5+kq (L1-resident, reloaded per tile)
#pragma unroll
        for (int i = 0; i < kMaxQTiles; ++i) {
            acc[i][0] = (d4){0.0, 0.0, 0.0, 0.0};
            acc[i][1] = (d4){0.0, 0.0, 0.0, 0.0};
            const int slot = qslot[i] >= 0 ? qslot[i] : col;
            wfirst[i] = plan.q_first8[(int64_t)kq * n_slots + slot];
            wrun[i] = plan.q_first8[(int64_t)(4 + kq) * n_slots + slot];
        }
        const double *arow0 = Gs + col * LD + kq;
        const double *arow1 = Gs + (16 + col) * LD + kq;
        for (int step = 0; step < max_steps; ++step) {
            const double a0 = arow0[4 * step];
            const double a1 = arow1[4 * step];
            const int o_here = 1 + 4 * step + kq;
#pragma unroll
            for (int i = 0; i < kMaxQTiles; ++i) {
                if (step < nsteps[i]) { // wave-uniform
                    // b_o: o = 1..8 from the host, then b_{o+4} = b_o * (1-q)^4  (models.py:198-206)
                    double w = (step == 0) ? wfirst[i] : wrun[i];
                    if (step >= 1)
                        wrun[i] *= r4[i];
                    w = (o_here < tq[i]) ? w : 0.0; // o ranges over 1..T-1 (models.py:239)
                    acc[i][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, w, acc[i][0], 0, 0, 0);
                    acc[i][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, w, acc[i][1], 0, 0, 0);
                }
            }
        }

        // ================= phase C: h_j * log p_j from the accumulators =================
        // f64 C/D layout: register r of a lane is row (lane>>4) + 4r, column lane&15.
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int bin = 16 * u + kq + 4 * r;
                const double h = tv.cnt[(int64_t)t * kTileBins + bin];
                const bool in_sp = tv.in_sp[(int64_t)t * kTileBins + bin] != 0.0;
#pragma unroll
                for (int i = 0; i < kMaxQTiles; ++i) {
                    if (qslot[i] >= 0) { // wave-uniform: the tile exists
                        const double p = acc[i][u][r];
                        if (in_sp) {
                            if (TAIL)
                                spacc[i].add(p);
                            if (h != 0.0)
                                llacc[i] += h * ((p <= 0.0) ? -INFINITY : fast_log(p, log_tab)); // utils.safe_log
                        }
                    }
                }
            }
        __syncthreads(); // Gs is rewritten by the next tile's phase A
    }

    // ---- per-q results: reduce over the 4 row groups of the accumulator layout ----
#pragma unroll
    for (int i = 0; i < kMaxQTiles; ++i) {
        double ll = llacc[i];
        ll += __shfl_xor(ll, 16, kWave);
        ll += __shfl_xor(ll, 32, kWave);
        double tail_term = 0.0;
        if (TAIL) {
            CompSum sp = spacc[i];
#pragma unroll
            for (int off = 16; off <= 32; off <<= 1) {
                const double ohi = __shfl_xor(sp.hi, off, kWave);
                const double olo = __shfl_xor(sp.lo, off, kWave);
                double e;
                two_sum(sp.hi, ohi, sp.hi, e);
                sp.lo += olo + e;
            }
            double s = sp.hi + sp.lo;
            if (!(s < 1.0))
                s = 1.0;
            if (s < 1.0)
                tail_term = m.tail * log(1.0 - s);
        }
        if (lane < 16 && qslot[i] >= 0) {
            const int32_t qo = plan.q_orig[qslot[i]];
            if (qo >= 0) {
                const int64_t flat = ce * plan.n_q + qo;
                if (flat >= plan.flat_begin && flat < plan.flat_end)
                    out_ll[flat - plan.flat_begin] = finite ? ll + tail_term : NAN;
            }
        }
    }
}

template <int NT, bool TAIL>
hipError_t launch_nt_tail(const DevModel &m, const TileView &tv, const FactoredPlan &plan, double *out_ll,
                     hipStream_t stream)
{
    const size_t lds = (size_t)kTileBins * (NT + 2) * sizeof(double);
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&ll_factored_kernel<NT, TAIL>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess)
            return e;
        configured = true;
    }
    const int tiles_per_block = kMaxQTiles * (NT / kWave);
    const unsigned n_qblocks = (unsigned)((plan.n_qtiles + tiles_per_block - 1) / tiles_per_block);
    const dim3 grid((unsigned)(plan.ce_end - plan.ce_begin), n_qblocks);
    hipLaunchKernelGGL((ll_factored_kernel<NT, TAIL>), grid, dim3(NT), lds, stream, m, tv, plan, out_ll);
    return hipGetLastError();
}

template <int NT>
hipError_t launch_nt(const DevModel &m, const TileView &tv, const FactoredPlan &plan, double *out_ll,
                     hipStream_t stream)
{
    return m.tail != 0.0 ? launch_nt_tail<NT, true>(m, tv, plan, out_ll, stream)
                         : launch_nt_tail<NT, false>(m, tv, plan, out_ll, stream);
}

} // namespace

hipError_t launch_ll_factored(const DevModel &m, const TileView &tv, const FactoredPlan &plan,
                              double *out_ll, hipStream_t stream)
{
    if (plan.ce_end <= plan.ce_begin)
        return hipSuccess;
    if (m.n_err != 8 || m.kind != 1 || plan.max_o > 512)
        return hipErrorInvalidValue;
    if (plan.max_o <= 256)
        return launch_nt<256>(m, tv, plan, out_ll, stream);
    if (plan.max_o <= 384)
        return launch_nt<384>(m, tv, plan, out_ll, stream);
    return launch_nt<512>(m, tv, plan, out_ll, stream);
}

} // namespace covest
