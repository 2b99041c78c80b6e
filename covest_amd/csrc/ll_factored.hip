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
// takes one (c, e) and all Q of them, key tile by key tile (32 keys):
//
//   phase A  lane = copy number o: the 8 error-class streams of streams.h walk the
//            32 keys (2 fp64 instr per pmf term) and store G'[key][o] -- their sum, without
//            the key's scale (round 3, see the kernel) -- to LDS.  Waves whose lanes are
//            all beyond Tmax skip it.
//   phase B  P[key][q] = sum_o G[key][o] * b_o(q) on the fp64 matrix pipe:
//            v_mfma_f64_16x16x4_f64, A = 16 keys x 4 o from LDS (conflict-free
//            ds_read_b64: row stride = 4 dwords mod 64), B = 4 o x 16 q generated
//            IN REGISTERS: b_{o+4} = b_o (1-q)^4 is one multiply per MFMA, the first
//            8 weights and the cut-off T come from the host (libm pow, as CPython)
//            -- the contraction reads no weights from memory at all.
//   phase C  h_j * log P[key][q] straight from the accumulator registers: in the
//            f64 C/D layout a lane keeps ONE q column, so the running LL of a unit
//            is a single register per lane (fast_log, fastmath.h).
//
// Round 2 (each step in DESIGN.md 6):
//   * phase A walks a tile with the LIVE error classes only: the classes' rates fall geometrically, so beyond the
//     first few hundred keys two of the eight streams are left and the others hold exact zeros in every lane
//     (streams.h enter_tile / `gone`): variants of the straight-line walk for 8, 4, 2, 1 and no stream;
//   * SHARED STEPS (tiles.h): the host lays the weight vectors out so that a q-tile's 16 columns differ in q1, q2
//     only; below the tile's smallest cut-off the contraction is then beta_col times a sum that does not depend
//     on the column -- summed once per key on the vector unit (Horner in (1-q)^-4) and brought in by ONE MFMA;
//   * the four logs of a unit go through fast_log_n stage by stage (their table reads in flight together);
//   * a p_j deep in the subnormal range is clamped and its row recorded: ll_fix_list_kernel (argmin.hip) redoes
//     those rows with K-direct's arithmetic (direct_point.h).
//
// The unit of phases B/C is (q-tile of 16 weight vectors, half of the key tile);
// the host deals units to waves longest-first, balanced per SIMD (tiles.h), and a
// wave's o-loop stops at its own unit's T.  gfx950 measured (tools/
// microbench_f64.hip): v_fma_f64 62 TFLOP/s, v_mfma_f64_16x16x4 75 TFLOP/s, and
// the two do NOT overlap (one fp64 datapath), so the phases are sequential and the
// kernel is bound by the fp64 pipe:
//   flops per (c,e) = B*8*(Tmax-1)*2  +  B*sum_q(T_q-1)*2  +  B*Q*(one log)
// against B*8*sum_q(T_q-1)*4 for the per-point formulation of SURVEY 8(d).
//
// Reference restated: covest/models.py:100-107 (LL), :211-242 (p_j), over
// covest/grid.py:59-64 (the grid map).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <mutex>
#include <type_traits>

#include "direct_point.h"
#include "fastmath.h"
#include "kernels.h"
#include "point_fetch.h"
#include "streams.h"
#include "wave.h"

namespace covest {

namespace {

typedef double d4 __attribute__((ext_vector_type(4)));

// The chunks' shares of p_j (list modes 2 and 3, tiles.h) travel through HBM TIMES 2^128: p_j <= 2.5, so nothing
// overflows, and a p_j below half a grid step -- which the reference's roundings may still keep alive, direct_point.h
// kZeroSteps -- does not flush to 0 on the way: finish_point decides on the exact value.
constexpr double kShareScale = 0x1p128;


// One MFMA step of the first N accumulator slots of a wave (they are sorted by length, so the active
// slots are always a prefix): all A fragments first -- N independent LDS reads in flight -- then per
// slot the weight b_o (advanced by (1-q)^4, cut off at o >= T: models.py:198-206,239) and the MFMA.
// Straight-line code: no per-slot branch separates the reads from their MFMAs.
// (Measured and not kept, round 3: the fragments of step i + 1 read during step i -- 0.896 against 0.864 ms: the
// registers of the second set are spilled elsewhere.)
template <int N, int MU>
__device__ __forceinline__ void contract_step(int i, const double *cur, const int (&a_off)[MU], const int (&cut)[MU],
                                              const double (&r4)[MU], double (&wrun)[MU], d4 (&acc)[MU])
{
    double a[N];
#pragma unroll
    for (int k = 0; k < N; ++k)
        a[k] = cur[a_off[k] + 4 * i];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const double w = (i < cut[k]) ? wrun[k] : 0.0; // this lane's o has reached T: models.py:239
        wrun[k] *= r4[k];
        acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[k], w, acc[k], 0, 0, 0);
    }
}

// PLAIN: the shape every dense grid of the reference's own set-up has -- list_mode 0, at most 8 error classes (one
// lane per copy number) -- compiled on its own so that the other modes' code costs it no registers.
//
// Round 3: the rows of a plain item stay UNSCALED in LDS and in the accumulators -- G' = the sum of the streams'
// scaled terms (streams.h), P'_j = p_j / scal_j -- because the per-key scale is the same for every copy number and
// every weight vector: log p_j = log(P'_j 2^-SC) + ln((k0-1)!/(k0+b)!), and the second term, weighted by h_j, is a
// CONSTANT of the histogram (tiles.h item_lconst, long double on the host) added once per point.  Phase A loses two
// multiplies per key pair and the 64 scalar registers the tile's scales took (the compiler fetched them pair by pair
// right before their use: an exposed scalar-load latency every two keys); the 2^-SC disappears into the exponent
// arithmetic of the log (fastmath.h fast_log_bits_n).  Only where p_j itself is needed -- sp_j with a tail, the
// chunks' shares of p_j in list modes 2 and 3 -- a row is multiplied by its scale, read from LDS.
// LDC: the row stride of G as a COMPILE-TIME constant (0: plan.ld) -- kLdWide = 290, the stride of every plain grid whose
// largest threshold_o - 1 lies in 257 .. 288 (C3: 284): the 32 stores of a tile's walk and the fragments' offsets become
// immediates (round 4).
[[maybe_unused]] constexpr int kLdWide = 290;
template <int NT, int HU, bool TAIL, bool PLAIN, int LDC = 0>
__global__ __launch_bounds__(NT) void ll_factored_kernel(const DevModel m, const int32_t n_tiles, const int32_t n_items,
                                                         const double *__restrict__ tile_dbl,
                                                         const int32_t *__restrict__ tile_int,
                                                         const FactoredPlan plan,
                                                         double *__restrict__ out_ll, const SubList sub_list)
{
    const TileView tv = tile_view_from(n_tiles, n_items, tile_dbl, tile_int);
#ifdef COVEST_DIAG // (COVEST_FACTORED_DIAG=4: where a workgroup's time OUTSIDE the walk goes -- stamps at the stages' ends)
    const long long dgx_0 = (long long)clock64();
    long long dgx_1 = 0, dgx_2 = 0, dgx_3 = 0, dgx_4 = 0;
#endif
    constexpr int NW = NT / kWave;
    constexpr int MU = 2 * HU; // accumulator slots per wave (6: the specialised step loops below assume it)
    static_assert(MU == 6, "contract loops are written for 6 slots");
    // SPO (round 5): a plain grid WITH A TAIL sums sp_j -- the sum of p_j over EVERY key, covest/models.py:103 -- per COPY
    // NUMBER in phase A instead of per weight vector in phase C:
    //     sp_q = sum_j sum_o b_o(q) G[o][j] = sum_o b_o(q) S[o],   S[o] = sum_j G[o][j]  (scaled: sum_j scal_j G'[j][o]),
    // one fused multiply-add a key in the builder lane that holds G'[j][o] in a register anyway, a compensated add a tile,
    // and ONE extra contraction at the very end whose two rows are S (hi, lo) -- instead of a multiply and a two-sum
    // (seven dependent instructions) per row, unit and tile in phase C, twelve accumulator registers a lane, and a
    // contraction per thirty-two count-less tiles.  The tiles without a count then only cost their walk, and a unit
    // whose sums are all -inf needs nothing more at all (the tail term is finite or 0): the dead-unit skip and the
    // last-tile-first order of the tail-less grids apply.
    constexpr bool SPO = PLAIN && TAIL;
    constexpr bool NEED_SCAL = !PLAIN; // p_j itself is needed row by row: the chunks' shares, a point list's sp_j
    constexpr int LOG_DEG = PLAIN ? 3 : 5;     // (point lists keep the 2e-16 log: refinements difference their values;
                                               // dense grids: 6e-13 absolute and the exponent's bias taken off per sum)
    const int LD = LDC ? LDC : plan.ld; // G row stride in doubles: 4 dwords (mod 64) -> conflict-free A reads
    extern __shared__ double Gs[]; // [n_buf][kTileBins][LD]; reused for the final per-q combine
    __shared__ __attribute__((aligned(16))) double log_tab[kLogTableDoubles];
    // rows handed back (direct_point.h), per accumulator slot and weight vector: first unit, last unit + 1 (0: none;
    // a unit = the 16 rows of a half tile: index 2 * tile + half).  One writer per entry: the lane with kq == 0.
    __shared__ unsigned sub_rec[NW * MU * 16 * 2];
    // per row of the key tile being logged and of the next one: {h_j, p_clamp in the row's units (0: no count -- such
    // a row is never "low")}, and the row's scale where p_j itself is needed.  Every unit reads the 4 rows of its lanes
    // from here (LDS, addressed by the unit's half) instead of selecting between two register sets per row.
    __shared__ __attribute__((aligned(16))) double2 rowc[2][kTileBins];
    __shared__ __attribute__((aligned(16))) double rows[NEED_SCAL ? 2 : 1][NEED_SCAL ? kTileBins : 2];
    // {(1-q)^16, -, (1-q)^4, (1-q)^(-4 nsh)} of every slot with shared steps (tiles.h): wave-uniform, read back as LDS broadcasts
    __shared__ __attribute__((aligned(16))) double rho_tab[PLAIN ? NW * MU * 4 : 4];
    __shared__ double lconst_s; // the constant the rows' scales add to every point's sum (tiles.h item_lconst)

    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    const double p_clamp = plan.p_clamp; // direct_point.h
    const double zero_frac = kZeroSteps * (kGridStep / p_clamp); // direct_point.h kZeroSteps, in units of p_clamp
    const int list_mode = PLAIN ? 0 : plan.list_mode;

    // ---- the (c, e) of this workgroup ----
    // list mode: workgroup = (unit, key segment); a unit is a point or a chunk of a point's copy numbers, the
    // key tiles are cut into n_seg contiguous segments so that even ONE point spreads over several CUs
    const int n_seg = list_mode ? plan.n_seg : 1;
    const int unit = list_mode ? (int)blockIdx.x / n_seg : (int)blockIdx.x;
    const int seg = list_mode ? (int)blockIdx.x - unit * n_seg : 0;
    // (the loop below walks ITEMS, tiles.h: a plain tile, or up to 32 all-zero-count tiles summed -- the latter
    // only exist with a tail; without one item i is tile i)
    const int seg_items = (n_items + n_seg - 1) / n_seg;
    const int t_begin = seg * seg_items, t_end = min(n_items, t_begin + seg_items);
    // (a dense grid's (c, e) pairs are taken from the END of the block: a pair costs the more the larger its rates are --
    // 452 k to 467 k ticks along c on C3 -- and the workgroups launched last decide how long the chip's last round runs;
    // round 4, from the per-workgroup stamps: 1.6 % of the launch was that tail, 0.4 % with the long ones first)
    const int64_t ce = (PLAIN || list_mode == 0) ? plan.ce_end - 1 - unit : plan.ce_begin + unit;
    // (list mode: workgroup i takes point i of a point list -- its own (c, e) AND its own single weight
    // vector, see tiles.h FactoredPlan::list_mode)
    const bool list = list_mode == 1 || list_mode == 2; // (3 is a dense grid's chunk: addressed like mode 0)
    const int64_t ic = list ? ce : ce / plan.n_e;
    const int64_t ie = list ? ce : ce - ic * plan.n_e;
    // A workgroup's start (round 5): everything it fetches before its first barrier -- the point, the log table, the
    // shared steps' constants, the items' constants -- is ASKED FOR first and written to LDS afterwards.  In the order
    // of their uses each fetch stood behind the LDS store of the one before (a loop or a branch between them: the
    // compiler moves no load across those), four round trips to the cache one after the other in front of every
    // workgroup's first barrier (the stage stamps of the diagnostic build, profiles/r05_c3_factored_stage_stamps.txt).
    double par[kMaxParams] = {plan.c_axis[ic], plan.e_axis[ie], 0, 0, 0};
    constexpr int kLogPerThread = (kLogTableDoubles + NT - 1) / NT;
    double log_v[kLogPerThread];
#pragma unroll
    for (int u = 0; u < kLogPerThread; ++u)
        log_v[u] = kLogTable[min(tid + u * NT, kLogTableDoubles - 1)];
    const double rho_v = PLAIN ? plan.unit_rho[(int64_t)blockIdx.y * NW * MU * 4 + min(tid, NW * MU * 4 - 1)] : 0.0;
    // (the items' constants: the last wave's, one load per lane and 64 items, added in a fixed order below)
    double lc_first = 0.0;
    if (wave == NW - 1 && t_begin + lane < t_end)
        lc_first = tv.item_lconst[t_begin + lane];
#pragma unroll
    for (int u = 0; u < kLogPerThread; ++u)
        if (tid + u * NT < kLogTableDoubles)
            log_tab[tid + u * NT] = log_v[u];
    if (PLAIN && tid < NW * MU * 4)
        rho_tab[tid] = rho_v;
#pragma unroll
    for (int u = 0; u < (NW * MU * 16 + NT - 1) / NT; ++u)
        if (tid + u * NT < NW * MU * 16) {
            sub_rec[2 * (tid + u * NT)] = 0xFFFFFFFFu;
            sub_rec[2 * (tid + u * NT) + 1] = 0;
        }
    clamp_point<2>(m, par);
    const bool finite = isfinite(par[0]) && isfinite(par[1]);
    // the rates of ALL error classes (padded to a multiple of 8; comb = 0 beyond the model's: such a class weighs 0)
    const int n_pass = PLAIN ? 1 : plan.n_pass; // lanes per copy number: one per 8 error classes (1 when max_error <= 8)
    // (by multiplication, as K-basic forms them -- point_fetch.h error_class_rate_mul, round 5: the device library's two
    // pows were 420 dependent instructions that every wave of the workgroup waited for at the barrier below; the strict
    // re-evaluation of the rows handed back, argmin.hip, forms the same products)
    if (tid < 8 * n_pass)
        Gs[tid] = error_class_rate_mul(m, par[0], par[1], tid, m.n_err);
    if (wave == NW - 1) { // the items' constants: one load per lane and 64 items, added in a fixed order
        double lc = 0.0 + lc_first; // (the first 64 items' were asked for above)
        for (int t = t_begin + lane + kWave; t < t_end; t += kWave)
            lc += tv.item_lconst[t];
        lc = wave_sum(lc);
        if (lane == 0) // (PLAIN: the logs come with the exponent's bias on, fastmath.h RAW -- off again for the whole sum:
                       // sum of h_j over the counted keys of this workgroup's items, tiles.h suf_h; a plain grid has one segment)
            lconst_s = PLAIN ? lc - kLogRawBias<kScaleBits> * tv.suf_h[0] : lc;
    }
    __syncthreads();
#ifdef COVEST_DIAG
    dgx_1 = (long long)clock64(); // the table, the point, the classes' rates, the items' constants; the first barrier
#endif

    // ---- phase-A state: lane = (pass, copy number): column pass * pass_stride + (o - o_base - 1) of G ----
    // With more than 8 error classes a copy number's classes are dealt to n_pass lanes, 8 each, whose columns the
    // contraction treats as extra copy numbers with the same weight: sum_s a_os TP = sum over the passes' partial sums.
    const int my_pass = n_pass == 1 ? 0 : tid / plan.pass_stride;
    const int o_local = n_pass == 1 ? tid : tid - my_pass * plan.pass_stride; // 0-based inside the chunk
    const bool wave_builds = wave * kWave < plan.n_columns; // wave-uniform
    // chunked point list (list_mode 2): copy numbers before the chunk per item; dense chunks (3): one for the launch
    const int o_base = PLAIN ? 0 : list_mode == 2 ? plan.item_obase[ce] : plan.o_base;
    const int o_mine = o_base + o_local + 1;
    double lam[8];
#pragma unroll
    for (int s = 0; s < 8; ++s)
        lam[s] = Gs[8 * my_pass + s];
    double n_total = -1.0;
    if (n_pass > 1) { // the mixture weights a_os are normalised over ALL classes: covest/models.py:225-233
        n_total = 0.0;
        for (int s = 0; s < 8 * n_pass; ++s)
            n_total += m.comb[s] * (1.0 - exp_neg_rn((double)o_mine * Gs[s]));
    }
    __syncthreads();
    StreamSet<8> st;
    // (measured and not kept, round 4: the prologue by ALL the waves -- the 8 max_o (copy number, class) pairs dealt to
    // the 512 lanes, n_os / ln x / the normaliser of each made once into the G area, the copy number's lane adding,
    // dividing and taking ln a_os: 30 KB less code, the same bits -- 0.705 against 0.702 ms: no gain, although the three
    // waves that build nothing wait 23 k ticks at the first barrier by the stamps of the diagnostic build)
    if (wave_builds) { // (wave-uniform) the waves that build nothing skip the mixture weights' exps, divisions and logs
        // (the stream constants by the reference's own route for every lane: the lanes of a wave are copy numbers, their
        // rates o x lambda_s span all three regimes of StreamSet::init's shortcut, and a wave that takes all three pays more
        // than the one route costs -- measured, round 5: C3 0.675 against 0.667 ms with the shortcut)
        // (measured and not kept, round 5: the constants stage by stage over the classes, four streams side by side -- exp(-x)
        // with both of its routes evaluated and one selected, the two logs and the division -- and the normaliser stream by
        // stream (without its branches the kernel spills 40 to 56 registers): the first four builder waves' constants
        // 15.3 k -> 13.4 k cycles by the stage stamps, the LAST builder wave's -- copy numbers beyond 256, every route of
        // every class taken, the one the first barrier waits for -- 17.3 k -> 20.2 k: C3 0.647 against 0.640 ms,
        // profiles/r05_c3_ab_staged_stream_constants_not_kept.txt)
        st.template init<false>(m, lam, o_mine, finite && my_pass < n_pass && o_local < plan.max_o, log_tab,
                                log_tab, 8 * my_pass, n_total);
    } else {
        st.gone = 0u;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            st.v[s] = 0.0;
            st.x[s] = 0.0;
            st.an.set(s, 0.0, -INFINITY);
        }
    }
#ifdef COVEST_DIAG
    dgx_2 = (long long)clock64(); // the streams' constants (builder waves)
#endif
    const bool lane_in_row = tid < LD - 2; // columns of G that exist (waves past them build nothing)
    if (tid < 64)
        Gs[(size_t)plan.n_buf * kTileBins * LD + tid] = 0.0; // the slack behind the buffers (see launch)
    // the two pad columns of every row are read by masked steps too: keep them finite
    if (tid < plan.n_buf * kTileBins) {
        Gs[(size_t)tid * LD + LD - 2] = 0.0;
        Gs[(size_t)tid * LD + LD - 1] = 0.0;
    }

    // ---- phase-B/C state: this wave's (q-tile, half) units ----
    const int col = lane & 15; // q column inside a tile / key row of the A fragment
    const int kq = lane >> 4;  // which of the 4 o of an MFMA step
    int len[MU], cont[MU], uhalf[MU], qslot[MU], cut[MU], a_off0[MU], nsh[MU];
    double r4[MU], llacc[MU];
    uint64_t dead[MU]; // lanes that met a p_j <= 0 with h_j != 0
    // (round 4) slots whose unit is DEAD: every one of its 16 weight vectors has met a p_j = 0 at a counted key, so
    // each of their sums is -inf whatever the later keys are worth (utils.safe_log; a tail term is finite or 0) --
    // the unit's remaining shared steps, MFMA steps and logs are skipped.  The units that die are the short ones
    // (small threshold_o against large keys): 13 % of C3's logs.  Wave-uniform, bit k = slot k and its pieces.
    unsigned off_slots = 0;
    bool newly_dead = false; // a column of this wave's met its first zero in the item just logged
    CompSum spacc[(TAIL && !SPO) ? MU : 1]; // (a point list's sp_j: per row, in phase C)
    CompSum so = {0.0, 0.0};                // (SPO, builder lanes) S[o] of this lane's copy number, times 2^SC
    // wave w's block of MU slots in the unit tables
    auto wave_block = [&](int w) -> int {
        if (list) // block 0 is empty (waves with no unit); point i owns blocks 1 + 2 i and 2 + 2 i, for the last two waves
            return w >= NW - 2 ? 1 + 2 * (int)ce + (w - (NW - 2)) : 0;
        return (int)blockIdx.y * NW + w;
    };
    const int slot_base = wave_block(wave) * MU;
    // The slots' entries of the unit tables in THREE rounds of loads, each round's in flight together (round 5): every
    // table holds a harmless entry for a slot without a unit (tile -1, lengths 0, first copy number 1 -- tiles.h), so
    // nothing needs to wait for the slot's tile to know whether it may be read.  Until then every load of a slot stood
    // behind `on ? ... : 0`, the slot's tile came first and the six slots one after the other: a dozen round trips to
    // the cache, 7 000 cycles of a builder wave between its streams' constants and the first key tile (the stage
    // stamps of the diagnostic build, profiles/r05_c3_factored_stage_stamps.txt).
    int u_qt[MU], u_s0[MU], u_len[MU], u_cont[MU], u_half[MU], u_nsh[MU], u_o0[MU], u_T[MU];
#pragma unroll
    for (int k = 0; k < MU; ++k) {
        const int at = slot_base + k;
        u_qt[k] = plan.unit_tile[at];
        u_s0[k] = plan.unit_s0[at];
        u_len[k] = plan.unit_len[at];
        u_cont[k] = plan.unit_cont[at];
        u_half[k] = plan.unit_half[at];
        u_nsh[k] = PLAIN ? plan.unit_nsh[at] : 0;
        u_o0[k] = plan.unit_o0[at];
    }
#pragma unroll
    for (int k = 0; k < MU; ++k) {
        const int qt = __builtin_amdgcn_readfirstlane(u_qt[k]);
        const bool on = qt >= 0;
        const int slot = (on ? qt : 0) * 16 + col;
        qslot[k] = on ? slot : -1;
        u_T[k] = plan.q_T[slot];
        r4[k] = plan.q_r4[slot];
    }
#pragma unroll
    for (int k = 0; k < MU; ++k) {
        const bool on = qslot[k] >= 0;
        const int first_step = __builtin_amdgcn_readfirstlane(u_s0[k]);
        len[k] = __builtin_amdgcn_readfirstlane(u_len[k]);
        cont[k] = __builtin_amdgcn_readfirstlane(u_cont[k]);
        uhalf[k] = __builtin_amdgcn_readfirstlane(u_half[k]);
        // shared steps (tiles.h): steps 1 .. nsh of the unit are summed on the vector unit; the MFMA loops below
        // then run over step 0 and the steps AFTER them, which is what a_off, cut and len are counted in
        nsh[k] = PLAIN ? __builtin_amdgcn_readfirstlane(u_nsh[k]) : 0;
        a_off0[k] = (16 * uhalf[k] + col) * LD + kq + 4 * first_step; // this lane's A fragment of the piece's first step
        // iterations of the piece during which this lane's copy number o0 + 4 i + kq (o0: the piece's first one,
        // counted from the chunk's start) is below T (the chunk's local one)
        const int t_lane = on ? u_T[k] : 0;
        const int o0 = __builtin_amdgcn_readfirstlane(u_o0[k]);
        cut[k] = ((t_lane - (o0 + kq) + 3) >> 2) - nsh[k];
        llacc[k] = 0.0;
        dead[k] = 0;
        spacc[(TAIL && !SPO) ? k : 0].hi = 0.0;
        spacc[(TAIL && !SPO) ? k : 0].lo = 0.0;
    }

    // what kind of slot k is, as bit k of a mask (wave-uniform; tested once per tile and slot): the first slot of a unit
    // (it takes the logs), a piece behind one (its accumulator is added to the slot before), a unit with shared steps
    unsigned m_first = 0, m_cont = 0, m_sh = 0;
#pragma unroll
    for (int k = 0; k < MU; ++k) {
        m_first |= (qslot[k] >= 0 && !cont[k]) ? 1u << k : 0u;
        m_cont |= cont[k] ? 1u << k : 0u;
        m_sh |= nsh[k] > 0 ? 1u << k : 0u;
    }
    // b_o of every slot's first (wfirst) and second (wrun, then advanced by (1-q)^4 per step) MFMA step:
    // an L2-resident host table.  The loads for tile t+1 are issued between tile t's MFMAs and its
    // logs, so their latency (2-3 us per tile when exposed) hides behind the logs and the barrier.
    double wfirst[MU], wrun[MU];
    auto load_weights = [&]() {
#pragma unroll
        for (int k = 0; k < MU; ++k) {
            // ONE 16-byte load per slot: {b_o of the first step, b_o the running weight starts from}
            const double2 pw = reinterpret_cast<const double2 *>(plan.piece_w)[(int64_t)(slot_base + k) * kWave + lane];
            wfirst[k] = pw.x;
            wrun[k] = pw.y;
        }
    };
    load_weights();
    // A NaN among (q1, q2, q) makes every b_o from the third on a NaN (covest/models.py:193-208; its threshold_o is
    // max(hist)), and the reference's likelihood with it.  The logs below work on the bits and would turn it into
    // a finite number: the columns concerned are found from the weights themselves -- AFTER the walk (round 4: found
    // here and kept, the six masks held twelve scalar registers through the whole kernel, which spills them).

    // ================= phase A: G'[key][o] of key tile t into `dst` =================
    // in-kernel stamps (diagnostic builds only): cycles per wave in build / contract / log / barrier
    long long dg_a = 0, dg_b = 0, dg_b0 = 0, dg_c = 0, dg_w = 0, dg_t0 = 0, dg_zero = 0, dg_enter = 0, dg_first = 0;
#ifdef COVEST_DIAG
    const bool diag = plan.diag != nullptr;
#else
    constexpr bool diag = false; // (the stamps exist in diagnostic builds only: tiles.h)
#endif
    // the 32 keys of a full tile with the streams 0 .. N-1 (the others are zero in every lane of the wave).  One
    // region under the row mask, straight-line inside: the sums go to LDS as they stand (no scale, see above)
    // SPO: returns sum_b u_b 2^SC of this lane's copy number over the tile's 32 keys (u_b: the true terms' sum at key
    // b, streams.h) WITHOUT the keys' 32 scales: with  W_b = (u_0 + .. + u_b) / scal_b,  scal_(b-1) / scal_b = k0 + b
    // (tiles.h) gives
    //     W_b = W_(b-1) (k0 + b) + g_b,          g_b = the streams' sum at key b, what goes to LDS anyway,
    // a fused multiply-add and an add a key on a counter that holds the key as a double (exact: keys <= 16384), and the
    // tile's sum is W_31 scal_31 = W_31 renorm 2^-SC.  W stays inside the range the streams' own terms are scaled for
    // (partial sums <= 2.5: at most 2.5 x 2^540 x 1e140).  No scalar is fetched in the walk (the tile's 32 scales
    // through scalar registers cost the TAIL variants 45 more spilled scalars and an exposed scalar-cache latency a
    // half).  Tiles with a FILLER key (a gap of the histogram: its scale is 0, it must add nothing) and short tiles
    // take the key-by-key path of build_tile with the scales.
    auto walk_tile = [&](auto n_tag, double *colp, double renorm, double k0) __attribute__((always_inline)) -> double {
        constexpr int N = decltype(n_tag)::value;
        if (!lane_in_row)
            return 0.0; // (lanes past the row hold no copy number: nothing of theirs is ever read)
        // squared rates (the streams advance two keys per step, streams.h step2): N multiplies per tile
        // rather than 16 registers held through phases B and C -- the empty asm keeps the compiler from
        // hoisting them back out of the tile loop (it would spill them)
        double xx[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            double xs = st.x[s];
            if (s < N) {
                asm volatile("" : "+v"(xs));
                xx[s] = xs * xs;
            } else {
                xx[s] = 0.0;
            }
        }
        double *row = colp; // (walked row by row: one vector add a store, no scalar multiply by the row's number)
        double tsum = 0.0;
        if (SPO) {
            double kk = k0; // the key of row b
#pragma unroll
            for (int b = 0; b < kTileBins; b += 2) {
                double g1, g2;
                st.template step2n<N>(xx, g1, g2);
                row[0] = g1;
                row[LD] = g2;
                row += 2 * LD;
                tsum = fma(tsum, kk, g1);
                kk += 1.0;
                tsum = fma(tsum, kk, g2);
                kk += 1.0;
            }
            tsum *= renorm;
        } else {
#pragma unroll
            for (int b = 0; b < kTileBins; b += 2) {
                double g1, g2;
                st.template step2n<N>(xx, g1, g2);
                row[0] = g1;
                row[LD] = g2;
                row += 2 * LD;
            }
        }
        st.template leave_tile_n<N>(renorm);
        return tsum;
    };
    auto build_tile = [&](int t, bool seg_start, double *dst) __attribute__((always_inline)) -> double {
        if (COVEST_SKIP_PHASE(plan, 1))
            return 0.0;
        const TileRec rc = tv.rec[t]; // (the tile's constants: one scalar load of one cache line, tiles.h)
        const double k0 = rc.k0;
        const int nb = rc.nb;
        // n_live: streams above it are zero in every lane of this wave (streams.h) -- most of them, for most tiles.
        // (Measured and not kept, round 5: the entry compiled for one and for two classes as well and picked by `gone` --
        // C3 0.671 against 0.661 ms, the trimmed histogram with a tail 0.395 against 0.396:
        // profiles/r05_c3_ab_entry_by_live_classes_not_kept.txt.  The gone classes' tests are scalar and cost little;
        // three copies of the entry cost the instruction cache more.)
        const int n_live = st.enter_tile(k0 - 1.0, k0 + (double)(nb - 1), rc.lgam_prev, rc.lgam_last,
                                         rc.run_start != 0 || seg_start); // (a key segment starts like a run: every stream anchored)
        if (diag) { // (diagnostic builds: entering the tile apart from walking it)
            const long long now__ = (long long)clock64();
            dg_enter += now__ - dg_t0;
            dg_t0 = now__;
        }
        double *colp = dst + (lane_in_row ? tid : 0);
        const double *scal = tv.scal + (int64_t)t * kTileBins;
        if (nb == kTileBins && !(SPO && rc.has_filler != 0)) { // the common case: straight-line code
            if (n_live > 4)
                return walk_tile(std::integral_constant<int, 8>{}, colp, rc.renorm, k0);
            if (n_live > 2)
                return walk_tile(std::integral_constant<int, 4>{}, colp, rc.renorm, k0);
            if (n_live == 2)
                return walk_tile(std::integral_constant<int, 2>{}, colp, rc.renorm, k0);
            if (n_live == 1)
                return walk_tile(std::integral_constant<int, 1>{}, colp, rc.renorm, k0);
            if (lane_in_row) { // nothing is on: G = 0 for this wave's copy numbers
                if (diag)
                    dg_zero += 1;
#pragma unroll
                for (int b = 0; b < kTileBins; ++b)
                    colp[b * LD] = 0.0;
            }
            return 0.0;
        }
        double tsum = 0.0;
        for (int b = 0; b < kTileBins; ++b) { // (rows past the tile's keys: 0 -- they carry no count and no scale)
            const double g = b < nb ? st.step() : 0.0;
            if (SPO && b < nb)
                tsum = fma(g, scal[b], tsum);
            if (lane_in_row)
                colp[b * LD] = g;
        }
        st.leave_tile(rc.renorm);
        return tsum * 0x1p476; // (tv.scal carries 2^(kBasicShift - SC); S[o] is kept times 2^SC: exact)
    };
    // The same walk over a tile WITHOUT counts (tail != 0 only): sum_j G[o][j] over its keys stays in a register
    // -- 32 terms of one sign added plainly, the compensated accumulator of phase C gets their contraction --
    // and is returned instead of 32 stores.  These rows ARE scaled (a sum over keys needs every key's own scale).
    auto sum_tile = [&](auto n_tag, const double *scal_ptr, double scal_or_k0, double renorm) __attribute__((always_inline)) -> double {
        constexpr int N = decltype(n_tag)::value; // the live streams, as in walk_tile
        double xx[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            double xs = st.x[s];
            if (s < N) {
                asm volatile("" : "+v"(xs));
                xx[s] = xs * xs;
            } else {
                xx[s] = 0.0;
            }
        }
        double gsum = 0.0;
        if (SPO) { // (the W recurrence of walk_tile: no scales; `scal` is the tile's first key here, as a double)
            double kk = scal_or_k0;
#pragma unroll
            for (int b = 0; b < kTileBins; b += 2) {
                double g1, g2;
                st.template step2n<N>(xx, g1, g2);
                gsum = fma(gsum, kk, g1);
                kk += 1.0;
                gsum = fma(gsum, kk, g2);
                kk += 1.0;
            }
            st.template leave_tile_n<N>(renorm);
            return gsum * renorm;
        }
        const double *scal = SPO ? nullptr : scal_ptr;
        // two halves of 16 keys, each half's scales fetched into scalar registers up front (one s_load_dwordx16 pair,
        // as K-basic does): fetched pair by pair right before their use, every second key waited for the scalar cache
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            double sc[16];
#pragma unroll
            for (int b = 0; b < 16; ++b)
                sc[b] = scal[16 * half + b];
            asm volatile("" ::: "memory"); // (keeps the loads above the half's arithmetic)
#pragma unroll
            for (int b = 0; b < 16; b += 2) {
                double g1, g2;
                st.template step2n<N>(xx, g1, g2);
                gsum = fma(g1, sc[b], gsum);
                gsum = fma(g2, sc[b + 1], gsum);
            }
        }
        st.template leave_tile_n<N>(renorm);
        return gsum * (1.0 / kBasicScale); // (tv.scal carries 2^kBasicShift, tiles.h)
    };
    auto build_tile_sum = [&](int t, bool seg_start) __attribute__((always_inline)) -> double {
        const TileRec rc = tv.rec[t];
        const double k0 = rc.k0;
        const int nb = rc.nb;
        // (SPO: the tile only enters S[o] -- streams.h enter_sum_tile: a stream is on where it matters to a sum)
        const int n_live = SPO ? st.enter_sum_tile(k0 - 1.0, k0 + (double)(nb - 1), rc.lgam_prev, rc.lgam_last,
                                                   rc.run_start != 0 || seg_start)
                               : st.enter_tile(k0 - 1.0, k0 + (double)(nb - 1), rc.lgam_prev, rc.lgam_last,
                                               rc.run_start != 0 || seg_start);
        const double *scal = tv.scal + (int64_t)t * kTileBins;
        if (nb == kTileBins && !(SPO && rc.has_filler != 0)) { // (the count-less tiles of a histogram's tail are full ones: the live streams only)
            if (n_live > 2)
                return sum_tile(std::integral_constant<int, 8>{}, scal, k0, rc.renorm);
            if (n_live == 2)
                return sum_tile(std::integral_constant<int, 2>{}, scal, k0, rc.renorm);
            if (n_live == 1)
                return sum_tile(std::integral_constant<int, 1>{}, scal, k0, rc.renorm);
            return 0.0; // nothing is on
        }
        double gsum = 0.0;
        for (int b = 0; b < nb; ++b)
            gsum = fma(st.step(), scal[b], gsum);
        st.leave_tile(rc.renorm);
        return SPO ? gsum * 0x1p476 : gsum * (1.0 / kBasicScale); // (SPO: S[o] is kept times 2^SC)
    };
    // One item into `dst`: a plain tile, or (TAIL) the per-tile sums of up to 32 count-less tiles as its rows.
    bool resync = false; // (SPO, wave-uniform) tiles were skipped: the next one walked anchors every stream afresh
    auto build_item = [&](int it, bool seg_start, double *dst) __attribute__((always_inline)) {
        const int first = TAIL ? __builtin_amdgcn_readfirstlane(tv.item_first[it]) : it;
        if (TAIL && tv.item_sum[it] != 0) {
            if (COVEST_SKIP_PHASE(plan, 1))
                return;
            const int n = __builtin_amdgcn_readfirstlane(tv.item_ntiles[it]);
            if (SPO) { // the tiles without a count only enter S[o]: nothing is stored, nothing contracted
                // The whole item -- up to 32 tiles, 1024 keys -- at a glance first: a stream's log-term is concave in the
                // key, so where the item does not hold its mode its largest value over the item is at an end.  If that is
                // below what a sum can hold (streams.h kSumWindowLn) for every stream of every lane of this wave, nothing
                // of the item reaches S[o]: it is skipped for the price of two multiply-adds a stream, where entering
                // each of its tiles in turn costs a thousand cycles a tile -- the long count-less stretch of a 10 000-key
                // histogram is mostly such items (a wave's copy numbers have their mass within a few hundred keys).
                // The streams are anchored afresh at the next tile that is walked (resync), as at the start of a run.
                if (!seg_start) {
                    const int tl = first + n - 1;
                    const TileRec ra = tv.rec[first], rb = tv.rec[tl];
                    const double ka = ra.k0 - 1.0, kb = rb.k0 + (double)(rb.nb - 1);
                    const double lga = ra.lgam_prev, lgb = rb.lgam_last;
                    bool matters = false;
#pragma unroll
                    for (int s = 0; s < 8; ++s) {
                        if ((st.gone >> s) & 1u)
                            continue; // (wave-uniform)
                        const double lx = st.an.lx(s), c = st.an.c(s);
                        const double top = fmax(fma(ka, lx, c - lga), fma(kb, lx, c - lgb));
                        matters = matters || !(top < StreamSet<8>::kSumWindowLn) || (st.x[s] >= ka - 1.0 && st.x[s] <= kb + 1.0);
                    }
                    if (!__any(matters)) { // wave-uniform
#pragma unroll
                        for (int s = 0; s < 8; ++s)
                            st.v[s] = 0.0;
                        resync = true;
                        return;
                    }
                }
                for (int r = 0; r < n; ++r) {
                    // (every stream of this wave's copy numbers has GONE -- off in all lanes, outside the window, past its
                    // mode, streams.h: exact zeros from here on.  The far end of a long histogram: a wave of small copy
                    // numbers is done with H10k_rep's 10 000 keys after the first two thousand)
                    if (st.gone == 0xFFu && !(seg_start && r == 0))
                        break;
                    so.add(build_tile_sum(first + r, (seg_start && r == 0) || resync));
                    resync = false;
                }
                return;
            }
            double *colp = dst + (lane_in_row ? tid : 0);
            for (int r = 0; r < n; ++r) {
                const double gsum = build_tile_sum(first + r, seg_start && r == 0);
                if (lane_in_row)
                    colp[r * LD] = gsum;
            }
            for (int r = n; r < kTileBins; ++r)
                if (lane_in_row)
                    colp[r * LD] = 0.0;
        } else {
            const double tsum = build_tile(first, seg_start || (SPO && resync), dst);
            resync = false;
            if (SPO)
                so.add(tsum);
        }
    };

    // in-kernel stamps (diagnostic builds only): cycles per wave in build / contract / log / barrier
#define STAMP(acc)                                    \
    if (diag) {                                       \
        const long long now__ = (long long)clock64(); \
        acc += now__ - dg_t0;                         \
        dg_t0 = now__;                                \
    }
    if (diag)
        dg_t0 = (long long)clock64();
#ifdef COVEST_DIAG
    dgx_3 = (long long)clock64(); // the units' tables and first weights
#endif

    // With two buffers the builders fill item t+1 while every wave contracts item t: one barrier per item, and the
    // host's unit assignment charges the builders for phase A.  ONE loop for both set-ups (and ONE copy of phase A in
    // the code: three used to be inlined): interval p builds the item of position p + 1 and contracts that of position p.
    // (Measured and not kept, round 3: no barrier between the items but two counters per buffer in LDS -- builder waves
    // that have filled it, waves that are done with it -- so that a wave may run an item ahead: 0.846-0.849 against
    // 0.831-0.832 ms; the polling costs more than the hardware barrier's lock step, whose waits are 10 % of the kernel.)
    const bool dbuf = plan.n_buf == 2;
    // the rows' constants of the item an interval builds are fetched one interval ahead (32 lanes of wave 0; the two
    // or three loads are independent and have a whole interval to arrive)
    double nxt_h = 0.0, nxt_c = 0.0, nxt_s = 0.0;
    auto fetch_rows = [&](int it) {
        if (tid < kTileBins && it < t_end) {
            const int64_t at = (int64_t)it * kTileBins + tid;
            nxt_h = tv.item_cnt[at];
            nxt_c = tv.item_iscal[at]; // (0 for a row without a count)
            if (NEED_SCAL)
                nxt_s = tv.item_scal[at];
        }
    };
    // THE LAST KEY TILE FIRST (a plain grid without a tail; round 4).  A weight vector whose sum is -inf -- 44 % of C3's
    // points -- has met a p_j = 0 at a counted key, and the key where that happens is, with hardly an exception, the
    // LARGEST one: p_j is a mixture of Poissons whose means stop at (threshold_o - 1) c, so it only underflows beyond
    // them.  Taken in ascending order a doomed unit is walked, contracted and logged until its columns die near the end;
    // with the last tile taken first it dies in the first interval and the dead-unit skip (above) leaves out all the
    // rest.  The order of a point's sums changes by that one term; its value is -inf or finite either way, and the rows
    // handed back are recorded as a range (min, max).  Position p of the walk is tile tile_at(p); the buffers, rowc and
    // rows alternate with the POSITION.  The streams are anchored afresh at the first position (the last tile: nothing
    // is on yet) and at the second (tile 0 is a run start anyway; what they know of the last tile -- `gone` -- is
    // forgotten there).
    // Round 5, WITH A TAIL (SPO): the same, with the last item that HOLDS A COUNT taken first (the items behind it are
    // sums over count-less tiles: nothing dies there) -- sp_j no longer hangs on the contraction (S[o] above), and the
    // streams are anchored afresh once more, at the item behind the one taken out of turn.
    int last_item = t_end - 1; // the item taken first
    if (SPO)
        while (last_item > t_begin && tv.item_sum[last_item] != 0)
            --last_item;
    const bool last_first = PLAIN && t_end - t_begin >= 2 && last_item > t_begin;
    auto tile_at = [&](int p) -> int { return last_first ? (p == t_begin ? last_item : (p <= last_item ? p - 1 : p)) : p; };
    // SPO: one more position behind the items -- the contraction of S (rows 0 and 1 of the buffer: hi, lo)
    const int t_endx = t_end + (SPO ? 1 : 0);
    fetch_rows(tile_at(t_begin));
    for (int p = t_begin - (dbuf ? 1 : 0); p < t_endx; ++p) {
        const int pb = dbuf ? p + 1 : p;
        if (SPO && pb == t_end) {
            if (wave_builds && lane_in_row) { // S[o] (x 2^-kBasicShift: exact) as the two rows of one more "item"
                double *dst = Gs + (dbuf ? (pb & 1) * kTileBins * LD : 0) + tid;
                dst[0] = so.hi * 0x1p-540;
                dst[LD] = so.lo * 0x1p-540;
            }
        } else if (pb < t_end) {
            if (tid < kTileBins) { // read after the barrier that makes the item readable
                rowc[pb & 1][tid] = make_double2(nxt_h, p_clamp * nxt_c);
                if (NEED_SCAL)
                    rows[pb & 1][tid] = nxt_s;
            }
            fetch_rows(pb + 1 < t_end ? tile_at(pb + 1) : t_end);
            if (wave_builds) {
                if (last_first && pb == t_begin + 1)
                    st.gone = 0u;
                build_item(tile_at(pb), pb == t_begin || (last_first && (pb == t_begin + 1 || pb == last_item + 1)),
                           Gs + (dbuf ? (pb & 1) * kTileBins * LD : 0));
            }
        }
        STAMP(dg_a)
        if (!dbuf)
            __syncthreads();
        const bool sp_item = SPO && p == t_end; // (wave-uniform) the last position: S x b
        // (SPO) where the units leave their shares of sp_j: [2][NW][MU][16] doubles in the buffer that is NOT contracted in
        // the last interval (one buffer only: behind the two rows of S -- rows whose products nobody uses)
        double *const sp_parts = dbuf ? Gs + ((t_end + 1) & 1) * kTileBins * LD + NW * MU * 16 : Gs + 2 * LD + NW * MU * 16;
        if (p >= t_begin && !(SPO && !sp_item && tv.item_sum[tile_at(p)] != 0)) { // (SPO: a sum item was phase A only)
            const int t = tile_at(p); // the tile this interval contracts and logs
            const double *cur = Gs + (dbuf ? (p & 1) * kTileBins * LD : 0);
            // ================= phase B: P' = G' x b on the matrix pipe =================
            d4 acc[MU];
            const d4 zero4 = (d4){0.0, 0.0, 0.0, 0.0};
            // Every slot's accumulator starts from its first MFMA (C = 0 costs nothing; zeroing 8 registers per slot
            // does); the step-0 fragments of all slots are read first, in flight together.
            // A unit with shared steps (tiles.h) sums them on the vector unit:
            //     sum_{i = 1 .. nsh} G'[key][1 + 4 i + kq] r^(i - 1),   r = (1-q)^4 (wave-uniform: the tile has one q)
            // for this lane's key and o mod 4 -- weights RELATIVE TO THE FIRST shared step, so they only fall (the
            // unscaled rows reach 1e297; with weights relative to the step AFTER the shared ones, up to 1e10, the sum
            // could leave the range) -- as four Horner chains in r^4 (chain c: the steps i = 1 + c + 4 m) walked from
            // the last step down, eight loads in flight per trip.  (Measured and not kept, round 3: the loads issued
            // three groups ahead of their FMAs through a ring of four register sets, 0.934 against 0.872 ms -- the
            // registers it takes are spilled elsewhere; and one fused pass over all the wave's units with one chain
            // each, 1.08 ms.)  ONE MFMA brings the sum in: the four lanes of a key add up inside it,
            // every column gets its own b_o of that first shared step, made from the weight the slot holds (b_o of
            // the first step after them) times r^-nsh.
            double a0[MU];
#pragma unroll
            for (int k = 0; k < MU; ++k)
                a0[k] = cur[a_off0[k]]; // (an idle slot reads a valid address and uses nothing)
            if (diag) { // (diagnostic builds: until the first fragments and the weights are there)
                double sink = 0.0;
#pragma unroll
                for (int k = 0; k < MU; ++k)
                    sink += a0[k] + wfirst[k];
                asm volatile("" ::"v"(sink));
                const long long now__ = (long long)clock64();
                dg_first += now__ - dg_t0;
                dg_t0 = now__;
            }
#pragma unroll
            for (int k = 0; k < MU; ++k) {
                // (a dead unit, off_slots: its len[k] is 0 by now and its bit of m_sh cleared -- it takes the branch below
                // and starts from zero)
                if (!PLAIN || !((m_sh >> k) & 1u) || COVEST_SKIP_PHASE(plan, 8)) { // wave-uniform
                    acc[k] = len[k] > 0 ? __builtin_amdgcn_mfma_f64_16x16x4f64(a0[k], wfirst[k], zero4, 0, 0, 0) : zero4;
                    continue;
                }
                const double rr16 = rho_tab[4 * (wave * MU + k)]; // (LDS broadcasts)
                const double2 rt = *reinterpret_cast<const double2 *>(&rho_tab[4 * (wave * MU + k) + 2]);
                const double rr4 = rt.x, rho_n = rt.y;
                const double *g1 = cur + a_off0[k]; // step 0 of the unit (o = 1 .. 4); step i at g1[4 i]
                // (the empty asm: what derives from the slot's step count -- the trip count, the three "is there a head"
                // masks -- is made afresh per tile, five scalar instructions, instead of being held across the walk: the
                // compiler hoisted all of it out of the tile loop, 40 scalar registers that it then spilled to vector
                // lanes and fetched back with v_readlane, a VECTOR instruction, every tile)
                int nsh_k = nsh[k];
                asm volatile("" : "+s"(nsh_k));
                const int n4 = nsh_k >> 2, rem = nsh_k & 3;
                // the top `rem` steps (m = n4) are the heads of chains 0 .. rem - 1
                const double *top = g1 + 4 * (1 + 4 * n4);
                // the heads WITHOUT branches: a head that does not exist is read from the zeroed slack behind the buffers
                // (one select on the address), so that the three loads, the slot's constants above and the first trip's
                // eight are in flight together -- one LDS round trip at the top of a slot instead of three or four
                const double *zp = Gs + (size_t)plan.n_buf * kTileBins * LD;
                double h0 = *(rem > 0 ? top : zp), h1 = *(rem > 1 ? top + 4 : zp), h2 = *(rem > 2 ? top + 8 : zp), h3 = 0.0;
                {
                    // The trips' reads are written out as ds_read_b64: left alone the compiler pairs them into
                    // ds_read2_b64, which the LDS serves at half the rate and with banks counted mod 32 -- rows 8 apart
                    // collide there (the row stride is 4 dwords mod 64: conflict-free for ds_read_b64 only).  The
                    // compiler does not count reads it cannot see: the wait is part of the statement.
                    unsigned qa = (unsigned)(uintptr_t)(top - 32); // LDS byte address of the trip's lowest step
                    int gq = n4;
                    for (; gq >= 2; gq -= 2, qa -= 256) { // eight loads in flight per trip
                        double b0, b1, b2, b3, c0, c1, c2, c3;
                        asm volatile("ds_read_b64 %0, %8 offset:128\n\tds_read_b64 %1, %8 offset:160\n\t"
                                     "ds_read_b64 %2, %8 offset:192\n\tds_read_b64 %3, %8 offset:224\n\t"
                                     "ds_read_b64 %4, %8\n\tds_read_b64 %5, %8 offset:32\n\t"
                                     "ds_read_b64 %6, %8 offset:64\n\tds_read_b64 %7, %8 offset:96\n\t"
                                     "s_waitcnt lgkmcnt(0)"
                                     : "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3), "=&v"(c0), "=&v"(c1), "=&v"(c2), "=&v"(c3)
                                     : "v"(qa));
                        h0 = fma_vvv(fma(h0, rr16, b0), rr16, c0); // (three-address: no copy back into the chain's register)
                        h1 = fma_vvv(fma(h1, rr16, b1), rr16, c1);
                        h2 = fma_vvv(fma(h2, rr16, b2), rr16, c2);
                        h3 = fma_vvv(fma(h3, rr16, b3), rr16, c3);
                    }
                    if (gq > 0) {
                        double b0, b1, b2, b3;
                        asm volatile("ds_read_b64 %0, %4 offset:128\n\tds_read_b64 %1, %4 offset:160\n\t"
                                     "ds_read_b64 %2, %4 offset:192\n\tds_read_b64 %3, %4 offset:224\n\t"
                                     "s_waitcnt lgkmcnt(0)"
                                     : "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3)
                                     : "v"(qa));
                        h0 = fma_vvv(h0, rr16, b0);
                        h1 = fma_vvv(h1, rr16, b1);
                        h2 = fma_vvv(h2, rr16, b2);
                        h3 = fma_vvv(h3, rr16, b3);
                    }
                }
                const double hs = fma(fma(fma(h3, rr4, h2), rr4, h1), rr4, h0);
                acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(hs, wrun[k] * rho_n, zero4, 0, 0, 0);
                acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[k], wfirst[k], acc[k], 0, 0, 0);
            }
            STAMP(dg_b0)
            // the remaining steps, specialised on the number of slots still running (len is sorted)
            if (!COVEST_SKIP_PHASE(plan, 2)) {
                int a_off[MU]; // this lane's A fragments counted from the first step after the shared ones (per tile: 6 adds
                               // instead of 6 registers held through the other phases)
#pragma unroll
                for (int k = 0; k < MU; ++k)
                {
                    int nsh_k = nsh[k];
                    asm volatile("" : "+s"(nsh_k));
                    a_off[k] = a_off0[k] + 4 * nsh_k;
                }
                // (measured and not kept, round 4: one address register a slot with the buffer's base folded in and two
                // steps a trip, the second fragment an immediate offset away -- the compiler moves the induction to the
                // scalar unit, six s_add a trip: 0.867-0.871 against 0.855-0.858 ms)
                int i = 1;
                for (; i < len[5]; ++i)
                    contract_step<6, MU>(i, cur, a_off, cut, r4, wrun, acc);
                for (; i < len[4]; ++i)
                    contract_step<5, MU>(i, cur, a_off, cut, r4, wrun, acc);
                for (; i < len[3]; ++i)
                    contract_step<4, MU>(i, cur, a_off, cut, r4, wrun, acc);
                for (; i < len[2]; ++i)
                    contract_step<3, MU>(i, cur, a_off, cut, r4, wrun, acc);
                for (; i < len[1]; ++i)
                    contract_step<2, MU>(i, cur, a_off, cut, r4, wrun, acc);
                for (; i < len[0]; ++i)
                    contract_step<1, MU>(i, cur, a_off, cut, r4, wrun, acc);
            }
            // pieces of one unit: add the accumulators into the unit's first slot.  A BRANCH per slot (the empty asm
            // keeps the compiler from turning it into four adds and eight selects for every slot, taken or not)
#pragma unroll
            for (int k = MU - 1; k >= 1; --k)
                if ((m_cont >> k) & 1u) { // wave-uniform
                    asm volatile("" ::: "memory");
                    acc[k - 1] += acc[k];
                }
            STAMP(dg_b)
            // ================= phase C: h_j * log p_j from the accumulators =================
            const bool item_is_sum = TAIL && !SPO && tv.item_sum[t] != 0; // wave-uniform
            // f64 C/D layout: register r of a lane is row (lane>>4) + 4r, column lane&15.
#pragma unroll
            for (int k = 0; k < MU; ++k) {
                if (list_mode == 2) { // a chunk of a point's copy numbers: hand p_j's share on (column 0 only)
                    if (qslot[k] >= 0 && !cont[k] && col == 0) {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            plan.partial[(ce * tv.n_items + t) * kTileBins + 16 * uhalf[k] + kq + 4 * r] = // (x 2^128: kShareScale)
                                acc[k][r] * (rows[NEED_SCAL ? (p & 1) : 0][NEED_SCAL ? 16 * uhalf[k] + kq + 4 * r : 0] * kShareScale);
                    }
                    continue;
                }
                if (list_mode == 3) { // a chunk of the copy numbers of a dense grid's LONG weight vectors: every
                                           // column's share of p_j goes to (the first chunk) or is added to the block's
                                           // buffer in HBM; ll_finish_dense takes the logs.  Each element has one owner.
                    if (qslot[k] >= 0 && !cont[k]) {
                        double *row = plan.partial + (((ce - plan.ce_first) * plan.n_cols_partial + qslot[k]) * tv.n_items + t) * kTileBins +
                                      16 * uhalf[k] + kq;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const double share = acc[k][r] * (rows[NEED_SCAL ? (p & 1) : 0][NEED_SCAL ? 16 * uhalf[k] + kq + 4 * r : 0] * kShareScale);
                            row[4 * r] = plan.o_base == 0 ? share : row[4 * r] + share;
                        }
                    }
                    continue;
                }
                if (SPO && sp_item) {
                    // rows 0 and 1 are S (hi, lo): register 0 of the lanes with kq = 0 and kq = 1 holds this column's
                    // sum_o b_o S_hi[o] and sum_o b_o S_lo[o].  Straight to where the last step below looks for them
                    // (sp_parts: LDS nobody reads any more) -- held in registers until then they would be carried through
                    // every interval of the walk.  A dead unit writes its zeros: its sum is -inf whatever sp_j is.
                    if (qslot[k] >= 0 && !cont[k] && uhalf[k] == 0 && lane < 32)
                        sp_parts[(kq * NW * MU + wave * MU + k) * 16 + col] = acc[k][0];
                    continue;
                }
                if (TAIL && !SPO && item_is_sum) { // rows are sums over count-less tiles (scaled ones): they only enter sp_j
                    if ((m_first >> k) & 1u) {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            spacc[k].add(acc[k][r]);
                    }
                    continue;
                }
                // (a dead unit's bit of m_first is cleared: nothing it could add changes its -inf)
                if (((m_first >> k) & 1u) && !COVEST_SKIP_PHASE(plan, 4)) { // wave-uniform: first slot of a unit
                    // Everything out of the ordinary -- p_j <= 0, or deep in the subnormal range (below p_clamp,
                    // direct_point.h), at a key with h_j != 0 -- is caught by ONE compare per row against the clamp in
                    // the row's own units (0 for a row without a count: filler keys, a tile's padding, zero counts
                    // with a tail -- never "low"), made BEFORE the logs, and sorted out in a branch the wave takes
                    // for one unit in twenty.  Rows without a count add 0 * log below.
                    const double2 *rc = &rowc[p & 1][16 * uhalf[k] + kq]; // {h, clamp} of this lane's rows kq, kq + 4, ...
                    double h4[4], c4[4], x4[4], lg4[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double2 hc = rc[4 * r];
                        h4[r] = hc.x;
                        c4[r] = hc.y;
                    }
                    if (TAIL && !SPO) { // sp_j needs p_j itself (filler and padding keys: scale 0), before the clamp
                        const double *sr = &rows[NEED_SCAL ? (p & 1) : 0][NEED_SCAL ? 16 * uhalf[k] + kq : 0];
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            spacc[k].add(acc[k][r] * sr[NEED_SCAL ? 4 * r : 0]);
                    }
                    uint64_t low[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        low[r] = __ballot(acc[k][r] < c4[r]);
                    if (__builtin_expect(((low[0] | low[1] | low[2] | low[3]) & ~dead[k]) != 0, 0)) {
                        // wave-uniform, cold (a lane that is dead already has nothing more to report).  Kept SHORT: a
                        // wave in here holds up its whole workgroup at the tile's barrier
                        uint64_t subm = 0, zero = 0;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            // p_j = 0 in the reference: every term of its sum is at most the sum, and a term below
                            // 2^-1075 is flushed by the extension's cast to double (c_src/covest_poissonmodule.c:32) --
                            // so a row whose p_j (row value x scale) is safely below 1/8 of a grid step (the roundings
                            // on the reference's way can keep anything above that alive: direct_point.h kZeroSteps) is a
                            // zero, not a candidate for the strict evaluation.  (The unscaled row value itself never
                            // underflows: half of the C3 grid is -inf this way, and would otherwise queue up for the
                            // strict kernel.)
                            const uint64_t z = __ballot(acc[k][r] <= c4[r] * zero_frac) & low[r];
                            zero |= z;
                            subm |= low[r] & ~z;
                            // log(max(p_j, p_clamp)): what a p_j deep in the subnormal range contributes is then a
                            // known constant, which the strict evaluation of that key replaces (direct_point.h).  In
                            // place: the accumulator is of no further use, and a copy would cost every unit a move.
                            acc[k][r] = max_raw(acc[k][r], c4[r]);
                        }
                        // utils.safe_log: p_j <= 0 with h_j != 0 makes the sum -inf; kept as a lane mask in SGPRs -- for
                        // all four row groups of the weight vector (column) at once: its sum is -inf whichever of them
                        // met the zero, and the other three need not come through here for it
                        const uint64_t zc = (zero | (zero >> 16) | (zero >> 32) | (zero >> 48)) & 0xFFFFull;
                        dead[k] |= zc * 0x0001000100010001ull;
                        newly_dead = newly_dead || zc != 0; // (wave-uniform: look for dead units after this item's logs)
                        // p_j DEEP IN THE SUBNORMAL RANGE: the unit (this half tile) is recorded for every weight vector
                        // (column) concerned -- first and last unit met; one writer per entry, the lane of row group 0.
                        // The strict evaluation of its counted rows follows in ll_fix_list_kernel (argmin.hip)
                        const uint64_t cm = (subm | (subm >> 16) | (subm >> 32) | (subm >> 48)) & 0xFFFFull;
                        if (kq == 0 && ((cm >> col) & 1)) {
                            const unsigned u = 2u * (unsigned)(TAIL ? tv.item_first[t] : t) + (unsigned)uhalf[k];
                            unsigned *rec = &sub_rec[((wave * MU + k) * 16 + col) * 2];
                            rec[0] = min(rec[0], u);
                            rec[1] = max(rec[1], u + 1);
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        x4[r] = acc[k][r];
                    // no branch on h (it differs between the lanes' rows): the four logs of a unit go through
                    // fast_log_bits_n stage by stage, their table reads in flight together.  (A dead lane's row may be
                    // 0 here: the log of those bits is a finite number nobody uses.)
                    fast_log_bits_n<4, LOG_DEG, kScaleBits, PLAIN>(x4, lg4, log_tab);
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        llacc[k] = fma(h4[r], lg4[r], llacc[k]);
                    __builtin_amdgcn_sched_barrier(0); // ... one unit at a time: registers
                }
            }
            // the weights of the next key tile, behind the logs (round 4: issued before them -- where their latency hides
            // best -- the 24 registers they land in were live through the logs and the kernel spilled 18-74; here it
            // spills none and measures the same, 0.856 against 0.856-0.862 ms: the builders' phase A and the barrier hide
            // the loads just as well)
            load_weights();
            if (PLAIN && newly_dead) { // retire the units that died in this item: the first slot and the pieces behind it
                newly_dead = false;
#pragma unroll
                for (int k = 0; k < MU; ++k) {
                    const bool first_dead = ((m_first >> k) & 1u) && dead[k] == ~0ull;
                    const bool piece_dead = k > 0 && ((m_cont >> k) & 1u) && ((off_slots >> (k - 1)) & 1u);
                    if (first_dead || piece_dead) { // wave-uniform
                        off_slots |= 1u << k;
                        m_first &= ~(1u << k); // no logs, no shared steps any more ...
                        m_sh &= ~(1u << k);
                        len[k] = 0; // ... and no MFMA steps (the step loops stay correct with a zero among the sorted
                                    // lengths: every live slot k still gets max(len[k..5]) = len[k] steps)
                    }
                }
            }
            STAMP(dg_c)
        }
        __syncthreads(); // the tile just contracted may be overwritten, the one just built may be read
#ifdef COVEST_DIAG
        if (diag && (plan.skip_phases & 0x10000)) { // (COVEST_FACTORED_DIAG=2) the wait at the barrier by eighths of the walk
            const long long now__ = (long long)clock64();
            if (lane == 0) {
                const int grp = (plan.skip_phases & 0x20000) ? min(7, p - t_begin + 1) // (=3: the first seven intervals one by one)
                                                              : min(7, (p - t_begin + 1) * 8 / (t_end - t_begin + 1));
                plan.diag[((int64_t)(blockIdx.x * gridDim.y + blockIdx.y) * NW + wave) * 8 + grp] += now__ - dg_t0;
            }
        }
#endif
        STAMP(dg_w)
    }

#ifdef COVEST_DIAG
    dgx_4 = (long long)clock64(); // the walk
#endif
    if (diag && lane == 0 && !(plan.skip_phases & 0x50000)) {
        long long *d = plan.diag + ((int64_t)(blockIdx.x * gridDim.y + blockIdx.y) * NW + wave) * 8;
        d[0] = dg_a;
        d[1] = dg_b;
        d[2] = dg_c;
        d[3] = dg_w;
        d[4] = dg_b0;
        d[5] = dg_zero; // key tiles whose G columns of this (builder) wave were all zero
        d[6] = dg_enter; // of phase A: StreamSet::enter_tile
        d[7] = dg_first; // of phase B: waiting for the first A fragments (LDS) and the tile's weights (HBM / L2)
    }
#undef STAMP
    if (list_mode >= 2)
        return; // (wave-uniform for the whole workgroup) the chunks are combined by ll_finish_partials / _dense
    // ---- per-q results: sum the 4 row groups of the accumulator layout, then the two
    //      halves of each q-tile (they may live on different waves) through LDS ----
    const double lconst = lconst_s; // (read before the buffer below is reused: lconst_s is static LDS of its own)
    // The unit-table words of the entries this thread finishes below -- the four words of an entry, then its weight
    // vector's place in the grid: two round trips to the cache -- asked for together (round 5: entry after entry, word
    // after word behind `||`, they stood between the last barrier and every workgroup's stores).  With a tail they are
    // asked for HERE, and pass during the sums and the barrier below; without one, behind the barrier in the loop as
    // before: the walk's register allocation hangs on the shape of this epilogue, and the tail-less kernel measured
    // 0.6384 and 0.6500 against 0.6354 ms with two forms of the early fetch (the trimmed C3 with its tail: 0.3817 and
    // 0.3851 against 0.3861) -- profiles/r05_c3_ab_result_entries_early.txt.
    constexpr int kEntries = (NW * MU * 16 + NT - 1) / NT;
    constexpr bool kEntriesEarly = TAIL;
    int en_qt[kEntries], en_half[kEntries], en_cont[kEntries], en_pair[kEntries], en_qo[kEntries];
    auto fetch_entries = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < kEntries; ++u) {
            const int e = min(tid + u * NT, NW * MU * 16 - 1);
            const int at = wave_block(e / (MU * 16)) * MU + (e / 16) % MU;
            en_qt[u] = plan.unit_tile[at];
            en_half[u] = plan.unit_half[at];
            en_cont[u] = plan.unit_cont[at];
            en_pair[u] = plan.unit_pair[at];
        }
#pragma unroll
        for (int u = 0; u < kEntries; ++u)
            en_qo[u] = plan.q_orig[max(en_qt[u], 0) * 16 + (tid & 15)]; // (NT is a multiple of 16: the entry's column is tid & 15)
    };
    if (kEntriesEarly)
        fetch_entries();
    double *part_ll = Gs;                               // [NW][MU][16]
    // compensated sp_j parts (SPO: where the last contraction left them -- behind part_ll if that is buffer 0)
    double *part_hi = !SPO ? Gs + (size_t)NW * MU * 16
                           : (plan.n_buf == 2 ? Gs + ((t_end + 1) & 1) * kTileBins * LD : Gs + 2 * LD) + NW * MU * 16;
    double *part_lo = part_hi + (size_t)NW * MU * 16;
    // (the slots' first weights are in the registers: they do not depend on the key tile, and every interval that used
    // them up fetched them again behind its logs -- until round 5 they were loaded once more here, a round trip to the
    // cache in front of every workgroup's results.  A NaN column is a NaN in them)
#pragma unroll
    for (int k = 0; k < MU; ++k) {
        double ll = llacc[k];
        const uint64_t nm = __ballot(wfirst[k] != wfirst[k] || wrun[k] != wrun[k]);
        const uint64_t nan_cols_k = ((nm | (nm >> 16) | (nm >> 32) | (nm >> 48)) & 0xFFFFull) * 0x0001000100010001ull;
        if ((nan_cols_k >> lane) & 1)
            ll = NAN; // a NaN weight vector: math.log(nan), covest/models.py:105
        if ((dead[k] >> lane) & 1)
            ll = isnan(ll) ? ll : -INFINITY; // h * -inf summed with finite terms
        ll += __shfl_xor(ll, 16, kWave);
        ll += __shfl_xor(ll, 32, kWave);
        CompSum sp = spacc[(TAIL && !SPO) ? k : 0];
        if (TAIL && !SPO) {
#pragma unroll
            for (int off = 16; off <= 32; off <<= 1) {
                const double ohi = __shfl_xor(sp.hi, off, kWave);
                const double olo = __shfl_xor(sp.lo, off, kWave);
                double e;
                two_sum(sp.hi, ohi, sp.hi, e);
                sp.lo += olo + e;
            }
        }
        if (lane < 16) {
            const int at = (wave * MU + k) * 16 + lane;
            part_ll[at] = ll;
            if (TAIL && !SPO) {
                part_hi[at] = sp.hi;
                part_lo[at] = sp.lo;
            }
        }
    }
    __syncthreads();
    // one thread per (wave, unit, q) entry of a half-0 unit; it looks up the half-1 partner
    auto finish_entry = [&](int e, int qt, int e_half, int e_cont, int pslot, int qo_early) __attribute__((always_inline)) {
        const int c = e & 15;
        if (qt < 0 || e_half != 0 || e_cont)
            return;
        // the unit with the same tile and half 1 (always in the same workgroup): named by the host (tiles.h unit_pair;
        // until round 4 every thread searched the 48 slots for it, three dependent loads a slot)
        const int pe = pslot >= 0 ? pslot * 16 + c : -1;
        // + the constant of the rows' scales (see the kernel's header): sum_j h_j ln((k0-1)!/(k0+b)!) over this
        // workgroup's items
        const double ll = (part_ll[e] + (pe >= 0 ? part_ll[pe] : 0.0)) + lconst;
        // the units handed back for this weight vector: this half-0 slot's record and its half-1 partner's
        unsigned u_first = sub_rec[2 * e], u_end = sub_rec[2 * e + 1];
        if (pe >= 0) {
            u_first = min(u_first, sub_rec[2 * pe]);
            u_end = max(u_end, sub_rec[2 * pe + 1]);
        }
        const unsigned long long word = u_end ? sub_word(u_first, u_end - 1, true) : 0ull;
        double tail_term = 0.0;
        double hi = 0.0, lo = 0.0;
        if (TAIL) {
            hi = part_hi[e];
            lo = part_lo[e];
            if (pe >= 0 && !SPO) { // (SPO: the half-0 unit alone contracted S)
                double err;
                two_sum(hi, part_hi[pe], hi, err);
                lo += part_lo[pe] + err;
            }
        }
        const int32_t qo = kEntriesEarly ? qo_early : plan.q_orig[qt * 16 + c];
        if (list_mode == 1) { // a key segment of a point: {LL part, sp_j part (hi, lo)}; the host adds the segments
            if (qo >= 0) {
                double *o = plan.partial + ((int64_t)ce * n_seg + seg) * 4;
                o[0] = finite ? ll : NAN;
                o[1] = hi;
                o[2] = lo;
                o[3] = __longlong_as_double((long long)(finite ? word : 0ull)); // (bits; the host merges the segments' words)
            }
            return;
        }
        if (TAIL) {
            double s = hi + lo;
            if (!(s < 1.0))
                s = 1.0; // min(1, fsum(...)), NaN -> 1
            if (s < 1.0)
                tail_term = m.tail * log(1.0 - s);
        }
        if (qo >= 0) {
            const int64_t flat = ce * plan.n_q + qo;
            if (flat >= plan.flat_begin && flat < plan.flat_end) {
                const double v = finite ? ll + tail_term : NAN;
                out_ll[flat - plan.flat_begin] = v;
                if (word != 0 && isfinite(v)) // (rare; -inf stays -inf whatever the keys are worth)
                    sub_list.push(flat - plan.flat_begin, word);
            }
        }
    };
    if (kEntriesEarly) {
#pragma unroll
        for (int u = 0; u < kEntries; ++u)
            if (tid + u * NT < NW * MU * 16)
                finish_entry(tid + u * NT, en_qt[u], en_half[u], en_cont[u], en_pair[u], en_qo[u]);
    } else {
        for (int e = tid; e < NW * MU * 16; e += NT) {
            const int at = wave_block(e / (MU * 16)) * MU + (e / 16) % MU;
            // (the entry's four words fetched together: behind `||` each waited for the one before)
            const int qt = plan.unit_tile[at], e_half = plan.unit_half[at], e_cont = plan.unit_cont[at];
            finish_entry(e, qt, e_half, e_cont, plan.unit_pair[at], 0);
        }
    }
#ifdef COVEST_DIAG
    if (plan.diag && (plan.skip_phases & 0x40000) && lane == 0) {
        long long *d = plan.diag + ((int64_t)(blockIdx.x * gridDim.y + blockIdx.y) * NW + wave) * 8;
        const long long end = (long long)clock64();
        d[0] = dgx_1 - dgx_0;
        d[1] = dgx_2 - dgx_1;
        d[2] = dgx_3 - dgx_2;
        d[3] = dgx_4 - dgx_3;
        d[4] = end - dgx_4;
        d[5] = end - dgx_0;
        d[6] = dgx_0; // (absolute: a workgroup's start against the others')
        d[7] = end;
    }
#endif
}

#ifndef COVEST_FACTORED_VARIANT // (the finishing kernels live in the common translation unit only)
// One wave finishes ONE point whose p_j lie in HBM, summed over chunks of copy numbers (tiles.h list modes 2, 3):
// LL = sum_j h_j log p_j + tail log(1 - sp), covest/models.py:100-107.  read_pj(row): the p_j of row `row` of the
// items (tiles.h: a key, or the sum over a count-less tile), TIMES kShareScale.  A subnormal p_j at a key with h_j != 0
// -- down to kZeroSteps of a grid step, below which it is 0 in the reference whatever the roundings (direct_point.h) -- is
// replaced on the spot by its strict evaluation (direct_point.h: the whole wave, K-direct's arithmetic).  par: the
// point's parameters, clamped; every lane returns the value.
template <class ReadPj>
__device__ __forceinline__ double finish_point(const DevModel &m, const TileView &tv, const double *par, int T,
                                               ReadPj read_pj)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t n_rows = (int64_t)tv.n_items * kTileBins;
    double ll = 0.0;
    bool dead = false, poisoned = false;
    CompSum sp = {0.0, 0.0};
    for (int64_t base = 0; base < n_rows; base += kWave) { // wave-uniform trip count
        const int64_t row = base + lane;
        const bool valid = row < n_rows;
        const double pj_scaled = valid ? read_pj(row) : 0.0;
        double pj = pj_scaled * (1.0 / kShareScale); // (one rounding, onto the doubles' grid)
        const double h = valid ? tv.item_cnt[row] : 0.0;
        uint64_t sub = __ballot(h != 0.0 && pj_scaled > zero_steps_scaled(kShareScale) &&
                                pj < 2.2250738585072014e-308); // a subnormal p_j, or one the product above flushed
        while (sub) { // wave-uniform, rare
            const int who = __builtin_ctzll(sub);
            sub &= sub - 1;
            const int64_t r = base + who;
            const int item = (int)(r / kTileBins);
            const int bin = tv.row_bin[(int64_t)tv.item_first[item] * kTileBins + (r - (int64_t)item * kTileBins)];
            const double strict = strict_pj_wave<5>(m, par, T, m.bins.key[bin], -m.bins.lgam[bin]);
            if (lane == who)
                pj = strict;
        }
        if (m.tail != 0.0)
            sp.add(pj);
        if (h != 0.0) {
            dead |= pj <= 0.0; // utils.safe_log
            poisoned |= pj != pj; // a NaN parameter: NaN, as in the reference (math.log(nan))
            ll = fma(h, log(pj > 0.0 ? pj : 1.0), ll);
        }
    }
    ll = wave_sum(ll);
    double tail_term = 0.0;
    if (m.tail != 0.0) {
        double s = wave_comp_sum(sp);
        if (!(s < 1.0))
            s = 1.0;
        if (s < 1.0)
            tail_term = m.tail * log(1.0 - s);
    }
    if (__ballot(dead))
        ll = isnan(ll) ? ll : -INFINITY;
    double v = ll + tail_term;
    if (__ballot(poisoned) || !(isfinite(par[0]) && isfinite(par[1])))
        v = NAN;
    return v;
}

// Chunked point list (tiles.h list_mode 2): one wave per point adds the chunks' shares of p_j in chunk order and
// takes the logs.  point_par[5 n_points], point_T[n_points]: the points' parameters and threshold_o.
__global__ __launch_bounds__(kWave) void ll_finish_partials(const DevModel m, const int32_t n_tiles, const int32_t n_items,
                                                           const double *__restrict__ tile_dbl,
                                                           const int32_t *__restrict__ tile_int,
                                                           const double *__restrict__ partial,
                                                           const int32_t *__restrict__ first_item,
                                                           const double *__restrict__ point_par,
                                                           const int32_t *__restrict__ point_T, double *__restrict__ out_ll)
{
    const TileView tv = tile_view_from(n_tiles, n_items, tile_dbl, tile_int);
    const int p = blockIdx.x;
    const int c0 = first_item[p], c1 = first_item[p + 1];
    const int64_t n_rows = (int64_t)n_items * kTileBins;
    double par[kMaxParams];
#pragma unroll
    for (int d = 0; d < kMaxParams; ++d)
        par[d] = point_par[(int64_t)p * kMaxParams + d];
    clamp_point<5>(m, par);
    const double v = finish_point(m, tv, par, point_T[p], [&](int64_t row) {
        double pj = 0.0;
        for (int c = c0; c < c1; ++c)
            pj += partial[(int64_t)c * n_rows + row];
        return pj;
    });
    if (threadIdx.x == 0)
        out_ll[p] = v;
}

// The long weight vectors of a dense grid (tiles.h list_mode 3: threshold_o beyond a workgroup's lanes): one wave
// per ((c, e), slot) takes the logs of the p_j the chunk launches summed into `partial`.
__global__ __launch_bounds__(kWave) void ll_finish_dense(const DevModel m, const int32_t n_tiles, const int32_t n_items,
                                                        const double *__restrict__ tile_dbl,
                                                        const int32_t *__restrict__ tile_int, const PointSource src,
                                                        const double *__restrict__ partial, const int64_t ce_first,
                                                        const int64_t n_cols, const int32_t *__restrict__ q_orig,
                                                        const int64_t n_q, const int64_t flat_end,
                                                        double *__restrict__ out_ll)
{
    const TileView tv = tile_view_from(n_tiles, n_items, tile_dbl, tile_int);
    const int64_t ce_local = blockIdx.x / n_cols, slot = blockIdx.x - ce_local * n_cols;
    const int32_t qo = q_orig[slot];
    if (qo < 0)
        return; // padding column
    const int64_t flat = (ce_first + ce_local) * n_q + qo;
    if (flat < src.flat_begin || flat >= flat_end)
        return; // (ragged block ends)
    double par[kMaxParams];
    int T;
    fetch_point<5>(src, flat - src.flat_begin, par, T);
    clamp_point<5>(m, par);
    const int64_t n_rows = (int64_t)n_items * kTileBins;
    const double *mine = partial + (ce_local * n_cols + slot) * n_rows;
    const double v = finish_point(m, tv, par, T, [&](int64_t row) { return mine[row]; });
    if (threadIdx.x == 0)
        out_ll[flat - src.flat_begin] = v;
}

#endif // COVEST_FACTORED_VARIANT

} // namespace

// One instantiation per translation unit (COVEST_FACTORED_VARIANT, see the end of this file): the HIP runtime
// loads a translation unit's code object when one of its kernels is first launched, and a process that evaluates a
// plain dense grid should not pay for the seven other variants (75-98 KB of code each).
template <int NT, bool TAIL, bool PLAIN, int LDC = 0>
hipError_t launch_ll_factored_variant(const DevModel &m, const TileView &tv, const FactoredPlan &plan,
                                      double *out_ll, const SubList &sub_list, hipStream_t stream)
{
    constexpr int HU = kHalfUnits;
    // + 64 zeroed doubles: the last piece of a unit may run a few (masked, weight 0) steps past the end
    // of a G row; what it reads there must be finite
    // (the per-q combine at the end reuses the buffer for three [waves][slots][16] arrays)
    const size_t lds = std::max((size_t)plan.n_buf * kTileBins * plan.ld + 64, (size_t)3 * (NT / kWave) * 2 * HU * 16) *
                       sizeof(double);
    // the dynamic-LDS ceiling is a per-device attribute of the kernel: raise it once per device
    static size_t configured[64] = {0};
    static std::mutex configured_lock; // two model handles may be used from two host threads
    std::lock_guard<std::mutex> guard(configured_lock);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64)
        dev = 0;
    if (lds > configured[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&ll_factored_kernel<NT, HU, TAIL, PLAIN, LDC>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess)
            return e;
        configured[dev] = lds;
    }
    // HIP wraps a grid of more than 2^32 threads silently: at most 2^22 workgroups per launch
    const int64_t per_launch = std::max<int64_t>(1, ((int64_t)1 << 22) / plan.n_qblocks);
    for (int64_t first = plan.ce_begin; first < plan.ce_end; first += per_launch) {
        FactoredPlan part = plan;
        part.ce_begin = first;
        part.ce_end = std::min(plan.ce_end, first + per_launch);
        const dim3 grid((unsigned)((part.ce_end - part.ce_begin) * (plan.list_mode ? plan.n_seg : 1)),
                        (unsigned)plan.n_qblocks);
        hipLaunchKernelGGL((ll_factored_kernel<NT, HU, TAIL, PLAIN, LDC>), grid, dim3(NT), lds, stream, m, tv.n_tiles, tv.n_items,
                           tv.dbl_base, tv.int_base, part, out_ll, sub_list);
    }
    return hipGetLastError();
}

#ifdef COVEST_FACTORED_VARIANT
// this translation unit holds ONE variant: bit 2 = 512 threads (else 256), bit 1 = TAIL, bit 0 = PLAIN; 8 and 9: 512
// threads, PLAIN, the row stride kLdWide at compile time, bit 0 = TAIL
#if COVEST_FACTORED_VARIANT >= 8
template hipError_t launch_ll_factored_variant<512, (COVEST_FACTORED_VARIANT & 1) != 0, true, kLdWide>(
    const DevModel &, const TileView &, const FactoredPlan &, double *, const SubList &, hipStream_t);
#else
template hipError_t launch_ll_factored_variant<(COVEST_FACTORED_VARIANT & 4) ? 512 : 256, (COVEST_FACTORED_VARIANT & 2) != 0,
                                               (COVEST_FACTORED_VARIANT & 1) != 0>(const DevModel &, const TileView &,
                                                                                   const FactoredPlan &, double *,
                                                                                   const SubList &, hipStream_t);
#endif
#else
// the common translation unit: the finishing kernels and the dispatcher; the variants are linked in
extern template hipError_t launch_ll_factored_variant<256, false, false>(const DevModel &, const TileView &, const FactoredPlan &, double *, const SubList &, hipStream_t);
extern template hipError_t launch_ll_factored_variant<256, false, true>(const DevModel &, const TileView &, const FactoredPlan &, double *, const SubList &, hipStream_t);
extern template hipError_t launch_ll_factored_variant<256, true, false>(const DevModel &, const TileView &, const FactoredPlan &, double *, const SubList &, hipStream_t);
extern template hipError_t launch_ll_factored_variant<256, true, true>(const DevModel &, const TileView &, const FactoredPlan &, double *, const SubList &, hipStream_t);
extern template hipError_t launch_ll_factored_variant<512, false, false>(const DevModel &, const TileView &, const FactoredPlan &, double *, const SubList &, hipStream_t);
extern template hipError_t launch_ll_factored_variant<512, false, true>(const DevModel &, const TileView &, const FactoredPlan &, double *, const SubList &, hipStream_t);
extern template hipError_t launch_ll_factored_variant<512, true, false>(const DevModel &, const TileView &, const FactoredPlan &, double *, const SubList &, hipStream_t);
extern template hipError_t launch_ll_factored_variant<512, true, true>(const DevModel &, const TileView &, const FactoredPlan &, double *, const SubList &, hipStream_t);
extern template hipError_t launch_ll_factored_variant<512, false, true, kLdWide>(const DevModel &, const TileView &, const FactoredPlan &, double *, const SubList &, hipStream_t);
extern template hipError_t launch_ll_factored_variant<512, true, true, kLdWide>(const DevModel &, const TileView &, const FactoredPlan &, double *, const SubList &, hipStream_t);

namespace {

template <int NT>
hipError_t launch_nt(const DevModel &m, const TileView &tv, const FactoredPlan &plan, double *out_ll,
                     const SubList &sub_list, hipStream_t stream)
{
    const bool plain = plan.list_mode == 0 && plan.n_pass == 1;
    if (NT == 512 && plain && plan.ld == kLdWide) // the widest double-buffered shape: its row stride at compile time
        return m.tail != 0.0 ? launch_ll_factored_variant<512, true, true, kLdWide>(m, tv, plan, out_ll, sub_list, stream)
                             : launch_ll_factored_variant<512, false, true, kLdWide>(m, tv, plan, out_ll, sub_list, stream);
    if (m.tail != 0.0)
        return plain ? launch_ll_factored_variant<NT, true, true>(m, tv, plan, out_ll, sub_list, stream)
                     : launch_ll_factored_variant<NT, true, false>(m, tv, plan, out_ll, sub_list, stream);
    return plain ? launch_ll_factored_variant<NT, false, true>(m, tv, plan, out_ll, sub_list, stream)
                 : launch_ll_factored_variant<NT, false, false>(m, tv, plan, out_ll, sub_list, stream);
}

} // namespace

hipError_t launch_ll_finish_partials(const DevModel &m, const TileView &tv, const double *partial,
                                     const int32_t *first_item, const double *point_par, const int32_t *point_T,
                                     int64_t n_points, double *out_ll, hipStream_t stream)
{
    if (n_points <= 0)
        return hipSuccess;
    hipLaunchKernelGGL(ll_finish_partials, dim3((unsigned)n_points), dim3(kWave), 0, stream, m, tv.n_tiles, tv.n_items, tv.dbl_base,
                       tv.int_base, partial, first_item, point_par, point_T, out_ll);
    return hipGetLastError();
}

hipError_t launch_ll_finish_dense(const DevModel &m, const TileView &tv, const PointSource &src, const double *partial,
                                  int64_t ce_first, int64_t n_ce, int64_t n_cols, const int32_t *q_orig, int64_t n_q,
                                  int64_t flat_end, double *out_ll, hipStream_t stream)
{
    // HIP wraps a grid of more than 2^32 threads silently: at most 2^24 waves per launch
    const int64_t per_launch = std::max<int64_t>(1, ((int64_t)1 << 24) / n_cols);
    for (int64_t first = 0; first < n_ce; first += per_launch) {
        const int64_t cnt = std::min(per_launch, n_ce - first);
        hipLaunchKernelGGL(ll_finish_dense, dim3((unsigned)(cnt * n_cols)), dim3(kWave), 0, stream, m, tv.n_tiles, tv.n_items,
                           tv.dbl_base, tv.int_base, src, partial + first * n_cols * (int64_t)tv.n_items * kTileBins,
                           ce_first + first, n_cols, q_orig, n_q, flat_end, out_ll);
    }
    return hipGetLastError();
}

hipError_t launch_ll_factored(const DevModel &m, const TileView &tv, const FactoredPlan &plan,
                              double *out_ll, const SubList &sub_list, hipStream_t stream)
{
    if (plan.ce_end <= plan.ce_begin)
        return hipSuccess;
    if (m.kind != 1 || plan.n_columns > plan.n_threads || 8 * plan.n_pass < m.n_err)
        return hipErrorInvalidValue;
    if (plan.n_threads == 256 && plan.half_units == 3)
        return launch_nt<256>(m, tv, plan, out_ll, sub_list, stream);
    if (plan.n_threads == 512 && plan.half_units == 3)
        return launch_nt<512>(m, tv, plan, out_ll, sub_list, stream);
    return hipErrorInvalidValue;
}
#endif // COVEST_FACTORED_VARIANT

} // namespace covest
