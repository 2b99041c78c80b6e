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
// atomics, so the result is deterministic.  (Also here: ll_fix_list_kernel, the
// pass that patches the points a recurrence kernel handed back, direct_point.h.)
#include <hip/hip_runtime.h>

#include <algorithm>

#include "direct_point.h"
#include "fastmath.h"
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

__global__ __launch_bounds__(256) void argmin_stage1(const double *__restrict__ ll, int64_t n,
                                                     double *__restrict__ pv, int64_t *__restrict__ pi)
{
    Cand c;
    c.v = INFINITY;
    c.i = INT64_MAX;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const double v = -ll[i];
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

// The hand-back of the recurrence kernels (direct_point.h): the queued points' values are corrected in place.
// `fast` (the recurrence kernel's value, in which every p_j below p_clamp counted as p_clamp) gets, for every counted
// row of the range named in the side word whose STRICT p_j is below p_clamp, h_j (safe_log(p_j) - log(p_clamp)) added,
// in ascending row order.  The strict p_j is K-direct's (direct_point.h) to the letter: LANE r of a wave keeps the
// p_j of row r of a 64-row chunk; the mixture components are prepared 64 at a time (one per lane) and broadcast
// through the scalar unit, every term rounded to a double on its own, error classes inside, copy numbers outside, both
// ascending.  A point takes ONE wave (NW = 1), four points a workgroup.  (Until round 5 a repeats-model point was shared
// by the 4 waves of a workgroup -- NW = 4: lots of copy numbers dealt to them in turn, the partial p_j added through LDS,
// where it matters, p_j subnormal, every sum is exact whatever the order.  But the points that are handed back are the
// ones whose LAST keys underflow, which are the ones with a SMALL threshold_o -- three to six lots, most of them out
// of the rows' reach -- and every one of the four waves repeated the point's loads, its two pows and the rows' bins:
// C3's 2 962 queued points 46.8 us with four waves a point, 35.8 us with one, profiles/r05_c3_kstat_fix_one_wave_a_point.txt.)
// Launched after every K-basic / K-factored launch, before anything reads the values; with an empty queue it costs
// a launch and one load.  The queued points of a wide grid come in clusters (whole (c, e) rows of it) and the work
// of one grows with its threshold_o, which is why they are compacted into a queue and spread over the chip instead
// of being patched by whichever thread meets them.  The queue's counter is reset by whoever runs next on the
// stream: the arg-min pass (grids) or the host (point lists).
// NW: waves that share a point.
// (Measured and not kept, round 5: the kernel held to 96 registers for five waves a SIMD -- it takes 155, three waves --
// spills 27 of them and is slower, 42.9 against 35.8 us on C3's 2 962 queued points (threshold_o 11 .. 87, 27 rows
// each); to 128 for four waves, 36.4.  The launch is one trip of every wave: by the counters a point is 2 750 vector
// instructions at 17 cycles apiece -- the two pows of its rates, the preparation of the one or two lots of copy numbers
// that reach its rows, an exp per component kept -- profiles/r05_c3_kstat_fix_occupancy_not_kept.txt.)
template <int P, int NW>
__global__ __launch_bounds__(256) void ll_fix_list_kernel(const DevModel m, const int32_t n_tiles, const int32_t n_items,
                                                          const double *__restrict__ tile_dbl,
                                                          const int32_t *__restrict__ tile_int, const PointSource src,
                                                          double *__restrict__ ll, const SubList list)
{
    constexpr int PPB = 4 / NW; // points per workgroup
    __shared__ double pj_part[4][kWave];
    const TileView tv = tile_view_from(n_tiles, n_items, tile_dbl, tile_int);
    const int lane = threadIdx.x & (kWave - 1);
    const int wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int wave = wave_in_block % NW; // among the waves of its point
    const unsigned count = __builtin_amdgcn_readfirstlane(*list.count);
    if (count == 0)
        return; // (workgroup-uniform) the common case: a launch and one load
    // (round 4) the preparation of a lot -- ln x, the normaliser's two logs, the copy number's weight -- was most of this
    // kernel: the device library's log (72 issue slots) three times and its pow (210) once per component.  The logs go
    // through the fast_log table (absolute error 2e-16, what the recurrence kernels' anchors are made with), the weight
    // by squaring.
    __shared__ __attribute__((aligned(16))) double log_tab[kLogTableDoubles];
    load_log_table(log_tab);
    __syncthreads();
    const int S = m.n_err;
    const int OT = kWave / S; // copy numbers prepared per lot of 64 components
    const int s = lane % S;
    const int og = lane / S;
    const bool lane_in_tile = og < OT;
    const double comb_s = m.comb[s];
    for (unsigned at0 = blockIdx.x * PPB; at0 < count; at0 += gridDim.x * PPB) { // workgroup-uniform
        const unsigned at = at0 + wave_in_block / NW;
        if (NW == 1 && at >= count)
            continue; // (wave-uniform; with NW == 1 nothing below synchronises the workgroup)
        const int64_t pt = list.index[at];
        const unsigned long long word = list.word[at];
        double par[kMaxParams];
        int T;
        fetch_point<P>(src, pt, par, T);
        clamp_point<P>(m, par);
        const bool units16 = sub_units16(word);
        const int64_t row_first = units16 ? (int64_t)sub_first(word) * 16 : (int64_t)sub_first(word);
        const int64_t row_last = units16 ? (int64_t)sub_last(word) * 16 + 15 : (int64_t)sub_last(word);
        // (the products the recurrence kernels form, point_fetch.h error_class_rate_mul: a few 1e-16 relative from the
        // pow-made rates of K-direct -- far below the grain of the subnormal terms this kernel exists for -- and 40
        // instructions a point instead of the two pows' 420)
        const double lam = error_class_rate_mul(m, par[0], par[1], s, S);
        const int o_hi = T;
        double value = ll[pt];
        for (int64_t chunk = row_first; chunk <= row_last; chunk += kWave) { // workgroup-uniform
            const int64_t row = chunk + lane;
            const int bin = (row <= row_last && row < (int64_t)tv.n_tiles * kTileBins) ? tv.row_bin[row] : -1;
            const double h = bin >= 0 ? m.bins.cnt[bin] : 0.0;
            const bool counted = bin >= 0 && h != 0.0;
            const double key = counted ? m.bins.key[bin] : 0.0;
            const double nlg = counted ? -m.bins.lgam[bin] : 0.0;
            double pj = 0.0; // of this lane's row: this wave's share of the copy numbers
            const uint64_t cm = __ballot(counted);
            if (cm) {
                // the chunk's smallest and largest counted key (rows ascend with the keys)
                const int l_lo = __builtin_ctzll(cm), l_hi = 63 - __builtin_clzll(cm);
                const double k_lo = wave_bcast(key, l_lo), g_lo = wave_bcast(nlg, l_lo);
                const double k_hi = wave_bcast(key, l_hi), g_hi = wave_bcast(nlg, l_hi);
                // lots of OT copy numbers, dealt to the point's waves in turn
                for (int o0 = 1 + wave * OT; o0 < o_hi; o0 += NW * OT) {
                    // ---- lane-parallel preparation of up to OT * S mixture components (as K-direct) ----
                    const int o = o0 + og;
                    const bool live = lane_in_tile && o < o_hi;
                    const double x = (double)o * lam;
                    {
                        // A whole lot out of reach of the chunk (the usual case) is not prepared at all: the same
                        // test as below in single precision, with the normaliser's floor  D(x) >= x - 19  for
                        // x >= 1 (point_fetch.h: x + ln(1 - e^-xr), xr > 1e-8) and room for the float error of
                        // key * ln x (<= 0.02 at the key cap).
                        const float lxf = __logf((float)x);
                        const float reach = fmaxf(fmaf((float)k_lo, lxf, (float)g_lo), fmaf((float)k_hi, lxf, (float)g_hi));
                        const bool far = x >= 1.0 && (x < k_lo - 1.0 || x > k_hi + 1.0) &&
                                         (double)reach - (x - 19.0) < -745.5;
                        if (!__any(live && !far))
                            continue; // wave-uniform
                    }
                    const double n_os = comb_s * (1.0 - exp_neg_rn(x));
                    double tot = 0.0;
                    for (int t = 0; t < S; ++t)
                        tot += __shfl(n_os, og * S + t, kWave);
                    if (tot == 0.0)
                        tot = 1.0;
                    double a_os = n_os / tot;
                    const double b_o = (P == 5) ? copy_number_weight_by_squaring(par[2], par[3], par[4], o) : 1.0;
                    double lx = 0.0, nd = -INFINITY;
                    if (live && x > 0.0) {
                        lx = fast_log(x, log_tab);
                        nd = -log_trunc_norm(x, lx, log_tab);
                    }
                    if (!live)
                        a_os = 0.0;
                    // Which components reach any row of the chunk at all?  exp(arg) rounds to 0 below ln 2^-1075
                    // = -745.13, and arg is concave in the key with its top within 1 of x: outside [k_lo - 1,
                    // k_hi + 1] it is monotone over the chunk's keys and the nearer end bounds it.  (Most copy
                    // numbers, for the keys of a deep tail: their terms are exactly 0 in the reference too.)
                    const bool inside = x >= k_lo - 1.0 && x <= k_hi + 1.0;
                    const double top = fmax(fma(k_lo, lx, nd + g_lo), fma(k_hi, lx, nd + g_hi));
                    const uint64_t keep = __ballot(a_os != 0.0 && (inside || top >= -745.2));
                    // ---- every lane accumulates the kept ones for its own row ----
                    // inner = the classes of one copy number in ascending order, pj += b_o * inner per copy number that
                    // has any (covest/models.py:237) -- the TERMS four at a time (round 5): an exp is a chain of forty-five
                    // dependent instructions, and one wave a SIMD (a launch of this kernel is one trip of every wave) does
                    // not hide one behind another unless they are written side by side.  The sums are the same sums: a
                    // batch's terms are added one by one, in order, to the copy number they belong to.
                    {
                        uint64_t km = keep;
                        int cur_end = 0; // one past the last lane of the copy number `inner` belongs to (0: none yet)
                        double inner = 0.0;
                        while (km) { // wave-uniform
                            int idx[4];
                            int n = 0;
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                idx[u] = km ? __builtin_ctzll(km) : idx[0]; // (a short batch repeats its first term and drops it)
                                n += km ? 1 : 0;
                                km &= km - 1; // (0 stays 0)
                            }
                            double t[4];
#pragma unroll
                            for (int u = 0; u < 4; ++u)
                                t[u] = wave_bcast(a_os, idx[u]) * exp(fma(key, wave_bcast(lx, idx[u]), wave_bcast(nd, idx[u]) + nlg));
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                if (u >= n)
                                    break;
                                if (idx[u] >= cur_end) { // the first kept class of another copy number
                                    if (cur_end > 0)
                                        pj += wave_bcast(b_o, cur_end - S) * inner;
                                    inner = 0.0;
                                    while (idx[u] >= cur_end)
                                        cur_end += S;
                                }
                                inner += t[u];
                            }
                        }
                        if (cur_end > 0)
                            pj += wave_bcast(b_o, cur_end - S) * inner;
                    }
                }
            }
            if (NW > 1) {
                pj_part[wave][lane] = pj;
                __syncthreads();
                pj = ((pj_part[0][lane] + pj_part[1][lane]) + pj_part[2][lane]) + pj_part[3][lane];
            }
            const bool fix = counted && pj < list.p_clamp;
            const double contrib = fix ? h * ((pj <= 0.0 ? -INFINITY : log(pj)) - list.log_p_clamp) : 0.0;
            uint64_t todo = __ballot(fix);
            while (todo) { // ascending rows (every wave computes the same)
                const int kk = __builtin_ctzll(todo);
                todo &= todo - 1;
                value += wave_bcast(contrib, kk);
            }
            if (NW > 1)
                __syncthreads(); // pj_part is rewritten by the next chunk
        }
        if (lane == 0 && wave == 0)
            ll[pt] = value;
    }
}

// The same hand-back for the BASIC model, several points a wave (round 4).  A basic-model point has ONE copy number: its
// S mixture components are prepared by S lanes, and ll_fix_list_kernel<2, 1> above left the other 64 - S idle through
// the whole preparation (exp_neg_rn, a division, three logs) and then used a dozen of its 64 lanes for the point's
// dozen rows -- 1 900 instructions a point, 28 us of C2's 238 us step.  Here a wave takes G = 64 / S queued points at
// once: lane (g, s) prepares component s of point g, then stands for row s of a pass of S rows of point g, the
// components reaching it through the lanes' crossbar (the group's own, in ascending s).  The arithmetic of a row is the
// one above to the letter -- the terms a_s exp(key ln x_s - D_s - ln key!) added in ascending s (a component that is out
// of reach adds an exact 0: the kernel above skips it, which is the same), b_o = 1, the contributions of a point's rows
// in ascending order.
// Round 5.  C2 hands 7 266 of its 10^6 points back: 908 waves, fewer than the chip has SIMDs, so the launch lasts as
// long as ONE wave's chain of dependent loads and calls -- 16.8 us of a 188 us step.  What shortened it: the class count
// as a compile-time 8 (SC), the loops over the classes unrolled so that a row's eight exps and their crossbar reads
// interleave instead of following one another -- 16.5 -> 14.8 us (profiles/r05_c2_kstat_fix_classes_unrolled.txt).
// What did not, and stays because it is less code in flight: the class's rate by multiplication
// (error_class_rate_mul: the very products K-basic itself forms, point_fetch.h) instead of two calls of the device
// library's pow; a queue entry's loads -- the entry, the point's axis values and value, the rows' bins of the first
// pass -- issued together with the log table's, before its barrier, and the rows of the next pass during this one
// (16.93 -> 16.85 us).  Measured and not kept: a wave a point with lane (row, component) holding ONE term -- an exp a lane
// and pass instead of S -- is eight times the waves, each repeating the preparation: 36.5 us,
// profiles/r05_c2_kstat_fix_wave_per_point_not_kept.txt.
template <int SC> // the class count as a constant: every loop over the classes unrolled, a row's exps interleaved
__global__ __launch_bounds__(256) void ll_fix_basic_packed_kernel(const DevModel m, const int32_t n_tiles, const int32_t n_items,
                                                                  const double *__restrict__ tile_dbl,
                                                                  const int32_t *__restrict__ tile_int, const PointSource src,
                                                                  double *__restrict__ ll, const SubList list)
{
    const unsigned count = __builtin_amdgcn_readfirstlane(*list.count);
    if (count == 0)
        return; // (workgroup-uniform) the common case: a launch and one load
    __shared__ __attribute__((aligned(16))) double log_tab[kLogTableDoubles];
    load_log_table(log_tab);
    const TileView tv = tile_view_from(n_tiles, n_items, tile_dbl, tile_int);
    const int lane = threadIdx.x & (kWave - 1);
    constexpr int S = SC;             // m.n_err: 8, 16, 24 or 32 (padded: comb = 0 beyond the model's classes)
    constexpr int G = kWave / S;      // points a wave takes at once
    const int g = lane / S, s = lane - g * S;
    const bool in_group = g < G;
    const int first_lane = (in_group ? g : 0) * S; // the group's lane 0 (idle lanes shadow group 0 and store nothing)
    const double comb_s = m.comb[s];
    const int64_t n_rows_table = (int64_t)tv.n_tiles * kTileBins;
    const unsigned wave_global = blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave;
    const unsigned n_waves = gridDim.x * (blockDim.x / kWave);
    // A queue entry and what hangs on it by loads alone: the point, its value, this lane's row of the first pass
    // {h_j, key, -ln key!} (h = 0: no such row, or no count)
    bool have = false;
    int64_t pt = 0, row_first = 0, row_last = -1;
    double par[kMaxParams] = {0, 0, 0, 0, 0};
    int T = 0;
    double value = 0.0, h_n = 0.0, key_n = 0.0, nlg_n = 0.0;
    auto fetch_row = [&](int64_t r0, double &h, double &key, double &nlg) {
        const int64_t row = row_first + r0 + s;
        const int bin = (have && row <= row_last && row < n_rows_table) ? tv.row_bin[row] : -1;
        h = bin >= 0 ? m.bins.cnt[bin] : 0.0;
        key = (bin >= 0 && h != 0.0) ? m.bins.key[bin] : 0.0;
        nlg = (bin >= 0 && h != 0.0) ? -m.bins.lgam[bin] : 0.0;
    };
    auto fetch_entry = [&](unsigned base) {
        const unsigned at = base + (unsigned)(in_group ? g : 0);
        have = in_group && at < count;
        pt = list.index[have ? at : base];
        const unsigned long long word = have ? list.word[at] : 0ull;
        const bool units16 = sub_units16(word);
        row_first = units16 ? (int64_t)sub_first(word) * 16 : (int64_t)sub_first(word);
        row_last = units16 ? (int64_t)sub_last(word) * 16 + 15 : (int64_t)sub_last(word);
        fetch_row(0, h_n, key_n, nlg_n);
        fetch_point<2>(src, pt, par, T);
        value = have ? ll[pt] : 0.0;
    };
    const unsigned base0 = wave_global * (unsigned)G;
    if (base0 < count) // (wave-uniform) the first entry's loads and the table's are in flight together
        fetch_entry(base0);
    __syncthreads(); // the table is readable
    for (unsigned base = base0; base < count; base += n_waves * (unsigned)G) { // wave-uniform
        if (base != base0)
            fetch_entry(base);
        clamp_point<2>(m, par);
        // ---- component s of point g: covest/models.py:85-90, as K-basic prepares it (ll_basic.hip) ----
        const double x = error_class_rate_mul(m, par[0], par[1], s, S); // o = 1
        const bool live = have && T > 1;
        const double n_os = comb_s * (1.0 - exp_neg_rn(x));
        double tot = 0.0;
#pragma unroll
        for (int t = 0; t < S; ++t) // naive sum in s order
            tot += __shfl(n_os, first_lane + t, kWave);
        if (tot == 0.0)
            tot = 1.0;
        double a_s = n_os / tot;
        double lx = 0.0, nd = -INFINITY;
        if (live && x > 0.0) {
            lx = fast_log(x, log_tab);
            nd = -log_trunc_norm(x, lx, log_tab);
        }
        if (!live)
            a_s = 0.0;
        // ---- the point's rows, S at a time ----
        const int64_t my_rows = have ? row_last - row_first + 1 : 0;
        int64_t most = my_rows;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1)
            most = max(most, __shfl_xor(most, off, kWave)); // wave-uniform trip count
        for (int64_t r0 = 0; r0 < most; r0 += S) {
            const double h = h_n, key = key_n, nlg = nlg_n;
            if (r0 + S < most)
                fetch_row(r0 + S, h_n, key_n, nlg_n);
            const bool counted = h != 0.0;
            double pj = 0.0;
#pragma unroll
            for (int t = 0; t < S; ++t) { // error classes, ascending
                const double a_t = __shfl(a_s, first_lane + t, kWave);
                const double lx_t = __shfl(lx, first_lane + t, kWave);
                const double nd_t = __shfl(nd, first_lane + t, kWave);
                const double term = a_t * exp(fma(key, lx_t, nd_t + nlg));
                pj += (a_t != 0.0) ? term : 0.0; // (a component without weight is left out above: it adds nothing here)
            }
            const bool fix = counted && pj < list.p_clamp;
            const double contrib = fix ? h * ((pj <= 0.0 ? -INFINITY : log(pj)) - list.log_p_clamp) : 0.0;
#pragma unroll
            for (int t = 0; t < S; ++t) { // ascending rows of the group's point
                const double c_t = __shfl(contrib, first_lane + t, kWave);
                const bool f_t = __shfl((int)fix, first_lane + t, kWave) != 0;
                if (f_t)
                    value += c_t;
            }
        }
        if (have && s == 0)
            ll[pt] = value;
    }
}

// The winner where it is wanted: in HBM (the ranks' exchange reads it there) and, when the caller gave one, in a
// page-locked HOST mirror -- the kernel's own store, visible once the stream is synchronised: no copy launch and no
// staging for 16 bytes.
__device__ __forceinline__ void publish(Cand c, int64_t flat_begin, ArgminResult *__restrict__ result,
                                        ArgminResult *__restrict__ host_mirror)
{
    ArgminResult r;
    r.min_negll = c.v;
    r.index = (c.i == INT64_MAX) ? -1 : c.i;
    // the same as a pair of doubles with the GLOBAL flat index, for the cross-GPU exchange (flat indices
    // stay below 2^53)
    r.pair[0] = c.v;
    r.pair[1] = (c.i == INT64_MAX) ? -1.0 : (double)(flat_begin + c.i);
    *result = r;
    if (host_mirror)
        *host_mirror = r;
}

__global__ __launch_bounds__(256) void argmin_stage2(const double *__restrict__ pv,
                                                     const int64_t *__restrict__ pi, int n_part, int64_t flat_begin,
                                                     ArgminResult *__restrict__ result, ArgminResult *__restrict__ host_mirror,
                                                     unsigned *__restrict__ queue_count)
{
    if (threadIdx.x == 0 && queue_count)
        *queue_count = 0; // the hand-back queue of the launch before (drained by ll_fix_list_kernel) starts empty again
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
    if (threadIdx.x == 0)
        publish(c, flat_begin, result, host_mirror);
}

// A small grid (optimize_grid's have a few thousand points, C1 2 500): both stages in ONE workgroup and one launch.
// kSmallThreads threads, every thread's loads issued four at a time (round 5: 256 threads walked their 31 values of a
// 7 776-point grid one dependent load after the other -- 12 us for 62 KB by the trace of an optimize_grid search).
constexpr int kSmallThreads = 1024;

// The values of a small grid, negated, through `take(i, -ll[i])` in ascending i per thread.
template <class Take>
__device__ __forceinline__ void small_grid_values(const double *__restrict__ ll, int64_t n, Take take)
{
    for (int64_t base = threadIdx.x; base < n; base += 4 * kSmallThreads) {
        double v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t i = base + (int64_t)u * kSmallThreads;
            v[u] = i < n ? -ll[i] : INFINITY;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t i = base + (int64_t)u * kSmallThreads;
            if (i < n)
                take(i, v[u]);
        }
    }
}

__global__ __launch_bounds__(kSmallThreads) void argmin_small(const double *__restrict__ ll, int64_t n, int64_t flat_begin,
                                                             ArgminResult *__restrict__ result, ArgminResult *__restrict__ host_mirror,
                                                             unsigned *__restrict__ queue_count)
{
    if (threadIdx.x == 0 && queue_count)
        *queue_count = 0;
    Cand c;
    c.v = INFINITY;
    c.i = INT64_MAX;
    small_grid_values(ll, n, [&](int64_t i, double v) { // ascending i within a thread: strict < keeps the first
        if (v < c.v) {
            c.v = v;
            c.i = i;
        }
    });
    c = block_best(c);
    if (threadIdx.x == 0)
        publish(c, flat_begin, result, host_mirror);
}

// The selection scan of covest/grid.py:65-70 ON THE DEVICE, for the caller that runs it every iteration
// (optimize_grid): started from `start` -- the minimum the search holds when the iteration begins -- the loop
//     if val < min_val: diff += min_val - val; min_val = val; min_args = args
// changes its state exactly at the STRICT RUNNING-MINIMUM RECORDS below `start`, in index order: a handful of points
// once a search is under way.  The kernel lists them -- {flat index, -LL} -- straight into page-locked host memory, and
// the host replays the loop over that list alone (covest_amd/grid.py replay_records): same comparisons, same sums, same
// order.  Until round 5 every iteration copied the whole LL array back (62 KB for 7 776 points) to run the loop there.
// One workgroup (grids up to kArgminSmall points); the values are staged in LDS, every thread owns a contiguous run of
// them: its minimum, an exclusive prefix minimum across the threads, then its own records behind a prefix sum.  A NaN
// never passes `<`.  More than kScanCap records: `truncated`, and the caller reads the array back as before.
__global__ __launch_bounds__(kSmallThreads) void argmin_scan_small(const double *__restrict__ ll, int64_t n, int64_t flat_begin,
                                                                  double start, ArgminResult *__restrict__ result,
                                                                  ArgminResult *__restrict__ host_mirror,
                                                                  ScanRecords *__restrict__ scan, unsigned *__restrict__ queue_count)
{
    constexpr int NW = kSmallThreads / kWave;
    extern __shared__ double vals[]; // [n] -LL
    __shared__ double wave_min[NW];
    __shared__ int wave_cnt[NW];
    const int tid = threadIdx.x, lane = tid & (kWave - 1), w = tid / kWave;
    if (tid == 0 && queue_count)
        *queue_count = 0;
    Cand c;
    c.v = INFINITY;
    c.i = INT64_MAX;
    small_grid_values(ll, n, [&](int64_t i, double v) { // coalesced; ascending i within a thread: strict < keeps the first
        vals[i] = v;
        if (v < c.v) {
            c.v = v;
            c.i = i;
        }
    });
    c = block_best(c); // (ends with a barrier: vals is complete)
    if (tid == 0)
        publish(c, flat_begin, result, host_mirror);
    // every thread a contiguous run of the values; prefix minimum and prefix count across the threads by shuffles
    // inside a wave and a word per wave across them (two barriers in all)
    const int chunk = (int)((n + kSmallThreads - 1) / kSmallThreads);
    const int lo = min((int)n, tid * chunk), hi = min((int)n, lo + chunk);
    double m = INFINITY;
    for (int i = lo; i < hi; ++i)
        m = vals[i] < m ? vals[i] : m;
    double incl = m; // inclusive prefix minimum inside the wave
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        const double o = __shfl_up(incl, off, kWave);
        if (lane >= off)
            incl = o < incl ? o : incl;
    }
    if (lane == kWave - 1)
        wave_min[w] = incl;
    __syncthreads();
    // the running minimum this thread's run starts from: `start`, the waves before, the lanes before
    double run0 = start;
    for (int k = 0; k < w; ++k)
        run0 = wave_min[k] < run0 ? wave_min[k] : run0;
    {
        const double before = __shfl_up(incl, 1, kWave);
        if (lane > 0 && before < run0)
            run0 = before;
    }
    double run = run0;
    int cnt = 0;
    for (int i = lo; i < hi; ++i)
        if (vals[i] < run) {
            run = vals[i];
            ++cnt;
        }
    int pre = cnt; // inclusive prefix count inside the wave
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        const int o = __shfl_up(pre, off, kWave);
        if (lane >= off)
            pre += o;
    }
    if (lane == kWave - 1)
        wave_cnt[w] = pre;
    __syncthreads();
    int at = pre - cnt, total = 0;
    for (int k = 0; k < NW; ++k) {
        if (k < w)
            at += wave_cnt[k];
        total += wave_cnt[k];
    }
    run = run0;
    for (int i = lo; i < hi; ++i)
        if (vals[i] < run) {
            run = vals[i];
            if (at < kScanCap) {
                scan->rec[at].index = flat_begin + i;
                scan->rec[at].negll = run;
            }
            ++at;
        }
    if (tid == 0) {
        scan->start = start;
        scan->truncated = total > kScanCap ? 1 : 0;
        scan->n = min(total, kScanCap);
    }
}

} // namespace

hipError_t launch_argmin_scan(const double *ll, int64_t n, int64_t flat_begin, double start, ArgminResult *result,
                              ArgminResult *host_mirror, ScanRecords *scan, unsigned *queue_count, hipStream_t stream)
{
    if (n > kArgminSmall || n < 1)
        return hipErrorInvalidValue;
    const size_t lds = (size_t)n * sizeof(double);
    static bool raised[64] = {false};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64)
        dev = 0;
    if (!raised[dev]) { // (128 KB of dynamic LDS for the largest grid: above the default ceiling)
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&argmin_scan_small),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kArgminSmall * sizeof(double)));
        if (e != hipSuccess)
            return e;
        raised[dev] = true;
    }
    hipLaunchKernelGGL(argmin_scan_small, dim3(1), dim3(kSmallThreads), lds, stream, ll, n, flat_begin, start, result, host_mirror, scan,
                       queue_count);
    return hipGetLastError();
}

hipError_t launch_ll_fix_list(const DevModel &m, const TileView &tv, const PointSource &src, double *ll,
                              const SubList &list, hipStream_t stream, int64_t n_points)
{
    // enough workgroups to spread a few thousand queued points over the chip -- four points a workgroup, no more of them
    // than the launch before can have queued points for; an empty queue is the common case
    const dim3 grid((unsigned)std::min<int64_t>(2048, std::max<int64_t>(64, n_points > 0 ? (n_points + 3) / 4 : 2048))), block(256);
    // (the packed kernel: 64 / S points a wave; no more workgroups than the launch before can have queued points for -- an
    // optimize_grid search launches this hundreds of times on an empty queue)
    const int packed_blocks = (int)std::min<int64_t>(512, std::max<int64_t>(16, n_points > 0 ? (n_points + 31) / 32 : 512));
    if (m.kind == 0 && m.n_err <= 32 && m.n_err % 8 == 0) {
        auto go = [&](auto kern) {
            hipLaunchKernelGGL(kern, dim3(packed_blocks), block, 0, stream, m, tv.n_tiles, tv.n_items, tv.dbl_base, tv.int_base, src,
                               ll, list);
        };
        switch (m.n_err) {
        case 8: go(ll_fix_basic_packed_kernel<8>); break;
        case 16: go(ll_fix_basic_packed_kernel<16>); break;
        case 24: go(ll_fix_basic_packed_kernel<24>); break;
        default: go(ll_fix_basic_packed_kernel<32>); break;
        }
    } else if (m.kind == 0)
        hipLaunchKernelGGL((ll_fix_list_kernel<2, 1>), grid, block, 0, stream, m, tv.n_tiles, tv.n_items, tv.dbl_base,
                           tv.int_base, src, ll, list);
    else
        hipLaunchKernelGGL((ll_fix_list_kernel<5, 1>), grid, block, 0, stream, m, tv.n_tiles, tv.n_items, tv.dbl_base,
                           tv.int_base, src, ll, list);
    return hipGetLastError();
}

hipError_t launch_argmin(const double *ll, int64_t n, int64_t flat_begin, double *partial_val, int64_t *partial_idx,
                         ArgminResult *result, ArgminResult *host_mirror, unsigned *queue_count, hipStream_t stream)
{
    if (n <= kArgminSmall) {
        hipLaunchKernelGGL(argmin_small, dim3(1), dim3(kSmallThreads), 0, stream, ll, n, flat_begin, result, host_mirror, queue_count);
        return hipGetLastError();
    }
    // (a small grid needs no more workgroups than it has waves of points)
    const int blocks = (int)std::min<int64_t>(kArgminBlocks, std::max<int64_t>(1, (n + 255) / 256));
    // (measured and not kept, round 4: both stages in ONE launch -- every workgroup stores its candidate, adds to a
    // counter behind a fence, the workgroup whose add came last reduces the candidates -- is 20-27 us SLOWER a step than
    // the second launch it saves: 1 024 agent-scope fences cost more than a 4 us launch)
    hipLaunchKernelGGL(argmin_stage1, dim3(blocks), dim3(256), 0, stream, ll, n, partial_val, partial_idx);
    hipLaunchKernelGGL(argmin_stage2, dim3(1), dim3(256), 0, stream, partial_val, partial_idx, blocks, flat_begin, result,
                       host_mirror, queue_count);
    return hipGetLastError();
}

} // namespace covest
