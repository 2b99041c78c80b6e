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

#include <type_traits>

#include "direct_point.h"
#include "fastmath.h"
#include "kernels.h"
#include "point_fetch.h"
#include "streams.h"
#include "wave.h"

namespace covest {

namespace {

// S: error classes, padded to a multiple of 8 (comb = 0 beyond the model's: such a class weighs exactly 0); BD: threads
// per workgroup.  The per-lane anchors of the recurrence live in dynamic LDS: 16 S BD bytes.
template <int S, bool TAIL, int BD>
__device__ __forceinline__ void ll_basic_body(const DevModel &m, const int32_t n_tiles, const int32_t n_items,
                                              const double *__restrict__ tile_dbl, const int32_t *__restrict__ tile_int,
                                              const PointSource &src, const int64_t n, double *__restrict__ out_ll,
                                              const SubList &sub_list)
{
    const TileView tv = tile_view_from(n_tiles, n_items, tile_dbl, tile_int);
    __shared__ __attribute__((aligned(16))) double log_tab[kLogTableDoubles];
    // (Measured and not kept, round 5: the workgroups in another order -- the grid's rows from the last (large c: the
    // longest walks) to the first, 0.1627 against 0.1587 ms; strided through the grid so that cheap and dear rows mix,
    // stride 17: 0.1623, stride 1531: 0.1834.  Presumably: neighbouring workgroups take the same paths through the code
    // and share its cache lines: profiles/r05_c2_ab_block_order_not_kept.txt.)
    const int64_t pt = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = pt < n;
    const int64_t ptc = live ? pt : n - 1; // idle lanes shadow the last point: every lane stays in the wave ops

    // (the point's axis values are asked for BEFORE the log table is copied and its barrier rather than behind it -- one
    // round trip to the cache less at a workgroup's start; no difference that C2's timing shows, round 5)
    double par[kMaxParams];
    int T;
    fetch_point<2>(src, ptc, par, T);
    load_log_table(log_tab);
    __syncthreads();
    clamp_point<2>(m, par);
    const bool finite = isfinite(par[0]) && isfinite(par[1]);

    double lam[S];
    error_class_rates<S>(m, par[0], par[1], lam); // (by multiplication: point_fetch.h)
    // DOOMED WAVES LEAVE HERE (round 4).  A point whose sum is -inf -- 44 % of C2 -- has p_j = 0 at a counted key, and
    // the key where that is decided first is the LAST counted one.  p_j = 0 in the reference iff every class's term is
    // flushed by the extension's cast to double (c_src/covest_poissonmodule.c:32: below 2^-1075 = e^-745.13).  The
    // classes' rates fall with s for e < 3/4 (covest/models.py:76-79: x_{s+1} / x_s = e / (3 (1 - e))), and for
    // x <= j - 1 the truncated pmf x^j / (j! (e^x - 1)) grows with x, so every class's term at the last counted key j is
    // at most the error-free class's, which is at most 1.59 times the plain Poisson term (x >= 1), times what the
    // 200-chunk normaliser can add (point_fetch.h log_trunc_norm: at most 1 / (1 - e^-1e-8) = 1e8).  So
    //     j ln x_0 - x_0 - ln j!  <  -745.13 - ln 1.59 - ln 1e8 - 1.5  =  -765.5
    // is SUFFICIENT for -inf (safe_log, covest/utils.py; a tail term is finite or 0); points between that and the exact
    // decision take the ordinary route below.  A wave whose lanes are all doomed (or NaN) writes its values and ends
    // before the mixture weights, the anchors and the tiles that are walked with all streams: neighbouring points of a
    // grid are doomed together.
    {
        const double jl = tv.last_key[0], x0 = lam[0];
        bool sure = false;
        if (jl > 0.0) // (wave-uniform: the histogram has a counted key)
            sure = finite && par[1] < 0.75 && x0 >= 1.0 && x0 <= jl - 1.0 &&
                   fma(jl, fast_log(x0, log_tab), -(x0 + tv.last_key[1])) < -765.5;
        bool leave = !__any(finite && !sure);
#ifdef COVEST_DIAG
        leave = leave && sub_list.diag_class == 0; // (tools/dump_c2_classes.py wants every point's route)
#endif
        if (leave) { // wave-uniform
            if (live)
                out_ll[pt] = finite ? -INFINITY : NAN;
            return;
        }
    }
    extern __shared__ double anchors[]; // [2S][lane of the workgroup]: conflict-free columns
    StreamSet<S, LdsAnchors<S>> st;
    st.an.mine = anchors + threadIdx.x;
    st.an.stride = BD;
    st.init(m, lam, 1, finite, log_tab, log_tab);

    double acc_ll = 0.0;
    uint64_t dead = 0; // lanes that met a p_j <= 0 with h_j != 0
    unsigned long long subw = 0; // the rows this lane hands back (direct_point.h): first and last
    const double p_clamp = sub_list.p_clamp;
    CompSum acc_sp = {0.0, 0.0};

    // one key's p_j, TIMES 2^64 (tiles.h kBasicShift: the scales of the tile table carry the factor, so the product
    // of the streams' sum and the key's scale does not flush to 0 below half a grid step of the doubles -- the
    // reference's own roundings keep a p_j alive down to a quarter of one, direct_point.h kZeroSteps): into sp_j, and
    // its log, the 2^-64 folded into the log's exponent arithmetic (all branches wave-uniform)
    const double clamp_s = p_clamp * kBasicScale;
    const double zero_s = zero_steps_scaled(kBasicScale);
    // (TAIL: sp_j -- math.fsum of the p_j, covest/models.py:103 -- is summed plainly over the <= 32 keys of a tile,
    // terms of one sign, and the compensated accumulator gets one value per tile: a relative 32 eps at worst, where the
    // p_j themselves carry a few eps each.  Round 5: until then every key went through the two-sum, six dependent
    // instructions a key on ONE chain through the whole walk; the compiler hid that chain behind the logs of many keys
    // at once and spilled 266 registers doing so.)
    double tile_sp = 0.0;
    auto account = [&](double ps, double h, int row) {
        if (TAIL)
            tile_sp += ps; // (filler keys have scale 0: p == 0)
        if (h != 0.0) { // filler keys and zero counts: no log (`if h`, covest/models.py:106)
            // log(max(p_j, p_clamp)): what a p_j deep in the subnormal range contributes is then a known constant,
            // which the strict evaluation of that key replaces later (direct_point.h); p_j = 0 is remembered below
            const double xs[1] = {max_raw(ps, clamp_s)};
            double lg[1];
            fast_log_bits_n<1, 5, kBasicShift>(xs, lg, log_tab);
            acc_ll = fma(h, lg[0], acc_ll);
            // one compare per key for everything out of the ordinary (lanes already dead have nothing to add)
            const uint64_t low = __ballot(ps < clamp_s) & ~dead;
            if (__builtin_expect(low != 0, 0)) { // wave-uniform, cold
                // utils.safe_log: p_j = 0 makes the whole sum -inf -- remembered as a lane mask in SGPRs; decided on
                // the EXACT value (kZeroSteps).  Between that and the clamp: a row for the strict evaluation
                const bool zero = ps <= zero_s; // (NaN: not a zero -- a NaN stays a NaN)
                dead |= __ballot(zero);
                if (!zero && ps < clamp_s) // deep in the subnormal range: name the row (rows come in ascending order)
                    subw = subw ? sub_word(sub_first(subw), (unsigned)row, false) : sub_word((unsigned)row, (unsigned)row, false);
            }
        }
    };

    // One key tile with the streams 0 .. N-1 (N = S: all of them; N = 1: see below).  ll_done (wave-uniform, TAIL only):
    // the tile's share of sum h_j log p_j is accounted for already (the closed form below) -- its keys only enter sp_j.
    auto do_tile = [&](auto n_tag, int t, bool force_start, bool ll_done) __attribute__((always_inline)) {
        constexpr int N = decltype(n_tag)::value;
        const TileRec rc = tv.rec[t]; // (the tile's constants: one scalar load of one cache line, tiles.h)
        const double k0 = rc.k0;
        const int nb = rc.nb;
        // (a tile that only enters sp_j is entered on the SUM window -- streams.h enter_sum_tile: a stream is on where it
        // matters to a sum, e^-60 -- and walked with the classes that are live)
        const bool sums = TAIL && (ll_done || rc.all_zero != 0); // wave-uniform
        const int n_live = sums ? st.template enter_tile_n<N, true>(k0 - 1.0, k0 + (double)(nb - 1), rc.lgam_prev, rc.lgam_last,
                                                                    rc.run_start != 0 || force_start)
                                : st.template enter_tile_n<N>(k0 - 1.0, k0 + (double)(nb - 1), rc.lgam_prev, rc.lgam_last,
                                                              rc.run_start != 0 || force_start); // (behind skipped tiles: anchored afresh)
        const double *scal = tv.scal + (int64_t)t * kTileBins;
        const double *cnt = tv.cnt + (int64_t)t * kTileBins;
        double xx[S]; // squared rates, recomputed per tile (N multiplies) rather than held in 2 S registers
        if (nb == kTileBins) {
#pragma unroll
            for (int s = 0; s < S; ++s) {
                double xs = st.x[s];
                if (TAIL && s < N) // (the empty asm: the squares are made per tile, not hoisted out of the walk and held --
                    asm volatile("" : "+v"(xs)); // with a tail the walk's registers are short: 4 spilled at 3 waves a SIMD)
                xx[s] = s < N ? xs * xs : 0.0;
            }
        }
        if (TAIL && sums) {
            // a tile without a single count (they exist only with a tail), or one whose logs the closed form has
            // taken care of: its keys take no log, only their p_j enter sp_j (covest/models.py:103) -- add them up
            // plainly (32 terms of one sign) and hand the compensated accumulator ONE value per tile.  With the classes
            // that are LIVE (n_live: the ones above are exact zeros in every lane of the wave): the long count-less
            // stretch between the error k-mers and the genomic peak of a 10 000-key histogram was walked with all
            // eight classes until the last of them had underflowed -- 27 tiles at 400 instructions -- though two are
            // live for the first dozen and none for the rest.
            double tile_sum = 0.0;
            auto sum_full = [&](auto m_tag) __attribute__((always_inline)) {
                constexpr int M = decltype(m_tag)::value;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    double sc[16];
#pragma unroll
                    for (int b = 0; b < 16; ++b)
                        sc[b] = scal[16 * half + b];
#pragma unroll
                    for (int b = 0; b < 16; b += 2) {
                        double g1, g2;
                        st.template step2n<M>(xx, g1, g2);
                        tile_sum = fma(g1, sc[b], tile_sum);
                        tile_sum = fma(g2, sc[b + 1], tile_sum);
                    }
                }
            };
            if (nb == kTileBins) {
                if (N > 2 && n_live > 2)
                    sum_full(std::integral_constant<int, N>{});
                else if (N > 2 && n_live == 2)
                    sum_full(std::integral_constant<int, (N > 2 ? 2 : 1)>{});
                else if (n_live >= 1)
                    sum_full(std::integral_constant<int, 1>{});
                // (nothing is on: the tile adds nothing)
            } else {
                for (int b = 0; b < nb; ++b)
                    tile_sum = fma(st.template step_n<N>(), scal[b], tile_sum);
            }
            acc_sp.add(tile_sum);
            tile_sp = 0.0;
        } else if (nb == kTileBins) {
            tile_sp = 0.0;
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
                    st.template step2n<N>(xx, g1, g2);
                    // (measured and not kept, round 5: the two logs of a key pair through fast_log_bits_n<2>, their table
                    // reads in flight together -- 6 spilled registers at 128, C2 0.176 against 0.166 ms)
                    account(g1 * sc[b], hc[b], t * kTileBins + 16 * half + b);
                    account(g2 * sc[b + 1], hc[b + 1], t * kTileBins + 16 * half + b + 1);
                }
            }
        } else {
            tile_sp = 0.0;
            for (int b = 0; b < nb; ++b)
                account(st.template step_n<N>() * scal[b], cnt[b], t * kTileBins + b);
        }
        if (TAIL)
            acc_sp.add(tile_sp);
        st.template leave_tile_n<N>(rc.renorm);
    };
    // The rates of the error classes fall geometrically (covest/models.py:74-79): along the keys the streams go out
    // from the top, and once every class but the error-free one has gone for good in all lanes of the wave
    // (streams.h `gone`: off, outside the window, past the mode) the rest of the tiles -- most of them, for a
    // histogram whose counted keys lie in the hundreds and thousands -- is walked with that one stream, in a loop of
    // its own so that the other streams' registers are free there.  Their terms are exact zeros: no bit changes.
    int t = 0;
    for (; t < tv.n_tiles; ++t) {
        if (tv.run_start[t] != 0 && t > 0) { // (wave-uniform) a gap in the keys: who is left on the far side of it?
            const TileRec rc = tv.rec[t];
            st.retire_at_run_start(rc.k0 - 1.0, rc.k0 + (double)(rc.nb - 1), rc.lgam_prev, rc.lgam_last);
        }
        if (st.only_first_left())
            break;
        do_tile(std::integral_constant<int, S>{}, t, false, false);
    }
    // CLOSED FORM for the rest (round 3).  With one stream left, p_j = a_0 TP(x_0, j) for every later key
    // -- the other classes' terms are below e^-760, exact zeros in the reference's doubles too -- so
    //     log p_j = c_0 + j ln x_0 - ln j!        (c_0 = ln a_0 - D(x_0): the stream's anchor constants, streams.h)
    // and the rest of the sum over the counted keys is three multiply-adds against sums the host made once per
    // histogram (tiles.h suf_*):  c_0 sum h_j + ln x_0 sum j h_j - sum h_j ln j!.  It stands where every p_j is an
    // ordinary double: log p_j is concave in j, so its smallest value over the remaining counted keys is at the first
    // or the last of them, and that is compared with the clamp of direct_point.h.  Below e^-746 at either end p_j is 0
    // in the reference (its cast to double flushes below 2^-1075 = e^-745.13, c_src/covest_poissonmodule.c:32, and
    // a_0 <= 1) and the sum -inf (utils.safe_log).  A lane in between -- a p_j near or in the subnormal range -- sends
    // its WAVE through the key-by-key walk, which names the rows for the strict evaluation; in a grid that is a band
    // a few points wide.  C2 spends 330 of its 367 keys here.
    // Round 5: WITH A TAIL the same.  What a tail adds is sp_j = sum of p_j over EVERY key (covest/models.py:103), and
    // that needs the walk -- but a walk that only sums: two instructions a key for the one stream that is left, against
    // twenty with the log.  So the decision below is taken as without a tail, the closed form supplies the logs' sum
    // wherever it stands, and the tiles it covers are walked for their sum alone (do_tile with ll_done).  Until then a
    // histogram with a tail -- what every real CovEst run hands the model, covest/histogram.py:105-134 -- took a log per
    // key and point: C2 on all 10 000 keys 1.64 ms against 0.17 without the tail.
    bool walk_rest = t < tv.n_tiles;
    bool sums_only = false; // (TAIL, wave-uniform) the logs of all the remaining tiles are accounted for
#ifdef COVEST_DIAG
    // the point's route, for tools/dump_c2_classes.py: 0 the closed form was never asked (every tile walked with all
    // streams, or no counted key left), 1 closed form taken, 2 -inf by its bound, 3 a lane that would have taken
    // 1 or 2 but whose WAVE walks because of another lane, 4 a lane that sends its wave through the walk
    double diag_class = 0.0, diag_lp = NAN;
#endif
    if (walk_rest) {
        const double first = tv.suf_first[t];
        if (first == 0.0) {
            sums_only = true; // (wave-uniform) no counted key is left
        } else {
            const double lx0 = st.an.lx(0), c0 = st.an.c(0);
            const double lp_first = fma(first, lx0, c0 - tv.suf_first_lg[t]);
            const double lp_last = fma(tv.last_key[0], lx0, c0 - tv.last_key[1]);
            const double lp_min = fmin(lp_first, lp_last);
            const bool fine = lp_min > sub_list.log_p_clamp + 0.5;
            const bool none = lp_first < -746.5 || lp_last < -746.5; // (a stream that is off has c_0 = -inf)
#ifdef COVEST_DIAG
            diag_lp = lp_min;
            const bool diag_walks = __any(finite && !fine && !none);
            diag_class = (finite && !fine && !none) ? 4.0 : diag_walks ? 3.0 : fine ? 1.0 : 2.0;
#endif
            if (!__any(finite && !fine && !none)) { // wave-uniform
                sums_only = true;
                dead |= __ballot(none && !fine);
                if (fine)
                    acc_ll += fma(c0, tv.suf_h[t], fma(lx0, tv.suf_jh[t], -tv.suf_lgh[t]));
            }
        }
    }
    // A wave that must walk (round 4: walks only WHERE it must).  log p_j is concave in j, so over the keys of ONE tile
    // its minimum lies at an end of the tile too: a tile at whose ends every lane is well above the clamp holds no row for
    // the strict evaluation and no zero, and its share of the sum is the closed form again -- three multiply-adds
    // against the tile's own sums (differences of the suffix sums).  Only the tiles some lane comes near the clamp in
    // are walked key by key -- for C2 the last two or three of the forty the walk used to take, which was a sixth of
    // the kernel's instructions; without a tail the other tiles are skipped and the stream is anchored afresh behind
    // them, as at the start of a run; with one they are walked for their sums.  (ONE call of do_tile for all of this:
    // it is inlined, a thousand instructions a copy.)
    if (walk_rest && (TAIL || !sums_only)) {
        const double lx0 = st.an.lx(0), c0 = st.an.c(0);
        bool skipped = false;
        for (; t < tv.n_tiles; ++t) {
            // (the one stream has GONE -- off in every lane of the wave, outside the window, past its mode, streams.h -- and
            // the rest only enters sp_j: every later p_j is an exact zero.  The far end of a long histogram: with c around
            // 4000 the keys beyond 4500 of C2's 10 000 add nothing, and walking them was most of a tail-carrying launch)
            if (sums_only && (st.gone & 1u))
                break;
            bool ll_done = sums_only;
            if (sums_only) {
                // ... and long before it has gone it has stopped MATTERING to a sum: sp_j is a sum of p_j <= 2.5 whose
                // error is felt in absolute terms (tail * log(1 - sp_j)), so a key whose p_j is below e^-60 = 9e-27 adds
                // nothing a double of sp_j can hold -- ten thousand of them 1e-22.  log p_j is concave in j: over a tile
                // that does not hold the mode x_0 its largest value is at an end.  A tile that is negligible for every
                // lane is skipped (the stream anchored afresh behind it), and once every lane is past its mode the walk
                // ends: the stream is walked 11 standard deviations either side of its mode instead of 39.
                const TileRec rc = tv.rec[t];
                const double k0 = rc.k0, x0 = st.x[0];
                const double lp_lo = fma(k0 - 1.0, lx0, c0 - rc.lgam_prev);
                // (inside a stretch that is being skipped: EIGHT tiles at a glance first -- the same test at the two ends
                // of the eight, one record more -- so that a long dead stretch costs a test per 256 keys)
                if (skipped && t + 7 < tv.n_tiles) {
                    const TileRec r8 = tv.rec[t + 7];
                    const double klast8 = r8.k0 + (double)(r8.nb - 1);
                    const double lp_hi8 = fma(klast8, lx0, c0 - r8.lgam_last);
                    const bool matters8 = !(fmax(lp_lo, lp_hi8) < -60.0) || (x0 >= k0 - 2.0 && x0 <= klast8 + 1.0);
                    if (!__any(finite && matters8)) { // wave-uniform
                        if (!__any(finite && !(x0 < k0 - 2.0)))
                            break;
                        t += 7;
                        continue;
                    }
                }
                const double klast = k0 + (double)(rc.nb - 1);
                const double lp_hi = fma(klast, lx0, c0 - rc.lgam_last);
                const bool holds_mode = x0 >= k0 - 2.0 && x0 <= klast + 1.0;
                const bool matters = !(fmax(lp_lo, lp_hi) < -60.0) || holds_mode; // (a NaN: matters)
                if (!__any(finite && matters)) { // wave-uniform
                    if (!__any(finite && !(x0 < k0 - 2.0)))
                        break; // every lane is past its mode: nothing later matters either
                    skipped = true;
                    continue;
                }
            }
            if (!sums_only) {
                const TileRec rc = tv.rec[t];
                const double k0 = rc.k0;
                const double lp_lo = fma(k0 - 1.0, lx0, c0 - rc.lgam_prev); // (at the key before the tile: a superset)
                const double lp_hi = fma(k0 + (double)(rc.nb - 1), lx0, c0 - rc.lgam_last);
                const bool near = !(fmin(lp_lo, lp_hi) > sub_list.log_p_clamp + 0.5); // (a stream that is off, a NaN: near)
                if (!__any(finite && near)) { // wave-uniform
                    acc_ll += fma(c0, tv.suf_h[t] - tv.suf_h[t + 1],
                                  fma(lx0, tv.suf_jh[t] - tv.suf_jh[t + 1], -(tv.suf_lgh[t] - tv.suf_lgh[t + 1])));
                    ll_done = true;
                    if (!TAIL) {
                        skipped = true;
                        continue;
                    }
                }
            }
            do_tile(std::integral_constant<int, 1>{}, t, skipped, ll_done);
            skipped = false;
        }
    }

    double tail_term = 0.0;
    if (TAIL) {
        double sp = (acc_sp.hi + acc_sp.lo) * (1.0 / kBasicScale); // (the p_j were summed times 2^64: exact)
        if (!(sp < 1.0))
            sp = 1.0;
        if (sp < 1.0)
            tail_term = m.tail * log(1.0 - sp);
    }
    if ((dead >> (threadIdx.x & (kWave - 1))) & 1)
        acc_ll = isnan(acc_ll) ? acc_ll : -INFINITY; // h * -inf summed with finite terms
    double ll = acc_ll + tail_term;
    if (!finite) {
        ll = NAN; // a NaN parameter poisons every p_j in the reference
        subw = 0;
    }
    if (!isfinite(ll))
        subw = 0; // -inf (or NaN) whatever the handed-back keys are worth
#ifdef COVEST_DIAG
    if (sub_list.diag_class != 0) {
        ll = sub_list.diag_class == 1 ? diag_class : diag_lp;
        subw = 0;
    }
#endif
    if (live) {
        out_ll[pt] = ll;
        if (subw != 0) // (rare) queue the point for ll_fix_list_kernel
            sub_list.push(pt + sub_list.index_offset, subw);
    }
}

// The case the reference's `main` builds (max_error = 8): 4 waves per SIMD at 128 registers.
// TWO waves a workgroup (round 5; four until then).  A workgroup's LDS -- the lanes' anchors and the log table, 36 KB for
// four waves: four workgroups to a CU -- stays allocated until its LAST wave has ended, and the waves of a workgroup end
// at very different times: 44 % of C2's are doomed and leave at once, the others walk as many tiles as their error rates
// keep classes alive.  By the counters the launch held 1.5 waves a SIMD on average (SQ_WAVE_CYCLES over the launch's
// SIMD-cycles) where the registers allow four: the slots of the waves that had left could not be refilled.  With 128
// threads (20 KB: eight workgroups to a CU, the same sixteen waves) a slot comes free when two waves are done instead of
// four: C2 0.159 -> 0.145 ms, the trimmed C2 with its tail 0.197 -> 0.169; with 64 threads (every wave its own
// workgroup, thirteen to a CU by the LDS) 0.150 and 0.166 -- profiles/r05_c2_ab_threads_per_workgroup.txt.
#ifndef COVEST_BASIC_BD
#define COVEST_BASIC_BD 128
#endif
template <bool TAIL>
__global__ __launch_bounds__(COVEST_BASIC_BD) __attribute__((amdgpu_waves_per_eu(TAIL ? 3 : 4, TAIL ? 3 : 4))) void ll_basic_kernel(
    const DevModel m, const int32_t n_tiles, const int32_t n_items, const double *__restrict__ tile_dbl,
    const int32_t *__restrict__ tile_int, const PointSource src, const int64_t n, double *__restrict__ out_ll,
    SubList sub_list)
{
    ll_basic_body<8, TAIL, COVEST_BASIC_BD>(m, n_tiles, n_items, tile_dbl, tile_int, src, n, out_ll, sub_list);
}

// More error classes (max_error = k + 1 = 22 when a model is built directly, covest/models.py:28-31): the same walk
// with 16, 24 or 32 streams per lane -- more registers, one wave per workgroup.
template <int S, bool TAIL>
__global__ __launch_bounds__(64) void ll_basic_wide_kernel(const DevModel m, const int32_t n_tiles, const int32_t n_items,
                                                           const double *__restrict__ tile_dbl,
                                                           const int32_t *__restrict__ tile_int, const PointSource src,
                                                           const int64_t n, double *__restrict__ out_ll, SubList sub_list)
{
    ll_basic_body<S, TAIL, 64>(m, n_tiles, n_items, tile_dbl, tile_int, src, n, out_ll, sub_list);
}

} // namespace

// One (S, TAIL) variant per translation unit (COVEST_BASIC_VARIANT, as ll_factored.hip: the HIP runtime loads a
// translation unit's code object on the first launch of one of its kernels; all eight together are 1.5 MB of code).
template <int S, bool TAIL>
void launch_ll_basic_variant(dim3 grid, hipStream_t stream, const DevModel &m, const TileView &tv, const PointSource &part,
                             int64_t cnt, double *out, const SubList &sl)
{
    if (S == 8) {
        const size_t lds8 = (size_t)2 * 8 * COVEST_BASIC_BD * sizeof(double);
        hipLaunchKernelGGL((ll_basic_kernel<TAIL>), grid, dim3(COVEST_BASIC_BD), lds8, stream, m, tv.n_tiles, tv.n_items, tv.dbl_base,
                           tv.int_base, part, cnt, out, sl);
    } else {
        const size_t lds = (size_t)2 * S * 64 * sizeof(double);
        hipLaunchKernelGGL((ll_basic_wide_kernel<S, TAIL>), grid, dim3(64), lds, stream, m, tv.n_tiles, tv.n_items,
                           tv.dbl_base, tv.int_base, part, cnt, out, sl);
    }
}

#define COVEST_BASIC_ARGS dim3, hipStream_t, const DevModel &, const TileView &, const PointSource &, int64_t, double *, const SubList &
#ifdef COVEST_BASIC_VARIANT
// this translation unit holds ONE variant: bits 2..1 = 0: 8 streams, 1: 16, 2: 24, 3: 32; bit 0 = TAIL
template void launch_ll_basic_variant<8 * ((COVEST_BASIC_VARIANT >> 1) + 1), (COVEST_BASIC_VARIANT & 1) != 0>(COVEST_BASIC_ARGS);
#else
extern template void launch_ll_basic_variant<8, false>(COVEST_BASIC_ARGS);
extern template void launch_ll_basic_variant<8, true>(COVEST_BASIC_ARGS);
extern template void launch_ll_basic_variant<16, false>(COVEST_BASIC_ARGS);
extern template void launch_ll_basic_variant<16, true>(COVEST_BASIC_ARGS);
extern template void launch_ll_basic_variant<24, false>(COVEST_BASIC_ARGS);
extern template void launch_ll_basic_variant<24, true>(COVEST_BASIC_ARGS);
extern template void launch_ll_basic_variant<32, false>(COVEST_BASIC_ARGS);
extern template void launch_ll_basic_variant<32, true>(COVEST_BASIC_ARGS);

namespace {
template <int S>
void launch_s(bool tail, dim3 grid, hipStream_t stream, const DevModel &m, const TileView &tv, const PointSource &part,
              int64_t cnt, double *out, const SubList &sl)
{
    if (tail)
        launch_ll_basic_variant<S, true>(grid, stream, m, tv, part, cnt, out, sl);
    else
        launch_ll_basic_variant<S, false>(grid, stream, m, tv, part, cnt, out, sl);
}
} // namespace

hipError_t launch_ll_basic(const DevModel &m, const TileView &tv, const PointSource &src, int64_t n,
                           double *out_ll, const SubList &sub_list, hipStream_t stream)
{
    if (n <= 0)
        return hipSuccess;
    if (m.n_err > 32 || m.kind != 0)
        return hipErrorInvalidValue;
    const int s_pad = ((m.n_err + 7) / 8) * 8;
    const int bd = s_pad == 8 ? COVEST_BASIC_BD : 64;
    // HIP wraps a grid of more than 2^32 threads silently: at most 2^23 workgroups per launch
    const int64_t per_launch = (int64_t)bd << 23;
    for (int64_t first = 0; first < n; first += per_launch) {
        const int64_t cnt = n - first < per_launch ? n - first : per_launch;
        const dim3 grid((unsigned)((cnt + bd - 1) / bd));
        PointSource part = src;
        SubList sl = sub_list;
        sl.index_offset = first;
        if (src.is_grid)
            part.flat_begin = src.flat_begin + first;
        else
            part.params = src.params + first * 2;
        const bool tail = m.tail != 0.0;
        if (s_pad == 16)
            launch_s<16>(tail, grid, stream, m, tv, part, cnt, out_ll + first, sl);
        else if (s_pad == 24)
            launch_s<24>(tail, grid, stream, m, tv, part, cnt, out_ll + first, sl);
        else if (s_pad == 32)
            launch_s<32>(tail, grid, stream, m, tv, part, cnt, out_ll + first, sl);
        else
            launch_s<8>(tail, grid, stream, m, tv, part, cnt, out_ll + first, sl);
    }
    return hipGetLastError();
}
#endif // COVEST_BASIC_VARIANT

} // namespace covest
