// plan_factored.cpp -- K-factored's work descriptions (tiles.h FactoredPlan): the parts of a dense repeats-model grid
// (units dealt to waves longest-first, pieces, shared steps) and the point lists of list mode.
#include "host.h"

using namespace covest;

namespace covest {

// RepeatsModel.get_b_o, covest/models.py:193-208, with libm pow as CPython's float ** int.
double copy_number_weight_host(double q1, double q2, double q, int o)
{
    if (o == 1)
        return q1;
    if (o == 2)
        return (1 - q1) * q2;
    return (1 - q1) * (1 - q2) * q * std::pow(1 - q, (double)(o - 3));
}

// ---- K-factored plans of a dense repeats grid (tiles.h FactoredPlan) ----
// The Q = |q1| x |q2| x |q| weight vectors are sorted by threshold_o (descending) into slots, 16 per q-tile.  A PART
// is one launch's work description: a range of q-tiles and a CHUNK of copy numbers o_base + 1 .. o_base + chunk.
//   * weight vectors whose threshold_o - 1 fits the lanes of a workgroup (`chunk` copy numbers) form ONE part that
//     writes log-likelihoods (list_mode 0);
//   * the longer ones (optimize_grid walks q down to 0.01: threshold_o ~ 1500) are the first q-tiles of the sorted
//     order; they get one part per chunk (list_mode 3) that sums p_j into HBM, and ll_finish_dense takes the logs.
//     Only those tiles pay for it: the rest of the grid stays on the one-launch path.
// With more than 8 error classes (max_error = k + 1 = 22 when a model is built directly, covest/models.py:28-31) a
// copy number's classes are dealt to n_pass = ceil(S / 8) lanes, which the contraction treats as extra columns with
namespace {
// the same weight; a chunk then holds 512 / n_pass copy numbers.
struct QOrder {
    int64_t n1, n2, n3, nq;
    // slot -> index in the (q1, q2, q) product, -1 = padding; [n_qtiles * 16].  Either all weight vectors by
    // descending threshold_o, or (shared steps, tiles.h) tile by tile: the 16 slots of a tile share q, descending
    // threshold_o inside, the tiles by descending largest threshold_o.
    std::vector<int32_t> order;
    std::vector<int32_t> tile_nsh; // [n_qtiles] shared steps of the tile's units (0: none)
    int32_t n_qtiles;
    int t_max;
};

// One part: q-tiles [tile_lo, tile_hi) of the sorted order, copy numbers o_base + 1 .. o_base + chunk.
int build_plan_part(covest_grid *g, const double *const *axes, const std::vector<int32_t> &t_table, const QOrder &qo,
                    int32_t tile_lo, int32_t tile_hi, int o_base, int chunk, int n_pass, int list_mode, DevBuf &buf,
                    FactoredPlan &pl, std::vector<int32_t> *q_orig_out)
{
    covest_model *m = g->model;
    const int64_t n2 = qo.n2, n3 = qo.n3, nq = qo.nq;
    const int32_t n_qtiles = tile_hi - tile_lo;
    const size_t n_slots = (size_t)n_qtiles * 16;
    std::vector<int32_t> nsteps((size_t)n_qtiles, 0), q_t(n_slots, 0), q_orig(n_slots, -1);
    std::vector<double> r4(n_slots, 0.0);
    int t_loc_max = 1; // largest LOCAL threshold: copy numbers of the chunk are o_base + 1 .. o_base + t_local - 1
    auto weights_of = [&](size_t slot_global, double &q1, double &q2, double &q) {
        const int64_t qi = qo.order[slot_global];
        const int64_t a = qi / (n2 * n3), b = (qi / n3) % n2, c = qi % n3;
        q1 = clamp_one(m->dm, 2, axes[2][a]);
        q2 = clamp_one(m->dm, 3, axes[3][b]);
        q = clamp_one(m->dm, 4, axes[4][c]);
    };
    // shared steps (tiles.h): only the plain dense shape has them
    std::vector<int32_t> nsh((size_t)n_qtiles, 0);
    if (list_mode == 0 && n_pass == 1 && o_base == 0)
        for (int32_t qt = 0; qt < n_qtiles; ++qt)
            nsh[(size_t)qt] = qo.tile_nsh[(size_t)(tile_lo + qt)];
    double r4_q = NAN, r4_of_q = 0.0; // (neighbouring slots of a shared tile have one q: one pow for them)
    for (size_t ls = 0; ls < n_slots; ++ls) {
        const size_t gs = (size_t)tile_lo * 16 + ls;
        if (qo.order[gs] < 0)
            continue; // padding column
        const int64_t qi = qo.order[gs];
        const int t_loc = std::min(chunk + 1, std::max(0, (int)t_table[(size_t)qi] - o_base));
        q_t[ls] = t_loc;
        q_orig[ls] = (int32_t)qi;
        double q1, q2, q;
        weights_of(gs, q1, q2, q);
        if (!(q == r4_q)) {
            r4_q = q;
            r4_of_q = std::pow(1 - q, 4.0);
        }
        r4[ls] = r4_of_q;
        const int steps = t_loc > 1 ? (t_loc - 1 + 3) / 4 : 0;
        nsteps[ls / 16] = std::max(nsteps[ls / 16], steps);
        t_loc_max = std::max(t_loc_max, t_loc);
    }
    if (q_orig_out)
        *q_orig_out = q_orig;
    const int max_o = t_loc_max - 1;
    const int pass_stride = ((max_o + 3) / 4) * 4; // a pass begins on an MFMA step
    const int n_columns = n_pass == 1 ? max_o : n_pass * pass_stride;
    // ---- deal (q-tile, half) units to the waves of a workgroup (tiles.h) ----
    const int ld = ((n_columns + 31) / 32) * 32 + 2;
    const int n_buf = (2 * (size_t)kTileBins * ld + 64) * sizeof(double) + 13440 <= 160 * 1024 ? 2 : 1; // (+ the kernel's static LDS: log table, hand-back records, row constants)
    const int n_units = 2 * n_qtiles;
    const int hu = kHalfUnits; // (768 threads with 2 slots per half, 3 waves/SIMD, was measured: +1 %)
    const int mu = 2 * hu;
    // a unit needs at least one piece per pass: fewer units fit a wave's slots
    const int units_per_wave = std::max(1, mu / n_pass);
    const int nt = (n_columns <= 256 && n_units <= 4 * units_per_wave) ? 256 : 512;
    const int nw = nt / 64;
    const int cap_block = nw * units_per_wave;
    // workgroups per (c, e): as many as the units need -- and, for a grid with few (c, e) pairs (optimize_grid's
    // have 36), enough to put the chip's 256 CUs to work: each rebuilds G, but they share the contraction and the logs
    // (decided by the WHOLE grid's (c, e) count, not the block's: a point's value may not depend on how the grid was
    // cut into blocks, and the assignment of units to waves fixes the order of its sums)
    const int64_t n_ce_grid = std::max<int64_t>(1, g->len[0] * g->len[1]);
    // (as many as FIT the chip in one round: 36 (c, e) pairs x 8 would be 288 workgroups for 256 CUs, a second round for the
    // last 32 -- round 5: 7, by the trace of an optimize_grid search whose K-factored launches took 25 us for 15 keys)
    const int want_blocks = n_ce_grid >= 192 ? 1 : (int)std::min<int64_t>(n_qtiles, std::max<int64_t>(1, 256 / n_ce_grid));
    const int n_qblocks = std::max(std::max(1, (n_units + cap_block - 1) / cap_block), want_blocks);
    // cost model of the assignment, in MFMA steps: a unit costs its steps (in every pass) plus its share of the
    // logs; a builder wave starts with the cost of phase A (tuned on C3 with the in-kernel stamps)
    // (the assignment fixes the order of a point's sums: the shipped library takes the constants of tiles.h, only a
    // diagnostic build -- tiles.h -- or a TUNING build of this file alone, -DCOVEST_TUNE linked against the shipped
    // kernels (tools/build_tune.sh, profiles/r04_c3_factored_lpt_sweep_*.txt), lets the environment override them)
    int unit_overhead = kUnitOverhead, build_cost = kBuildCost, shared_div = kSharedStepsPerMfma;
    int last_builder_extra = kLastBuilderExtra;
    // (with a tail an item may stand for up to 32 count-less tiles, tiles.h: the builders walk every one of them
    // while the contraction sees one item -- charge them for the tiles an item holds on average)
    if (m->has_tiles && m->tv.n_items > 0)
        build_cost = (int)std::lround((double)(m->low_tile_share >= 0.75 ? kBuildCostLowKeys : kBuildCost) *
                                      (double)m->tv.n_tiles / (double)m->tv.n_items);
#if defined(COVEST_DIAG) || defined(COVEST_TUNE)
    if (const char *v = std::getenv("COVEST_FACTORED_UNIT_OVERHEAD"))
        unit_overhead = std::atoi(v);
    if (const char *v = std::getenv("COVEST_FACTORED_BUILD_COST"))
        build_cost = std::atoi(v);
    if (const char *v = std::getenv("COVEST_FACTORED_SHARED_DIV"))
        shared_div = std::max(1, std::atoi(v));
    if (const char *v = std::getenv("COVEST_FACTORED_LAST_BUILDER_EXTRA"))
        last_builder_extra = std::atoi(v);
#endif
    const size_t n_unit = (size_t)n_qblocks * nw * mu;
    std::vector<int32_t> unit_tile(n_unit, -1), unit_half(n_unit, 0), unit_s0(n_unit, 0), unit_o0(n_unit, 1),
        unit_len(n_unit, 0), unit_cont(n_unit, 0), unit_nsh(n_unit, 0);
    // MFMA steps of a tile's units: all of them, or step 0 and those after the shared ones
    auto mfma_steps = [&](int qt) { return (int)nsteps[(size_t)qt] - (int)nsh[(size_t)qt]; };
    for (int blk = 0; blk < n_qblocks; ++blk) {
        struct Unit {
            int tile, half, cost, pieces; // pieces: per pass
        };
        std::vector<Unit> units;
        for (int qt = blk; qt < n_qtiles; qt += n_qblocks) // tiles are sorted by T: interleave over blocks
            for (int h = 0; h < 2; ++h)
                units.push_back({qt, h, n_pass * std::max(1, mfma_steps(qt)) + unit_overhead +
                                            (nsh[(size_t)qt] ? 1 + (nsh[(size_t)qt] + shared_div - 1) / shared_div : 0), 1});
        std::stable_sort(units.begin(), units.end(), [](const Unit &a, const Unit &b) { return a.cost > b.cost; });
        // longest first into the lightest SIMD (waves w and w + 4 share one) that still has room,
        // then into the lighter of that SIMD's waves with room
        const int n_bins = std::min(4, nw);
        std::vector<long> bin_load((size_t)n_bins, 0), wave_load((size_t)nw, 0);
        if (n_buf == 2) // builders contract less: they fill the next key tile in the same interval
            for (int w = 0; w < nw && w * 64 < n_columns; ++w) {
                // (the builder of the TOP copy numbers is the wave every interval waits for -- the stamps of round 3:
                // its streams stay live over the widest range of keys, and it shares its SIMD with another builder)
                const int cost = build_cost + (((w + 1) * 64 >= n_columns && w >= n_bins) ? last_builder_extra : 0);
                bin_load[(size_t)(w % n_bins)] += cost;
                wave_load[(size_t)w] += cost;
            }
        std::vector<std::vector<Unit>> held((size_t)nw);
        for (const Unit &u : units) {
            int best_wave = -1;
            for (int w = 0; w < nw; ++w) {
                if ((int)held[(size_t)w].size() >= units_per_wave)
                    continue;
                if (best_wave < 0) {
                    best_wave = w;
                    continue;
                }
                const long lb = bin_load[(size_t)(w % n_bins)], bb = bin_load[(size_t)(best_wave % n_bins)];
                if (lb < bb || (lb == bb && wave_load[(size_t)w] < wave_load[(size_t)best_wave]))
                    best_wave = w;
            }
            if (best_wave < 0)
                return fail(COVEST_E_INVALID, "K-factored plan: no wave has room for a unit (internal)");
            held[(size_t)best_wave].push_back(u);
            bin_load[(size_t)(best_wave % n_bins)] += u.cost;
            wave_load[(size_t)best_wave] += u.cost;
        }
        for (int w = 0; w < nw; ++w) {
            std::vector<Unit> &mine = held[(size_t)w];
            // cut the unit with the longest pieces once more (in every pass) while slots are free (tiles.h)
            auto piece_len = [&](const Unit &u) { return (mfma_steps(u.tile) + u.pieces - 1) / u.pieces; };
            int used = (int)mine.size() * n_pass;
            while (used + n_pass <= mu && !mine.empty()) {
                size_t longest = 0;
                for (size_t i = 1; i < mine.size(); ++i)
                    if (piece_len(mine[i]) > piece_len(mine[longest]))
                        longest = i;
                if (nsh[(size_t)mine[longest].tile])
                    break; // (a unit with shared steps is short already, and stays in one piece)
                Unit trial = mine[longest];
                trial.pieces += 1;
                if (piece_len(trial) < kMinPieceSteps)
                    break;
                mine[longest].pieces += 1;
                used += n_pass;
            }
            // slots sorted by piece length (descending), the pieces of a unit adjacent
            std::stable_sort(mine.begin(), mine.end(),
                             [&](const Unit &a, const Unit &b) { return piece_len(a) > piece_len(b); });
            size_t k = 0;
            for (const Unit &u : mine) {
                bool first = true;
                for (int pass = 0; pass < n_pass; ++pass)
                    for (int p = 0; p < u.pieces; ++p, ++k) {
                        const size_t at = ((size_t)blk * nw + w) * mu + k;
                        unit_tile[at] = u.tile;
                        unit_half[at] = u.half;
                        unit_s0[at] = pass * (pass_stride / 4) + p * piece_len(u);
                        unit_o0[at] = 1 + 4 * p * piece_len(u);
                        unit_len[at] = piece_len(u); // equal lengths: steps past the unit's end are cut off by T
                        unit_cont[at] = first ? 0 : 1;
                        unit_nsh[at] = nsh[(size_t)u.tile];
                        first = false;
                    }
            }
        }
    }
    // weights of every slot's first two MFMA steps, per lane (lane = 16 * (o mod 4) + column)
    std::vector<double> piece_w(n_unit * 64 * 2, 0.0), unit_rho(n_unit * 4, 1.0);
    // (eight consecutive copy numbers per slot and column: one libm pow, the rest by multiplication -- the kernel
    // advances the weights the same way from the third step on; a grid with few (c, e) pairs has many slots)
    for (size_t at = 0; at < n_unit; ++at) {
        const int qt = unit_tile[at];
        if (qt < 0)
            continue;
        // (the columns of a shared tile have ONE q: its powers are made once per slot, not once per column -- the same
        // calls of pow with the same arguments, so the same bits; they were most of a plan's build time, which is a third
        // of an optimize_grid iteration)
        double pw_q = NAN, pw_geo = 0.0, pw_16 = 0.0, pw_4 = 0.0, pw_inv = 0.0, pw_after = 0.0;
        for (int colx = 0; colx < 16; ++colx) {
            const size_t gs = ((size_t)tile_lo + (size_t)qt) * 16 + (size_t)colx;
            if (qo.order[gs] < 0)
                continue; // padding column
            double q1, q2, q;
            weights_of(gs, q1, q2, q);
            const int o_first = o_base + unit_o0[at];
            const double head = (1 - q1) * (1 - q2) * q, base = 1 - q;
            if (!(q == pw_q)) { // (NaN: never equal, made afresh)
                pw_q = q;
                pw_geo = o_first >= 3 ? std::pow(base, (double)(o_first - 3)) : 1.0;
                if (unit_nsh[at] > 0) {
                    pw_16 = std::pow(base, 16.0);
                    pw_4 = std::pow(base, 4.0);
                    pw_inv = 1.0 / std::pow(base, 4.0 * (double)unit_nsh[at]);
                    pw_after = std::pow(base, (double)(o_first + 4 * (unit_nsh[at] + 1) - 3));
                }
            }
            double geo = pw_geo; // base^(o - 3) at o = max(o_first, 3)
            for (int d = 0; d < 8; ++d) {
                const int o = o_first + d;
                double w;
                if (o < 3) {
                    w = copy_number_weight_host(q1, q2, q, o);
                } else {
                    w = head * geo;
                    geo *= base;
                }
                // (the piece's first step needs no mask in the kernel: a copy number at or beyond the column's
                // cut-off gets weight 0 here -- covest/models.py:239; later steps are cut off by the step count)
                if (d < 4 && unit_o0[at] + d >= (int)q_t[(size_t)qt * 16 + (size_t)colx])
                    w = 0.0;
                // layout [slot][lane][2]: {first step, the step the kernel's running weight starts from} -- the second
                // step of the piece, or (units with shared steps) the first step after them, written below
                if (d < 4 || unit_nsh[at] == 0)
                    piece_w[(at * 64 + (size_t)((d & 3) * 16 + colx)) * 2 + (size_t)(d >> 2)] = w;
            }
            if (unit_nsh[at] > 0) {
                // (one q per tile: every live column writes the same values) -- the shared steps are summed with
                // weights RELATIVE TO THE FIRST of them, (1-q)^(4 (i - 1)) <= 1 (Horner in (1-q)^4, four chains in
                // (1-q)^16), and the MFMA that brings the sum in multiplies by b_o of that first step, which the kernel
                // makes from the weight it holds anyway -- b_o of the first step AFTER them -- times (1-q)^(-4 nsh)
                // (<= 1e10: the cut-off is where b_o reaches 1e-8)
                unit_rho[4 * at] = pw_16;
                unit_rho[4 * at + 2] = pw_4;
                unit_rho[4 * at + 3] = pw_inv;
            }
            if (unit_nsh[at] > 0) { // the first step after the shared ones: o = 5 + 4 nsh .. 8 + 4 nsh
                double g2 = pw_after; // base^(o_after - 3), o_after = o_first + 4 (nsh + 1)
                for (int d = 0; d < 4; ++d, g2 *= base)
                    piece_w[(at * 64 + (size_t)(d * 16 + colx)) * 2 + 1] = head * g2;
            }
        }
    }
    // the half-1 partner of every half-0 unit (the same workgroup holds both): looked up by the kernel's last step
    std::vector<int32_t> unit_pair(n_unit, -1);
    for (int blk = 0; blk < n_qblocks; ++blk) {
        const size_t b0 = (size_t)blk * nw * mu, b1 = b0 + (size_t)nw * mu;
        for (size_t at = b0; at < b1; ++at) {
            if (unit_tile[at] < 0 || unit_half[at] != 0 || unit_cont[at])
                continue;
            for (size_t at2 = b0; at2 < b1; ++at2)
                if (unit_tile[at2] == unit_tile[at] && unit_half[at2] == 1 && !unit_cont[at2]) {
                    unit_pair[at] = (int32_t)(at2 - b0);
                    break;
                }
        }
    }
    // one buffer: doubles first (r4 | piece_w), then int32 (q_T | q_orig | unit tables)
    const size_t n_dbl = n_slots + piece_w.size() + unit_rho.size();
    const size_t n_int = 2 * n_slots + 8 * n_unit;
    HIP_TRY(buf.reserve(n_dbl * sizeof(double) + n_int * sizeof(int32_t)));
    double *dbase = buf.as<double>();
    int32_t *ibase = reinterpret_cast<int32_t *>(dbase + n_dbl);
    {
        // staged on the host in the device layout, ONE copy (a dozen small copies cost ~150 us of the plan build);
        // blocking through the process's staging buffer, or asynchronous through the handle's own (covest_grid_reset)
        const size_t stage_bytes = n_dbl * sizeof(double) + n_int * sizeof(int32_t);
        StageSlot slot;
        const int src = grid_stage_begin(g, stage_bytes, slot);
        if (src != COVEST_OK)
            return src;
        double *sd = reinterpret_cast<double *>(slot.ptr);
        int32_t *si = reinterpret_cast<int32_t *>(sd + n_dbl);
        std::copy(r4.begin(), r4.end(), sd);
        std::copy(piece_w.begin(), piece_w.end(), sd + n_slots);
        std::copy(unit_rho.begin(), unit_rho.end(), sd + n_slots + piece_w.size());
        std::copy(q_t.begin(), q_t.end(), si);
        std::copy(q_orig.begin(), q_orig.end(), si + n_slots);
        int32_t *sp = si + 2 * n_slots;
        std::copy(unit_tile.begin(), unit_tile.end(), sp);
        std::copy(unit_half.begin(), unit_half.end(), sp + n_unit);
        std::copy(unit_s0.begin(), unit_s0.end(), sp + 2 * n_unit);
        std::copy(unit_o0.begin(), unit_o0.end(), sp + 3 * n_unit);
        std::copy(unit_len.begin(), unit_len.end(), sp + 4 * n_unit);
        std::copy(unit_cont.begin(), unit_cont.end(), sp + 5 * n_unit);
        std::copy(unit_nsh.begin(), unit_nsh.end(), sp + 6 * n_unit);
        std::copy(unit_pair.begin(), unit_pair.end(), sp + 7 * n_unit);
        const int crc = grid_stage_commit(g, slot, buf.ptr, stage_bytes);
        if (crc != COVEST_OK)
            return crc;
    }
    pl = FactoredPlan{};
    pl.c_axis = g->src.axis[0];
    pl.e_axis = g->src.axis[1];
    pl.n_e = g->len[1];
    pl.n_q = nq;
    pl.ce_begin = g->flat_begin / nq;
    pl.ce_end = (g->flat_end + nq - 1) / nq;
    pl.n_qtiles = n_qtiles;
    pl.max_o = max_o;
    pl.n_pass = n_pass;
    pl.pass_stride = pass_stride;
    pl.n_columns = n_columns;
    pl.o_base = o_base;
    pl.n_threads = nt;
    pl.half_units = hu;
    pl.n_qblocks = n_qblocks;
    pl.ld = ld;
    pl.n_buf = n_buf;
#ifdef COVEST_DIAG
    if (std::getenv("COVEST_FACTORED_NBUF"))
        pl.n_buf = std::atoi(std::getenv("COVEST_FACTORED_NBUF"));
#endif
    int32_t *ub = ibase + 2 * n_slots;
    pl.unit_tile = ub;
    pl.unit_half = ub + n_unit;
    pl.unit_s0 = ub + 2 * n_unit;
    pl.unit_o0 = ub + 3 * n_unit;
    pl.unit_len = ub + 4 * n_unit;
    pl.unit_cont = ub + 5 * n_unit;
    pl.unit_nsh = ub + 6 * n_unit;
    pl.unit_pair = ub + 7 * n_unit;
    pl.piece_w = dbase + n_slots;
    pl.unit_rho = dbase + n_slots + piece_w.size();
    pl.q_first8 = nullptr;
    pl.q_r4 = dbase;
    pl.qtile_nsteps = nullptr;
    pl.qtile_nfull = nullptr;
    pl.q_T = ibase;
    pl.q_orig = ibase + n_slots;
    pl.flat_begin = g->flat_begin;
    pl.flat_end = g->flat_end;
    pl.list_mode = list_mode;
    pl.p_clamp = clamp_for(m, qo.t_max);
    pl.n_seg = 1;
    pl.item_obase = nullptr;
    pl.partial = nullptr;
    pl.ce_first = pl.ce_begin;
    pl.n_cols_partial = (int64_t)n_slots;
    pl.skip_phases = 0;
    pl.diag = nullptr;
#ifdef COVEST_DIAG // diagnostic builds only (tiles.h): the shipped library has no knob that changes values
    {
        const char *skip = std::getenv("COVEST_FACTORED_SKIP");
        pl.skip_phases = skip ? std::atoi(skip) : 0;
        if (list_mode == 0 && std::getenv("COVEST_FACTORED_DIAG")) { // leaked on purpose
            void *dp = nullptr;
            const size_t bytes = (size_t)(pl.ce_end - pl.ce_begin) * n_qblocks * nw * 8 * sizeof(long long);
            if (hipMalloc(&dp, bytes) == hipSuccess && hipMemset(dp, 0, bytes) == hipSuccess) {
                pl.diag = static_cast<long long *>(dp);
                if (std::atoi(std::getenv("COVEST_FACTORED_DIAG")) >= 2)
                    pl.skip_phases |= 0x10000; // the barrier waits by eighths of the walk instead of the phase sums
                if (std::atoi(std::getenv("COVEST_FACTORED_DIAG")) == 3)
                    pl.skip_phases |= 0x20000; // ... of the first seven intervals one by one, the rest in the eighth
                if (std::atoi(std::getenv("COVEST_FACTORED_DIAG")) == 4)
                    pl.skip_phases = (pl.skip_phases & ~0x10000) | 0x40000; // the stages outside the walk
                std::fprintf(stderr, "COVEST_FACTORED_DIAG %p %zu\n", dp, bytes);
            }
        }
    }
#endif
    return COVEST_OK;
}

// All the parts of a dense repeats grid (see above).  g->has_plan stays false where K-factored does not apply:
// no tile table (keys beyond 16384 ...), more than 32 error classes, or more weight vectors than 2^24.
} // namespace

int build_factored_plan(covest_grid *g, const double *const *axes, const int64_t *axis_len,
                        const std::vector<int32_t> &t_table)
{
    covest_model *m = g->model;
    g->has_plan = false;
    if (!g->long_parts.empty()) {
        (void)hipDeviceSynchronize(); // (their buffers go back to the process's cache, host.h)
        DeviceIdleScope idle;
        for (covest_grid::Part &part : g->long_parts)
            part.buf.release();
        g->long_parts.clear();
    }
    g->n_long_tiles = 0;
    if (!m->has_tiles || m->n_par != 5 || m->dm.n_err > 32)
        return COVEST_OK;
    QOrder qo;
    qo.n1 = axis_len[2];
    qo.n2 = axis_len[3];
    qo.n3 = axis_len[4];
    qo.nq = qo.n1 * qo.n2 * qo.n3;
    if (qo.nq > (int64_t)1 << 24)
        return COVEST_OK;
    qo.t_max = 1;
    for (int64_t i = 0; i < qo.nq; ++i)
        qo.t_max = std::max(qo.t_max, (int)t_table[(size_t)i]);
    const int n_pass = (m->dm.n_err + 7) / 8;
    const int chunk = ((512 / n_pass) / 4) * 4; // copy numbers one workgroup's lanes hold
    g->t_max = qo.t_max;
    // ---- the order of the weight vectors: slots of 16 per q-tile ----
    // Shared steps (tiles.h) want the 16 columns of a tile to differ in q1 and q2 only: the n1 * n2 vectors of one q
    // are then laid out by descending threshold_o and padded to whole tiles.  Padding columns cost logs, shared steps
    // save MFMAs: taken when the padding stays below a third (n1 * n2 = 12, 16, 24, 27 .. 32, 36 ...), one lane per
    // copy number (max_error <= 8).  (Diagnostic builds: COVEST_FACTORED_SHARE=0 switches it off for A/B runs.)
    const int64_t group = qo.n1 * qo.n2, group_padded = (group + 15) / 16 * 16;
    bool share = n_pass == 1 && 3 * (group_padded - group) <= group;
#ifdef COVEST_DIAG
    if (const char *share_env = std::getenv("COVEST_FACTORED_SHARE"))
        share = share && std::atoi(share_env) != 0;
#endif
    int min_shared = kMinSharedSteps;
#if defined(COVEST_DIAG) || defined(COVEST_TUNE)
    if (const char *v = std::getenv("COVEST_FACTORED_MIN_SHARED"))
        min_shared = std::atoi(v);
#endif
    auto by_t = [&](int32_t a, int32_t b) { return t_table[(size_t)a] > t_table[(size_t)b]; };
    if (!share) {
        std::vector<int32_t> all((size_t)qo.nq);
        for (int64_t i = 0; i < qo.nq; ++i)
            all[(size_t)i] = (int32_t)i;
        std::stable_sort(all.begin(), all.end(), by_t);
        qo.n_qtiles = (int32_t)((qo.nq + 15) / 16);
        qo.order.assign((size_t)qo.n_qtiles * 16, -1);
        std::copy(all.begin(), all.end(), qo.order.begin());
        qo.tile_nsh.assign((size_t)qo.n_qtiles, 0);
    } else {
        struct Tile {
            int32_t slot[16];
            int t_hi, t_lo;
        };
        std::vector<Tile> tiles;
        std::vector<int32_t> one((size_t)group);
        for (int64_t c = 0; c < qo.n3; ++c) {
            for (int64_t ab = 0; ab < group; ++ab)
                one[(size_t)ab] = (int32_t)(ab * qo.n3 + c);
            std::stable_sort(one.begin(), one.end(), by_t);
            for (int64_t at = 0; at < group; at += 16) {
                Tile t;
                const int64_t live = std::min<int64_t>(16, group - at);
                for (int64_t i = 0; i < 16; ++i)
                    t.slot[i] = i < live ? one[(size_t)(at + i)] : -1;
                t.t_hi = (int)t_table[(size_t)t.slot[0]];
                t.t_lo = (int)t_table[(size_t)t.slot[live - 1]];
                tiles.push_back(t);
            }
        }
        std::stable_sort(tiles.begin(), tiles.end(), [](const Tile &a, const Tile &b) { return a.t_hi > b.t_hi; });
        qo.n_qtiles = (int32_t)tiles.size();
        qo.order.resize(tiles.size() * 16);
        qo.tile_nsh.assign(tiles.size(), 0);
        for (size_t t = 0; t < tiles.size(); ++t) {
            std::copy(tiles[t].slot, tiles[t].slot + 16, qo.order.begin() + (std::ptrdiff_t)t * 16);
            // steps 1 .. nsh cover o = 5 .. 4 + 4 nsh, all below the tile's smallest threshold_o; a tile of the
            // long part (threshold_o - 1 > chunk) is contracted chunk by chunk, without them
            const int n = (tiles[t].t_lo - 5) / 4;
            qo.tile_nsh[t] = (tiles[t].t_hi - 1 <= chunk && n >= min_shared) ? n : 0;
        }
    }
    // useful flops of the contraction per row (covest_grid_work): 2 per (column, o < T) of the MFMA steps, 2 per
    // (o mod 4 lane, shared step) and the 4-term MFMA per column that brings a shared sum in
    g->contract_flops_per_row = 0.0;
    for (int32_t t = 0; t < qo.n_qtiles; ++t) {
        const int n = qo.tile_nsh[(size_t)t];
        for (int i = 0; i < 16; ++i) {
            const int32_t qi = qo.order[(size_t)t * 16 + (size_t)i];
            if (qi >= 0)
                g->contract_flops_per_row += 2.0 * (double)((int)t_table[(size_t)qi] - 1 - 4 * n) + (n ? 8.0 : 0.0);
        }
        g->contract_flops_per_row += 8.0 * n;
    }
    // the long weight vectors are the first tiles of the order
    int32_t n_long_tl = 0;
    while (n_long_tl < qo.n_qtiles && (int)t_table[(size_t)qo.order[(size_t)n_long_tl * 16]] - 1 > chunk)
        ++n_long_tl;
    const int32_t n_long_tiles = n_long_tl;
    if (n_long_tiles > 0) {
        const int n_chunks = (qo.t_max - 1 + chunk - 1) / chunk;
        g->long_parts.resize((size_t)n_chunks);
        for (int c = 0; c < n_chunks; ++c) {
            // tiles that still have copy numbers in this chunk: a prefix (sorted by threshold_o)
            int32_t hi = 0;
            while (hi < n_long_tiles && (int)t_table[(size_t)qo.order[(size_t)hi * 16]] - 1 > c * chunk)
                ++hi;
            if (hi == 0) {
                g->long_parts.resize((size_t)c);
                break;
            }
            covest_grid::Part &part = g->long_parts[(size_t)c];
            const int rc = build_plan_part(g, axes, t_table, qo, 0, hi, c * chunk, chunk, n_pass, 3, part.buf, part.plan,
                                           c == 0 ? &g->long_q_orig_host : nullptr);
            if (rc != COVEST_OK)
                return rc;
        }
        g->n_long_tiles = n_long_tiles;
        // q_orig of the long slots on the device, for ll_finish_dense (padded to whole tiles)
        g->long_q_orig_host.resize((size_t)n_long_tiles * 16, -1);
        HIP_TRY(g->long_q_orig.reserve(g->long_q_orig_host.size() * sizeof(int32_t)));
        HIP_TRY(hipMemcpy(g->long_q_orig.ptr, g->long_q_orig_host.data(), g->long_q_orig_host.size() * sizeof(int32_t),
                          hipMemcpyHostToDevice));
    }
    g->has_short_part = n_long_tiles < qo.n_qtiles;
    if (g->has_short_part) {
        const int rc = build_plan_part(g, axes, t_table, qo, n_long_tiles, qo.n_qtiles, 0, chunk, n_pass, 0, g->plan_buf,
                                       g->plan, nullptr);
        if (rc != COVEST_OK)
            return rc;
    } else {
        g->plan = FactoredPlan{};
        g->plan.n_q = qo.nq;
        g->plan.ce_begin = g->flat_begin / qo.nq;
        g->plan.ce_end = (g->flat_end + qo.nq - 1) / qo.nq;
        g->plan.max_o = 0;
    }
    g->has_plan = qo.t_max >= 2;
    return COVEST_OK;
}

// K-factored on a POINT LIST (tiles.h FactoredPlan::list_mode): every point is its own (c, e) workgroup with
// a q-tile of one real column.  What a refinement step needs -- a handful of points, each a full likelihood --
// then costs one workgroup's pass over the keys (the recurrence over o in 5 waves, a few MFMAs) instead of
// K-direct's single wave looping over every (key, o, s).  Built per call: ~13 KB of tables per point.
// An item is a point (o_base 0, list_mode 1) or a chunk of a point's copy numbers (list_mode 2): params of the
// point, threshold_o of the point, copy numbers before the chunk.
// in_place: the kernel reads the tables where they are staged -- page-locked host memory mapped into the device's
// address space -- instead of from a copy in HBM: for a handful of points (13 KB of tables each, read once by the
// point's 8 workgroups) the reads over the link cost less than the copy engine's start-up, 15-20 us of a single
// evaluation's 65.
int build_list_plan(covest_model *m, int64_t n, const double *params, const std::vector<int32_t> &t_list,
                    const std::vector<int32_t> *o_base_list, DevBuf &buf, FactoredPlan &pl, bool in_place)
{
    constexpr int NW = 8, MU = kMaxUnits;
    auto o_base_of = [&](int64_t i) { return o_base_list ? (*o_base_list)[(size_t)i] : 0; };
    int t_max = 1; // largest LOCAL threshold: copy numbers of an item are o_base + 1 .. o_base + t_local - 1
    for (int64_t i = 0; i < n; ++i)
        t_max = std::max(t_max, std::min(513, (int)t_list[(size_t)i] - o_base_of(i)));
    const int ld = ((t_max - 1 + 31) / 32) * 32 + 2;
    const int n_buf = (2 * (size_t)kTileBins * ld + 64) * sizeof(double) + 13440 <= 160 * 1024 ? 2 : 1; // (+ the kernel's static LDS: log table, hand-back records, row constants)
    const size_t n_slots = (size_t)n * 16, n_blocks = 1 + 2 * (size_t)n, n_unit = n_blocks * MU;
    std::vector<double> axes(2 * (size_t)n), r4(n_slots, 0.0), piece_w(n_unit * 64 * 2, 0.0);
    std::vector<int32_t> q_t(n_slots, 0), q_orig(n_slots, -1), unit_tile(n_unit, -1), unit_half(n_unit, 0),
        unit_s0(n_unit, 0), unit_o0(n_unit, 1), unit_len(n_unit, 0), unit_cont(n_unit, 0), unit_pair(n_unit, -1);
    for (int64_t p = 0; p < n; ++p) {
        const double *par = params + p * 5;
        axes[(size_t)p] = par[0];
        axes[(size_t)n + (size_t)p] = par[1];
        const double q1 = clamp_one(m->dm, 2, par[2]), q2 = clamp_one(m->dm, 3, par[3]), q = clamp_one(m->dm, 4, par[4]);
        const int ob = o_base_of(p);
        // local threshold: the kernel's lanes count from the chunk's start, and a chunk ends after 512 copy numbers
        const int t = std::min(513, std::max(0, (int)t_list[(size_t)p] - ob));
        const size_t slot = (size_t)p * 16;
        q_t[slot] = t;
        q_orig[slot] = 0;
        r4[slot] = std::pow(1 - q, 4.0);
        const int steps = t > 1 ? (t - 1 + 3) / 4 : 0;
        // the two halves of the key tile go to the workgroup's last two waves, each cut into equal pieces
        const int pieces = std::max(1, std::min(MU, steps / kMinPieceSteps));
        const int piece_len = std::max(1, (steps + pieces - 1) / pieces);
        for (int h = 0; h < 2; ++h)
            for (int k = 0; k < pieces; ++k) {
                const size_t at = (1 + 2 * (size_t)p + (size_t)h) * MU + (size_t)k;
                unit_tile[at] = (int32_t)p;
                unit_half[at] = h;
                unit_s0[at] = k * piece_len;
                unit_o0[at] = 1 + 4 * k * piece_len;
                unit_len[at] = piece_len;
                unit_cont[at] = k > 0;
                if (h == 0 && k == 0) // its half-1 partner: the first slot of the workgroup's last wave (tiles.h unit_pair)
                    unit_pair[at] = (NW - 1) * MU;
                for (int which = 0; which < 2; ++which)
                    for (int kq = 0; kq < 4; ++kq) { // column 0 only: lanes 16 kq
                        const int o_local = 1 + 4 * (unit_s0[at] + which) + kq;
                        // (the piece's first step comes masked by the cut-off, as in build_plan_part)
                        piece_w[(at * 64 + (size_t)(16 * kq)) * 2 + (size_t)which] =
                            (which == 0 && o_local >= t) ? 0.0 : copy_number_weight_host(q1, q2, q, ob + o_local);
                    }
            }
    }
    std::vector<std::pair<const void *, size_t>> dparts = {{axes.data(), axes.size()}, {r4.data(), r4.size()},
                                                           {piece_w.data(), piece_w.size()}};
    std::vector<std::pair<const void *, size_t>> iparts = {
        {q_t.data(), q_t.size()},             {q_orig.data(), q_orig.size()},       {unit_tile.data(), unit_tile.size()},
        {unit_half.data(), unit_half.size()}, {unit_s0.data(), unit_s0.size()},     {unit_len.data(), unit_len.size()},
        {unit_cont.data(), unit_cont.size()}, {unit_o0.data(), unit_o0.size()}, {unit_pair.data(), unit_pair.size()}};
    size_t n_dbl = 0, n_int = 0;
    for (auto &pr : dparts)
        n_dbl += pr.second;
    for (auto &pr : iparts)
        n_int += pr.second;
    // one staging buffer (page-locked, the model's), one copy
    const size_t stage_bytes = n_dbl * sizeof(double) + n_int * sizeof(int32_t);
    HIP_TRY(m->ws_stage.reserve(stage_bytes));
    char *stage = m->ws_stage.as<char>();
    if (!in_place)
        HIP_TRY(buf.reserve(stage_bytes));
    double *dbase = in_place ? reinterpret_cast<double *>(stage) : buf.as<double>();
    int32_t *ibase = reinterpret_cast<int32_t *>(dbase + n_dbl);
    std::vector<const double *> dptr;
    std::vector<const int32_t *> iptr;
    size_t off = 0;
    for (auto &pr : dparts) {
        std::memcpy(stage + off * sizeof(double), pr.first, pr.second * sizeof(double));
        dptr.push_back(dbase + off);
        off += pr.second;
    }
    off = 0;
    for (auto &pr : iparts) {
        std::memcpy(stage + n_dbl * sizeof(double) + off * sizeof(int32_t), pr.first, pr.second * sizeof(int32_t));
        iptr.push_back(ibase + off);
        off += pr.second;
    }
    // (asynchronous, from the model's page-locked staging memory: the launch queues up behind the copy on the null stream,
    // and the caller waits for the stream before it builds another list -- covest_eval_points)
    if (!in_place)
        HIP_TRY(hipMemcpyAsync(buf.ptr, stage, stage_bytes, hipMemcpyHostToDevice, nullptr));
    pl = FactoredPlan{};
    pl.c_axis = dptr[0];
    pl.e_axis = dptr[0] + n;
    pl.n_e = 1;
    pl.ce_begin = 0;
    pl.ce_end = n;
    pl.n_q = 1;
    pl.n_qtiles = (int32_t)n;
    pl.max_o = t_max - 1;
    pl.n_pass = 1;
    pl.pass_stride = ((t_max - 1 + 3) / 4) * 4;
    pl.n_columns = t_max - 1;
    pl.o_base = 0;
    pl.n_threads = NW * 64;
    pl.half_units = kHalfUnits;
    pl.n_qblocks = 1;
    pl.ld = ld;
    pl.n_buf = n_buf;
    pl.q_r4 = dptr[1];
    pl.piece_w = dptr[2];
    pl.q_T = iptr[0];
    pl.q_orig = iptr[1];
    pl.unit_tile = iptr[2];
    pl.unit_half = iptr[3];
    pl.unit_s0 = iptr[4];
    pl.unit_len = iptr[5];
    pl.unit_cont = iptr[6];
    pl.unit_o0 = iptr[7];
    pl.unit_pair = iptr[8];
    pl.unit_nsh = nullptr; // (list modes are not the PLAIN kernel: never read)
    pl.unit_rho = nullptr;
    pl.qtile_nsteps = nullptr;
    pl.qtile_nfull = nullptr;
    pl.q_first8 = nullptr;
    pl.flat_begin = 0;
    pl.flat_end = n;
    pl.list_mode = 1;
    // (one value for every point list, whatever it holds: a point's value must not depend on its company)
    pl.p_clamp = clamp_for(m, 513);
    pl.n_seg = std::max(1, std::min(kListSegments, (int)m->tv.n_items)); // a function of the histogram alone
    pl.item_obase = nullptr;
    pl.partial = nullptr;
    pl.diag = nullptr;
    pl.skip_phases = 0;
    return COVEST_OK;
}


} // namespace covest
