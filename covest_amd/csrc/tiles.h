// tiles.h -- the tile table of the pmf recurrence (see streams.h), shared by the
// host builder (tiles_host.cpp: build_tiles) and the fast kernels.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

namespace covest {

constexpr int kTileBins = 32;
constexpr int kScaleBits = 540;
constexpr double kScaleLn = 540.0 * 0.693147180559945309417232121458; // ln 2^SC
constexpr double kWindowLn = -760.0; // terms below e^-760 are 0 in double
// TileView::scal carries 2^kBasicShift on top: K-basic's p_j (streams' sum x key's scale) are p_j 2^64 -- nothing
// overflows (p_j <= 2.5), and a p_j below half a grid step of the doubles does not flush to 0 in the product, so the
// kernel can tell a zero of the reference from a row for the strict evaluation (direct_point.h kZeroSteps)
constexpr int kBasicShift = 64;
constexpr double kBasicScale = 0x1p64;

// Tiles of <= 32 consecutive keys over the evaluated bins (built on the host,
// tiles_host.cpp: build_tiles).  All arrays live in one device buffer; everything
// indexed by tile is wave-uniform and read through the scalar cache.
//
// ITEMS: what a kernel walks is a list of items, each one LDS buffer's worth of K-factored's phases B/C.  A
// "plain" item is one tile.  A "sum" item stands for up to 32 consecutive tiles whose counts are ALL zero
// (they exist only when tail != 0: then every key enters sp_j, covest/models.py:103, and a trimmed histogram
// is mostly such keys): none of their keys takes a log, only the sum of their p_j is needed, and
// sum_j sum_o b_o G[o][j] = sum_o b_o (sum_j G[o][j]) -- so the builders add G over a tile's keys in registers
// and the item's 32 rows are those per-tile sums: one contraction and no log for 1024 keys.
// A tile's constants in ONE 64-byte record (round 5): a kernel entering a tile fetches it with one scalar load of one
// cache line, where the six values used to come from six arrays -- six lines, and two round trips through the scalar
// cache, the second load's address waiting for n_bins.
struct alignas(64) TileRec {
    double k0, lgam_prev, lgam_last, renorm;         // first_key, ln (k0-1)!, ln (k0+nb-1)!, (k0-1)!/(k0+nb-1)!
    int32_t nb, run_start, all_zero, has_filler;     // n_bins and the tile's flags
    double pad[2];
};
static_assert(sizeof(TileRec) == 64, "one cache line");

struct TileView {
    int32_t n_tiles;
    int32_t n_items;
    // raw buffers behind the views below: [4*nt + 2*nt*32 + 3*ni*32 + ni + 5*(nt+1) + 2] doubles, [4*nt + 3*ni + 32*nt] int32.  The fast
    // kernels take these two as separate `const __restrict__` kernel arguments and rebuild the
    // view from them (tile_view_from): only then does hipcc know the table is read-only and
    // never aliased, and fetches the wave-uniform entries with s_load into SGPRs.
    const double *dbl_base;
    const int32_t *int_base;
    const double *first_key;   // [n_tiles] k0 as a double
    const int32_t *n_bins;     // [n_tiles] keys in the tile (1..32)
    const int32_t *run_start;  // [n_tiles] 1: keys are not contiguous with the previous tile -> re-anchor
    const int32_t *all_zero;   // [n_tiles] 1: every count of the tile is 0 (its keys only enter sp_j: tail != 0)
    const int32_t *has_filler; // [n_tiles] 1: a row of the tile is a filler key (a gap of the histogram: scale 0)
    const double *lgam_prev;   // [n_tiles] lgamma(k0)       = ln (k0-1)!
    const double *lgam_last;   // [n_tiles] lgamma(k0 + nb)  = ln (k0+nb-1)!
    const double *renorm;      // [n_tiles] (k0-1)! / (k0+nb-1)!   carries v into the next tile
    const double *scal;        // [n_tiles][32] 2^(kBasicShift-SC) (k0-1)!/(k0+b)!; 0 for filler keys (gaps of the histogram the
                               //   recurrence walks through) and padding: their p_j is exactly 0, so they add nothing to sp_j
    const double *cnt;         // [n_tiles][32] h_j (0 for padding and filler keys)
    const double *item_cnt;    // [n_items][32] the counts of an item's 32 rows: its tile's for a plain item, 0 for a sum item
    // K-factored keeps the rows of a PLAIN item UNSCALED (ll_factored.hip: G' = sum of the streams' scaled terms,
    // p_j = P'_j * scal_j): the scale enters the logs as a constant, log p_j = log(P'_j 2^-SC) + ln((k0-1)!/(k0+b)!)
    const double *item_scal;   // [n_items][32] scal of the row (0: filler / padding); 1 for the rows of a sum item (they ARE scaled)
    const double *item_iscal;  // [n_items][32] 1 / scal of a row WITH a count (0: filler / padding / zero count / sum item): p_clamp in the row's units
    const double *item_lconst; // [n_items] sum over the item's rows of h_j ln((k0-1)!/(k0+b)!) (long double on the host)
    // K-basic's closed form over the tiles t .. n_tiles - 1 (ll_basic.hip: one stream left, no tail): sums over their
    // COUNTED keys, long double on the host
    const double *suf_h;       // [n_tiles + 1] sum h_j
    const double *suf_jh;      // [n_tiles + 1] sum j h_j
    const double *suf_lgh;     // [n_tiles + 1] sum h_j ln j!
    const double *suf_first;   // [n_tiles + 1] the first counted key of those tiles (0: none) and
    const double *suf_first_lg; // [n_tiles + 1]   its ln j!
    const double *last_key;    // [2] the last counted key of the table and its ln j!
    const int32_t *item_first; // [n_items] first tile of the item
    const int32_t *item_ntiles; // [n_items] 1 for a plain item, 1..32 (rows in use) for a sum item
    const int32_t *item_sum;   // [n_items] 1 = sum item
    const int32_t *row_bin;    // [n_tiles][32] index of the row's key in DevModel::bins (-1: filler / padding): how a
                               //   recurrence kernel names a key it hands back (direct_point.h)
    const TileRec *rec;        // [n_tiles] the per-tile values above, gathered (64-byte aligned: tile_dbl_count)
};

// doubles of the raw double buffer: the arrays, rounded up to a whole cache line, then the records
inline __host__ __device__ int64_t tile_arrays_dbl(int32_t nt, int32_t ni)
{
    const int64_t n = 4 * (int64_t)nt + 2 * (int64_t)nt * kTileBins + 3 * (int64_t)ni * kTileBins + ni + 5 * ((int64_t)nt + 1) + 2;
    return (n + 7) / 8 * 8;
}
inline __host__ __device__ int64_t tile_dbl_count(int32_t nt, int32_t ni) { return tile_arrays_dbl(nt, ni) + 8 * (int64_t)nt; }

// The layout of tiles_host.cpp: build_tiles.
inline __host__ __device__ TileView tile_view_from(int32_t nt, int32_t ni, const double *dbl, const int32_t *ints)
{
    TileView tv;
    tv.n_tiles = nt;
    tv.n_items = ni;
    tv.dbl_base = dbl;
    tv.int_base = ints;
    tv.first_key = dbl;
    tv.lgam_prev = dbl + nt;
    tv.lgam_last = dbl + 2 * (int64_t)nt;
    tv.renorm = dbl + 3 * (int64_t)nt;
    tv.scal = dbl + 4 * (int64_t)nt;
    tv.cnt = tv.scal + (int64_t)nt * kTileBins;
    tv.item_cnt = tv.cnt + (int64_t)nt * kTileBins;
    tv.item_scal = tv.item_cnt + (int64_t)ni * kTileBins;
    tv.item_iscal = tv.item_scal + (int64_t)ni * kTileBins;
    tv.item_lconst = tv.item_iscal + (int64_t)ni * kTileBins;
    tv.suf_h = tv.item_lconst + ni;
    tv.suf_jh = tv.suf_h + (nt + 1);
    tv.suf_lgh = tv.suf_jh + (nt + 1);
    tv.suf_first = tv.suf_lgh + (nt + 1);
    tv.suf_first_lg = tv.suf_first + (nt + 1);
    tv.last_key = tv.suf_first_lg + (nt + 1);
    tv.n_bins = ints;
    tv.run_start = ints + nt;
    tv.all_zero = ints + 2 * (int64_t)nt;
    tv.has_filler = ints + 3 * (int64_t)nt;
    tv.item_first = ints + 4 * (int64_t)nt;
    tv.item_ntiles = tv.item_first + ni;
    tv.item_sum = tv.item_ntiles + ni;
    tv.row_bin = tv.item_sum + ni;
    tv.rec = reinterpret_cast<const TileRec *>(dbl + tile_arrays_dbl(nt, ni));
    return tv;
}

// Work description of K-factored (ll_factored.hip), built at covest_grid_create
// for a dense repeats-model grid.  The Q = |q1| x |q2| x |q| weight vectors are
// sorted by threshold_o (descending) into "slots", 16 per q-tile.  The unit of
// matrix work is (q-tile, half) -- 16 weight vectors x 16 of a tile's 32 keys --
// costing ceil((T-1)/4) MFMAs per key tile.  Units are dealt to the waves of a
// workgroup longest-first, balanced per SIMD (waves w and w+4 share one); a wave
// with free accumulator slots cuts its longest units into equal PIECES (disjoint
// ranges of MFMA steps) that run concurrently in adjacent slots and are added before
// the logs: a single dependent chain of MFMAs with its LDS read and weight update
// keeps the fp64 pipe busy only a third of the time.  A wave's slots are sorted by
// length, so the active ones are always a prefix and the step loop is specialised
// on their number (no per-slot branches inside it).
constexpr int kMaxUnits = 6;       // accumulator slots per wave
constexpr int kHalfUnits = 3;      // (kept for the 256-thread capacity rule: 4 waves x 6 slots)
constexpr int kBuildCost = 18;     // phase A of one wave and key tile, in MFMA-step equivalents (tuned on C3; 36 before round 3 took the scales and the scalar-load stalls out of phase A)
constexpr int kBuildCostLowKeys = 24; // ... of a histogram whose tiles are (nearly) all at low keys, where every error class is
                                   //   live in every tile -- what the reference's trimming leaves (380 keys of H10k_rep): swept on it in
                                   //   round 5, 0.4015 against 0.4103 ms at 18 (profiles/r05_factored_builder_charge_sweep.txt); C3 itself,
                                   //   a third of whose tiles are such, stays best at 18
constexpr int kLowKeyTile = 384;   // a tile that starts at or below this key counts as one of those
#ifndef COVEST_LIST_SEGMENTS
#define COVEST_LIST_SEGMENTS 16 // (8 until round 5: one evaluation on a 10 000-key histogram 53 -> 45 us, six 67 -> 56; 64: 173 -> 215)
#endif
constexpr int kListSegments = COVEST_LIST_SEGMENTS; // key segments of a point-list evaluation (list mode): latency of ONE point
constexpr int kMinPieceSteps = 6;  // pieces are not made shorter than this
constexpr int kSharedStepsPerMfma = 4; // cost of the shared steps in the assignment: this many weigh one MFMA step (5 until the fine sweep at the end of round 4: -0.7 %)
constexpr int kMinSharedSteps = 3; // fewer shared steps than this are left to the MFMA steps
constexpr int kLastBuilderExtra = 4; // extra charge of the builder of the top copy numbers when it shares a SIMD with another builder
constexpr int kUnitOverhead = 2;   // per-unit cost besides its MFMA steps (logs, setup), same unit

struct FactoredPlan {
    const double *c_axis, *e_axis; // device copies of axes 0 and 1
    int64_t n_e;                   // len(e axis): ce = ic * n_e + ie
    int64_t ce_begin, ce_end;      // (c, e) pairs this block covers
    int64_t n_q;                   // Q
    int32_t n_qtiles;              // ceil(Q / 16)
    int32_t max_o;                 // copy numbers to build (of this chunk): max LOCAL threshold_o - 1
    int32_t n_pass;                // lanes per copy number: ceil(max_error / 8); each takes 8 error classes
    int32_t pass_stride;           // columns of G per pass (max_o rounded up to a multiple of 4)
    int32_t n_columns;             // n_pass * pass_stride (n_pass == 1: max_o): columns of G that are built
    int32_t o_base;                // list_mode 3: copy numbers before this launch's chunk
    int32_t n_threads;             // workgroup size the unit tables were built for (256, 512 or 768)
    int32_t half_units;            // unit slots per wave and half (kHalfUnits, or 2 with 768 threads)
    int32_t n_qblocks;             // workgroups per (c, e) (gridDim.y); each rebuilds G
    int32_t ld;                    // G row stride in doubles: roundup32(max_o) + 2 (= 4 dwords mod 64)
    int32_t n_buf;                 // 2: G double-buffered in LDS (build tile t+1 while contracting tile t)
    // per accumulator slot, [n_qblocks][n_threads/64][kMaxUnits], sorted by length within a wave:
    const int32_t *unit_tile;      // q-tile of the slot, -1 = none
    const int32_t *unit_half;      // 0 = keys 0..15 of the key tile, 1 = keys 16..31
    const int32_t *unit_s0;        // first MFMA step of the slot's piece (a step = 4 columns of G)
    const int32_t *unit_o0;        // the copy number (1-based inside the chunk) of that step's first column
    const int32_t *unit_len;       // steps of the piece (equal for all pieces of a unit; steps past the
                                   //   unit's end are masked by the T cut-off)
    const int32_t *unit_cont;      // 1 = this slot continues the unit of the slot before it
    const int32_t *unit_nsh;       // SHARED steps of the slot's unit (0: none), see below
    const int32_t *unit_pair;      // first slot of a half-0 unit: wave * kMaxUnits + slot (inside the workgroup) of the first
                                   //   slot of the same q-tile's half-1 unit (-1: none); the kernel's last step adds the two
    const double *unit_rho;        // [slots][4] units with unit_nsh > 0: {(1-q)^16, unused, (1-q)^4, (1-q)^(-4 nsh)} of the unit's q-tile
    const double *piece_w;         // [slots][64 lanes][2] b_o at the piece's first step (masked by the column's
                                   //   cut-off) and at its second (o = 1 + 4 step + lane/16, column lane%16), libm
                                   //   pow on the host; units with unit_nsh > 0: second = the first step AFTER the
                                   //   shared ones.  16-byte aligned (one load per slot and lane)
    // Shared steps.  The 16 weight vectors of a q-tile of a dense grid can be chosen to differ in q1 and q2 only
    // (the host sorts the product that way when it pays): then b_o = beta_col (1-q)^(o-3) for o >= 3, and for
    // the copy numbers below EVERY column's cut-off the contraction  sum_o G[key][o] b_o(col)  is beta_col times a
    // sum that does not depend on the column.  Steps 1 .. unit_nsh of such a unit (o = 5 .. 4 + 4 nsh, all below
    // the tile's smallest T) are therefore summed ONCE per key, on the vector unit (Horner in (1-q)^4, lane =
    // (key, o mod 4), weights relative to the FIRST shared step: they only fall), and enter the accumulator through one
    // MFMA whose B is b_o of that step; step 0 (o = 1 .. 4, the weights that are not geometric) and the steps after the shared
    // ones (where the columns' cut-offs differ) are MFMA steps as before.  unit_len counts step 0 and the steps after
    // the shared ones.
    const int32_t *qtile_nsteps;   // [n_qtiles] ceil((max T in tile - 1) / 4): MFMA steps of the tile
    const int32_t *qtile_nfull;    // [n_qtiles] floor((min T in tile - 1) / 4): steps with no column cut off
    const int32_t *q_T;            // [n_qtiles*16] threshold_o per slot (0 = padding)
    const int32_t *q_orig;         // [n_qtiles*16] index into the (q1,q2,q) product (-1 = padding)
    const double *q_first8;        // [8][n_qtiles*16] b_o, o = 1..8   (covest/models.py:193-208)
    const double *q_r4;            // [n_qtiles*16] (1 - q)^4
    int64_t flat_begin, flat_end;  // flat indices whose LL is written (ragged block ends)
    int32_t list_mode;             // 3: a dense grid's long weight vectors, one chunk of copy numbers (see partial).
                                   // 1, 2: a POINT LIST, not a grid: workgroup i (gridDim.y == 1) evaluates ITEM i =
                                   //   (c_axis[i], e_axis[i]) with the single weight vector of q-tile i (slot 16 i;
                                   //   n_q == 1).  The unit tables then hold one empty wave block followed by two
                                   //   blocks per item, for the workgroup's last two waves (ll_factored.hip).
                                   //   1: an item is a whole point, its LL is written.  2: an item is a CHUNK of a
                                   //   point's copy numbers, o = item_obase[i] + 1 .. + 512 (threshold_o beyond one
                                   //   workgroup's lanes); the workgroup stores its share of p_j to `partial` and
                                   //   ll_finish_partials adds the chunks and takes the logs
    double p_clamp;                // (S + max threshold_o) * 7e-317: below it a p_j is handed back (direct_point.h)
    int32_t n_seg;                 // list modes: key tiles are cut into n_seg contiguous segments, workgroup
                                   //   blockIdx.x = unit * n_seg + segment (unit = point or chunk); a segment starts
                                   //   like a run (streams anchored).  list_mode 1 then writes {LL part, sp part hi, lo}
                                   //   per (point, segment) to `partial` and the host adds the segments in order
    const int32_t *item_obase;     // [units] list_mode 2: copy numbers before this chunk (multiple of 512); else NULL
    double *partial;               // list_mode 2: [units][n_items * 32] sum over the chunk's o of b_o G[o][key];
                                   //   list_mode 1: [points][n_seg][4]
                                   //   list_mode 3: [ce - ce_first][n_cols_partial][n_items * 32] p_j of the grid's long
                                   //   weight vectors (threshold_o beyond a workgroup's lanes), summed chunk by chunk
    int64_t ce_first;              // list_mode 3: the (c, e) of partial's first row
    int64_t n_cols_partial;        // list_mode 3: weight-vector slots per (c, e) in partial
    // The two fields below are read by DIAGNOSTIC builds only (-DCOVEST_DIAG, built to tools/bin/ by
    // `python -m covest_amd.build --out tools/bin/libcovest_amd_diag.so -DCOVEST_DIAG` and loaded through
    // COVEST_AMD_LIB): the shipped library never looks at them -- a knob that changes VALUES is a parity hazard.
    long long *diag;               // (env COVEST_FACTORED_DIAG) per-wave s_memtime sums [wg][wave][8]
    int32_t skip_phases;           // (env COVEST_FACTORED_SKIP) bit 0/1/2 skips phase A/B/C, bit 3 the shared steps; results are wrong
};

#ifdef COVEST_DIAG
#define COVEST_SKIP_PHASE(plan, bits) (((plan).skip_phases & (bits)) != 0)
#else
#define COVEST_SKIP_PHASE(plan, bits) false
#endif

} // namespace covest
