// kernels.h -- host-callable launchers of the gfx950 kernels (ll_direct.hip,
// argmin.hip, ...).  All launches are asynchronous on `stream`.
#pragma once
#include <hip/hip_runtime_api.h>

#include "device_model.h"
#include "direct_point.h"
#include "tiles.h"

namespace covest {

// K-direct: one wavefront per grid point, one exp per pmf term (ll_direct.hip).
// out_ll[n]; when out_p != nullptr (n must be 1) also writes p_j for every bin
// of `m.bins`.
// ref_overflow: the reference's long-double overflow reproduced (direct_point.h REF_OVF; out_p must be nullptr).
hipError_t launch_ll_direct(const DevModel &m, const PointSource &src, int64_t n, double *out_ll,
                            double *out_p, hipStream_t stream, bool ref_overflow = false);

// K-basic: basic model, one lane per grid point, pmf recurrence (ll_basic.hip).
// Needs n_err == 8 and a tile table (keys in 1..16384).
// sub_list: the queue of points handed back (direct_point.h); run launch_ll_fix_list after this.
hipError_t launch_ll_basic(const DevModel &m, const TileView &tv, const PointSource &src, int64_t n,
                           double *out_ll, const SubList &sub_list, hipStream_t stream);

// K-factored: repeats model on a dense grid, one workgroup per (c, e)
// (ll_factored.hip).  out_ll is the block's LL buffer (index flat - plan.flat_begin).
// sub_list: the queue of points handed back (direct_point.h; dense grids -- list mode 1 hands the side words over
// in `partial`, list mode 2 leaves the strict evaluation to ll_finish_partials); run launch_ll_fix_list after this.
hipError_t launch_ll_factored(const DevModel &m, const TileView &tv, const FactoredPlan &plan,
                              double *out_ll, const SubList &sub_list, hipStream_t stream);

// Chunked point list (tiles.h FactoredPlan::list_mode 2): combine the chunks' shares of p_j per point and
// take the logs.  first_item[n_points + 1] delimits each point's chunks; point_par[5 n_points] and point_T[n_points]
// are the points' parameters and threshold_o.
hipError_t launch_ll_finish_partials(const DevModel &m, const TileView &tv, const double *partial,
                                     const int32_t *first_item, const double *point_par, const int32_t *point_T,
                                     int64_t n_points, double *out_ll, hipStream_t stream);

// Dense grids, long weight vectors (tiles.h FactoredPlan::list_mode 3): take the logs of the p_j that the chunk
// launches summed into `partial` ([n_ce][n_cols][n_items * 32], first row = (c, e) number ce_first); q_orig[n_cols] maps
// a slot to its index in the (q1, q2, q) product (-1: padding); values go to out_ll[flat - src.flat_begin].
hipError_t launch_ll_finish_dense(const DevModel &m, const TileView &tv, const PointSource &src, const double *partial,
                                  int64_t ce_first, int64_t n_ce, int64_t n_cols, const int32_t *q_orig, int64_t n_q,
                                  int64_t flat_end, double *out_ll, hipStream_t stream);

// The pass after every K-basic / K-factored launch: one wave per point of the queue `list` (direct_point.h) adds
// the strict evaluation of the rows named in its side word to ll[], in place.  The queue's counter must be zero
// before the NEXT recurrence launch: launch_argmin resets it (grids), the host does for point lists.
// n_points: how many points the launch before evaluated (the queue cannot be longer; 0: unknown) -- sizes the launch.
hipError_t launch_ll_fix_list(const DevModel &m, const TileView &tv, const PointSource &src, double *ll,
                              const SubList &list, hipStream_t stream, int64_t n_points = 0);

// (min -LL, lowest index) over ll[n]: two-stage reduction (argmin.hip).
// partial_val/partial_idx need kArgminBlocks entries; result[0] = {min, bits of idx}.
constexpr int kArgminBlocks = 1024; // (256: 7.4 us for the 10^6 points of C2; four workgroups a CU hide the loads better)
struct ArgminResult {
    double min_negll;
    int64_t index;  // local index, -1 if no value is < +inf
    double pair[2]; // {min_negll, GLOBAL flat index as a double (-1 if none)}: what the ranks exchange
};
// queue_count: the hand-back queue's counter to reset (nullptr: none).
constexpr int64_t kArgminSmall = 16384; // grids up to this size: one workgroup, one launch
// host_mirror: page-locked host memory the winner is stored to as well (nullptr: none)
hipError_t launch_argmin(const double *ll, int64_t n, int64_t flat_begin, double *partial_val, int64_t *partial_idx,
                         ArgminResult *result, ArgminResult *host_mirror, unsigned *queue_count, hipStream_t stream);

// The same reduction and, beside it, the selection scan of covest/grid.py:65-70 started from `start`: the strict
// running-minimum records below it, in index order, written to `scan` (page-locked host memory).  n <= kArgminSmall.
constexpr int kScanCap = 120;
struct ScanRecords {
    int32_t n;         // records listed (written LAST, behind a system-scope fence)
    int32_t truncated; // 1: there were more than kScanCap -- read the values back instead
    double start;      // the minimum the scan started from (echo)
    struct {
        int64_t index; // GLOBAL flat index
        double negll;
    } rec[kScanCap];
};
hipError_t launch_argmin_scan(const double *ll, int64_t n, int64_t flat_begin, double start, ArgminResult *result,
                              ArgminResult *host_mirror, ScanRecords *scan, unsigned *queue_count, hipStream_t stream);

// ---- K-kmer: k-mer abundance histogram (kmer_count.hip), SURVEY 8(f) row F1 ----
// Open-addressing table in HBM, slots = 2^log2_slots, one 16-byte entry per slot: {key, count}
// (key all-ones = empty).  Key and count share a cache line on purpose: a k-mer costs ONE scattered
// line (relaxed load of the key + atomic add on the neighbouring count), not two.
struct KmerSlot {
    unsigned long long key;
    unsigned long long count;
};
struct KmerTable {
    KmerSlot *slots;
    unsigned long long mask;
    int log2_slots;
    int k; // bases per key (a key's first slot is a function of its minimizer: kmer_count.hip)
};
hipError_t launch_kmer_fill_empty(const KmerTable &t, hipStream_t stream);
// bases: ASCII acgt/ACGT; offsets[n_reads + 1] or nullptr with every read `fixed_len` long.
hipError_t launch_kmer_count(const unsigned char *bases, const int64_t *offsets, int64_t n_reads,
                             int64_t fixed_len, int k, int canonical, const KmerTable &t, int *overflow,
                             hipStream_t stream);
hipError_t launch_kmer_rehash(const KmerTable &src, const KmerTable &dst, int *overflow, hipStream_t stream);
hipError_t launch_kmer_stats(const KmerTable &t, unsigned long long *stats, hipStream_t stream);
hipError_t launch_kmer_histogram(const KmerTable &t, unsigned long long *hist, unsigned long long hist_len,
                                 hipStream_t stream);

// ---- K-kmer, partitioned (kmer_bulk.hip): minimizer buckets of super-k-mer records in HBM, counted in LDS ----
constexpr int kOvfShards = 64, kOvfStride = 16;
struct KmerBulk {
    int k, m, w;          // k-mer length, minimizer length, m-mers per k-mer (k - m + 1)
    int canonical;
    int log2_buckets;
    int max_run;          // windows per record: 32 - k + 1 (a record holds at most 32 bases)
    int sample;           // pass 0 looks at 1 block of tiles (1 read) in `sample`
    unsigned *sampled;    // [buckets] records pass 0 counted
    typedef unsigned long long fill_t; // (32-bit cursors are no faster -- measured)
    ulonglong2 *ctl;      // [buckets] {first, end}: the bucket's places in recs
    fill_t *fill;         // [buckets] records sent to the bucket (beyond its room: it overflowed, the surplus is in `overflow`)
    ulonglong2 *recs;     // {bases as 2-bit codes, base i at bits 2i; number of bases}
    // the overflow list, in kOvfShards parts with a counter each (ONE counter for all of it saturates at ~90 adds per
    // microsecond: 10 ms for the 9e5 records that overflow at 10 Gbp); a workgroup writes to the part of its number
    ulonglong2 *overflow;          // [kOvfShards][overflow_cap]
    unsigned long long *ovf_count; // [kOvfShards * kOvfStride] records sent to a part (beyond overflow_cap: lost -- the
                                   // caller starts over); a counter per 128-byte line
    unsigned long long overflow_cap; // per part
};
int kmer_bulk_block_bytes(const KmerBulk &p); // bytes of reads of one length a workgroup of pass 0/1 answers for
// offsets != nullptr: reads of any length -- base0 = offsets[0], total_bytes = offsets[n_reads] - base0, first_read: room
// for kmer_bulk_ragged_tiles(p, total_bytes) words
int64_t kmer_bulk_ragged_tiles(const KmerBulk &p, int64_t total_bytes);
hipError_t launch_kmer_scatter(const unsigned char *bases, const int64_t *offsets, int64_t n_reads, int64_t fixed_len,
                               int64_t base0, int64_t total_bytes, unsigned *first_read, const KmerBulk &p, bool count_only,
                               hipStream_t stream);
hipError_t launch_kmer_place_buckets(const KmerBulk &p, unsigned long long *partial, unsigned long long *total,
                                     hipStream_t stream);
// stats: [0] max count, [1] distinct keys, [2] entries of `big` (counts >= hist_len), [3] records pass 1 sent
// later: [0] buckets the wave-per-bucket kernel left to the workgroup-per-bucket one, later_list: [buckets]
// to_table: [0] buckets no LDS table could hold, [1] their k-mer occurrences, to_table_list: [buckets]
hipError_t launch_kmer_bucket_count(const KmerBulk &p, unsigned long long *hist, unsigned long long hist_len,
                                    unsigned long long *stats, unsigned long long *big, unsigned long long big_cap,
                                    unsigned *later, unsigned *later_list, unsigned long long *to_table, unsigned *to_table_list,
                                    bool small_buckets, int n_cu, hipStream_t stream);
// out[0] (1 on entry) = 0 unless every read is offsets[1] - offsets[0] bases long (n_reads >= 1)
hipError_t launch_kmer_one_length(const int64_t *offsets, int64_t n_reads, unsigned long long *out, hipStream_t stream);
// `ops` (rounded up to 64 per thread) returning atomic adds at pseudo-random places of words[slots]
hipError_t launch_kmer_scatter_rate(unsigned long long *words, unsigned long long slots, long long ops, unsigned long long *sink,
                                    hipStream_t stream);
// the overflow list's records and the listed buckets' into the table in HBM
hipError_t launch_kmer_to_table(const KmerBulk &p, bool any_overflowed, const KmerTable &t, int *overflow,
                                const unsigned long long *to_table, const unsigned *to_table_list, hipStream_t stream);

// ---- K-kmer for k > 31 (kmer_wide.hip): keys of w = 2, 4 or 8 words, slots of `stride` words {state/count, key[w]} ----
struct KmerWideTable {
    unsigned long long *words;
    unsigned long long mask; // slots - 1
    int log2_slots;
    int k;
    int w;      // 64-bit words per key
    int stride; // words per slot: 4, 8 or 16
};
hipError_t launch_kmer_wide_clear(const KmerWideTable &t, hipStream_t stream);
hipError_t launch_kmer_wide_count(const unsigned char *bases, const int64_t *offsets, int64_t n_reads, int64_t fixed_len,
                                  int canonical, const KmerWideTable &t, int *overflow, hipStream_t stream);
hipError_t launch_kmer_wide_rehash(const KmerWideTable &src, const KmerWideTable &dst, int *overflow, hipStream_t stream);
hipError_t launch_kmer_wide_stats(const KmerWideTable &t, unsigned long long *stats, hipStream_t stream);
hipError_t launch_kmer_wide_histogram(const KmerWideTable &t, unsigned long long *hist, unsigned long long hist_len,
                                      hipStream_t stream);

// ---- K-thin: expected histogram after down-sampling by `factor` (thin_hist.hip), SURVEY 8(f) row F3 ----
struct ThinSource {
    int32_t i;     // source count
    int32_t pad;
    double count;  // its multiplicity h_i
    double a;      // i < 100: ln i!            i >= 100: ln(i / factor)
    double b;      // i < 100: unused           i >= 100: i / factor
};
// src[n] and lgam[m] = ln m! (m = 0 .. max(max key, out_len)) on the device; partial needs
// thin_hist_chunks() * out_len doubles; out[j-1], j = 1..out_len.
int thin_hist_chunks();
hipError_t launch_thin_hist(const ThinSource *src, int64_t n, const double *lgam, double factor, int64_t out_len,
                            double *partial, double *out, hipStream_t stream);

} // namespace covest
