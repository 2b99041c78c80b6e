/*
 * covest_amd.h -- C ABI of the MI355X (gfx950) likelihood grid-search library.
 *
 * This is the drop-in boundary for ONE path of mhozza/covest: the truncated-
 * Poisson mixture log-likelihood evaluated over a parameter grid and reduced to
 * its arg-min.  Plain C, caller-owned buffers, integer status codes, no torch or
 * C++ types.  Every entry point cites the reference interface it replaces
 * (paths relative to the reference checkout, v0.5.6).  The ctypes binding a
 * maintainer adds on the reference side is shown in INTEGRATION.md.
 *
 * Conventions
 *   - All functions return 0 on success, a negative COVEST_E_* code on failure;
 *     covest_last_error() returns a thread-local message for the last failure.
 *   - Numeric outcomes are in-band IEEE values exactly as the reference returns
 *     them: -inf when a non-zero bin has p_j == 0 (covest/utils.py:32-35), NaN
 *     propagates, parameters outside the model bounds are clamped, never
 *     rejected (covest/models.py:60-69).
 *   - The library copies the histogram to the device at create time and keeps no
 *     pointer of the caller's past any call.
 *   - There is NO CPU fallback: without a usable HIP device every compute entry
 *     point fails with COVEST_E_NO_DEVICE.
 *   - A model handle is immutable after create; calls on one handle are
 *     serialised internally, distinct handles are independent.
 */
#ifndef COVEST_AMD_H
#define COVEST_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define COVEST_ABI_VERSION 1

#define COVEST_MODEL_BASIC 0   /* covest/models.py:17  BasicModel   (coverage, error_rate)          */
#define COVEST_MODEL_REPEATS 1 /* covest/models.py:173 RepeatsModel (coverage, error_rate, q1,q2,q) */
#define COVEST_MAX_PARAMS 5
#define COVEST_MAX_ERROR_CLASSES 64

#define COVEST_OK 0
#define COVEST_E_INVALID (-1)   /* bad argument */
#define COVEST_E_NO_DEVICE (-2) /* no HIP device / HIP runtime error at init */
#define COVEST_E_HIP (-3)       /* HIP runtime error during a call */
#define COVEST_E_NOMEM (-4)
#define COVEST_E_UNSUPPORTED (-5) /* the request is valid but this entry point does not serve it (another one does) */

/* Kernel selection for covest_grid_eval / covest_eval_points (mostly for tests
 * and benchmarks; AUTO picks the fastest kernel valid for the request). */
#define COVEST_KERNEL_AUTO 0
#define COVEST_KERNEL_DIRECT 1   /* one wavefront per grid point, one exp per pmf term        */
#define COVEST_KERNEL_RECUR 2    /* one wavefront per grid point, pmf recurrence along j      */
#define COVEST_KERNEL_FACTORED 3 /* repeats, dense grids: (c,e)-outer / (q1,q2,q)-inner reuse */
/* K-direct with the reference's long-double OVERFLOW reproduced: truncated_poisson (c_src/covest_poissonmodule.c:19-24)
 * forms its whole product before scaling and returns +inf once it passes LDBL_MAX, the likelihood becomes +inf / NaN
 * (covest/models.py:100-107) and optimize_grid would select -(+inf) (covest/grid.py:65-70).  The other kernels
 * return the finite value the formula defines; this one returns what the reference returns, specials included. */
#define COVEST_KERNEL_DIRECT_REF 4

typedef struct covest_model covest_model; /* opaque */
typedef struct covest_grid covest_grid;   /* opaque */

/* Arguments of the reference model constructors:
 *   BasicModel.__init__   covest/models.py:19-31
 *   RepeatsModel.__init__ covest/models.py:175-183
 * `comb[s]` is the reference's self.comb[s] = comb(k, s) * 3**s (models.py:25),
 * passed in so that the caller's scipy decides its rounding, as in the reference.
 * `lo/hi` are self.bounds; NaN stands for Python None. */
typedef struct {
    int32_t kind;   /* COVEST_MODEL_* */
    int32_t k;      /* k-mer size */
    int32_t r;      /* read length */
    int32_t n_err;  /* self.max_error = min(k + 1, max_error): error classes summed, 1..64 */
    const double *comb;    /* [n_err] */
    int64_t n_keys;        /* len(hist) */
    const int32_t *keys;   /* hist keys j in dict order (zero-count keys included) */
    const double *counts;  /* hist values h_j as doubles */
    double tail;           /* self.tail */
    double lo[COVEST_MAX_PARAMS];
    double hi[COVEST_MAX_PARAMS];
    double threshold;      /* RepeatsModel threshold (1e-8) */
    int32_t has_threshold; /* 0: threshold is None */
    int32_t device;        /* HIP device ordinal; -1 = the calling thread's current device */
} covest_model_desc;

/* Library / device probes (no reference counterpart). */
int covest_abi_version(void);
int covest_device_count(void); /* >= 0, or COVEST_E_NO_DEVICE */
const char *covest_last_error(void);

/* Model(k, r, hist, tail, max_error=, max_cov=, ...): covest/covest.py:136-140. */
int covest_model_create(const covest_model_desc *desc, covest_model **out);
void covest_model_destroy(covest_model *m);
int covest_model_param_count(const covest_model *m); /* BasicModel.param_count, models.py:40-42 */
/* number of histogram bins the kernels evaluate: all keys when tail != 0, else
 * only keys with a non-zero count (the tail term of models.py:104 is then 0). */
int64_t covest_model_bins_evaluated(const covest_model *m);

/* RepeatsModel.get_hist_threshold(get_b_o(q1,q2,q), threshold): models.py:185-208.
 * q123 is [n][3] (already clamped to the model bounds); out[n]; hist_max is
 * max(self.hist); has_threshold == 0 stands for threshold None.  Pure host code
 * (no device needed): computed with libm pow, as Python does, so that the
 * integer cut-off cannot differ from the reference by a device ulp. */
int covest_threshold_o(int64_t n, const double *q123, double threshold, int32_t has_threshold,
                       int32_t hist_max, int32_t *out);

/* model.compute_loglikelihood(*params) for a list of points: covest/models.py:100-107,
 * batched like compute_loglikelihood_multi (models.py:109-117).
 * params is [n][param_count] on the HOST; out_ll[n] on the HOST.
 * COVEST_KERNEL_AUTO: basic model -> the recurrence kernel; repeats model -> up to 4096 points go to the
 * factored kernel's list mode (one workgroup per point and key segment, points whose threshold_o exceeds 513 in
 * chunks of 512 copy numbers: the latency path of scipy-driven refinements, ~85 us for one point), longer lists
 * to the direct kernel.  Within either range a point's value does not depend on what else is in the call. */
int covest_eval_points(covest_model *m, int64_t n, const double *params, double *out_ll,
                       int32_t kernel);

/* Documented divergence made visible: the reference forms its pmf product in x87 long double BEFORE
 * scaling it (c_src/covest_poissonmodule.c:19-24), so for large rates against large keys
 * (ln(l^i / i!) > 11356.5 at i = min(j, floor(l))) truncated_poisson returns +inf, the likelihood
 * becomes +inf or NaN, and optimize_grid would select it (covest/grid.py:65-70).  This library
 * returns the finite value the formula defines.  flags[i] = 1 where the reference itself would have
 * overflowed at params[i] (HOST arrays, [n][param_count] and [n]); pure host arithmetic, no device
 * work.  No benchmark configuration contains such a point. */
int covest_reference_overflow(const covest_model *m, int64_t n, const double *params, uint8_t *flags);

/* model.compute_probabilities(*params): models.py:81-98, :211-242.  out_p[n_keys]
 * in key order (host).  clamp != 0 applies fit_to_bounds first, which is how
 * compute_loglikelihood calls it (models.py:101-102); clamp == 0 is the raw
 * method as plot_probs calls it.  Used by --plot and by the parity tests. */
int covest_probabilities(covest_model *m, const double *params, int32_t clamp, double *out_p);

/* Dense grid = itertools.product(*axes), last axis fastest (covest/grid.py:39-43,
 * notebooks/VisualiseLikelihood.ipynb cell 5).  n_axes must equal param_count; a
 * fixed parameter is an axis of length 1 (covest/grid.py:27-28).  The handle
 * evaluates flat indices [flat_begin, flat_end) of the product -- one contiguous
 * block per GPU (multi-GPU block partition); pass 0 and -1 for the whole grid.
 * Axes are copied to the device here.  A grid borrows its model: destroy the grid first. */
int covest_grid_create(covest_model *m, int32_t n_axes, const double *const *axes,
                       const int64_t *axis_len, int64_t flat_begin, int64_t flat_end,
                       covest_grid **out);
void covest_grid_destroy(covest_grid *g);
/* Give an existing handle other axes and/or another block (same meaning of the arguments as
 * covest_grid_create): the iterations of covest/grid.py:56-74 evaluate a new grid each time, and a
 * handle keeps its device memory.  Waits for the handle's last evaluation; what it uploads (axes, threshold_o
 * table, K-factored's plan) is staged in page-locked memory of the handle and copied asynchronously on the default
 * stream -- the next covest_grid_eval queues up behind the copies (on another stream: behind an event). */
int covest_grid_reset(covest_grid *g, int32_t n_axes, const double *const *axes, const int64_t *axis_len,
                      int64_t flat_begin, int64_t flat_end);
int64_t covest_grid_size(const covest_grid *g); /* flat_end - flat_begin */

/* Evaluate LL at every point of the block into a device buffer owned by the
 * handle and reduce it to (min -LL, lowest flat index attaining it) on the
 * device.  Asynchronous on `stream` (a hipStream_t, NULL = default stream).
 * Replaces the Pool.map + scan of covest/grid.py:63-70. */
int covest_grid_eval(covest_grid *g, int32_t kernel, void *stream);

/* Wait for the last covest_grid_eval and return the reduction.  Selection is the
 * scan of covest/grid.py:65-70 with maximize=False starting from +inf: strict <,
 * lowest GLOBAL flat index wins ties, NaN never wins; argmin -1 if nothing is
 * < +inf. */
int covest_grid_argmin(covest_grid *g, double *min_negll, int64_t *argmin_flat);

/* covest_grid_eval and, in the same arg-min launch, the SELECTION SCAN of covest/grid.py:65-70 started from the
 * minimum the caller holds (`start_min`: what optimize_grid's `min_val` is when an iteration begins, :49,67-69):
 *     if sgn * val < min_val: diff += min_val - val; min_val = sgn * val; min_args = args
 * changes its state exactly at the strict running-minimum records below start_min, taken in flat-index order -- a
 * handful of points once a search is under way.  The device lists them (page-locked host memory, the kernel's own
 * stores) and covest_grid_scan hands them over: index[i] (GLOBAL flat index, ascending) and negll[i] = -LL there,
 * so the caller replays the loop over those alone instead of reading every value back (covest_grid_ll_host).
 * *truncated != 0: the list is incomplete (more records than `cap` or than the device keeps -- 120 --, a block beyond
 * 16384 points, or the last evaluation was a plain covest_grid_eval): read the values back instead.  A NaN never
 * passes `<`, as in the reference.  maximize=False only (sgn = 1). */
int covest_grid_eval_scan(covest_grid *g, int32_t kernel, void *stream, double start_min);
int covest_grid_scan(covest_grid *g, int32_t cap, int64_t *index, double *negll, int32_t *n_records, int32_t *truncated);

/* Device pointer of the reduction of the last covest_grid_eval as two doubles in HBM,
 * {min -LL, GLOBAL flat index of the arg-min as a double (-1 if none; flat indices stay
 * below 2^53)}: what the ranks of a multi-GPU search exchange (one all-gather of these 16
 * bytes) without a round trip through the host.  Valid until the next
 * covest_grid_reset (which may move the handle's device arena) or destroy; written by
 * covest_grid_eval on its stream. */
const double *covest_grid_argmin_pair_device(const covest_grid *g);

/* Device pointer of the block's LL values (double[grid_size], valid until the next
 * covest_grid_reset or destroy) and a copy to the host. */
const double *covest_grid_ll_device(const covest_grid *g);
int covest_grid_ll_host(covest_grid *g, double *out_ll);

/* Work accounting of the last covest_grid_eval, for roofline reporting:
 * pmf terms evaluated (bins_evaluated * n_err * sum(T-1)), log evaluations,
 * and the name of the kernel that ran. */
int covest_grid_work(const covest_grid *g, double *pmf_terms, double *flops, const char **kernel);

/* Device-side timing of the likelihood kernel alone (not the arg-min pass), for
 * roofline reporting: with profiling enabled every covest_grid_eval brackets its
 * likelihood launch with hipEvents on the launch stream; covest_grid_kernel_ms
 * waits for them and returns the summed duration and the number of launches
 * since profiling was (re-)enabled. */
int covest_grid_profile(covest_grid *g, int32_t enable);
int covest_grid_kernel_ms(covest_grid *g, double *total_ms, int64_t *launches);

/* ---- k-mer abundance histogram: bin/kmer_hist.py (BASELINE config 5) ----
 * The counter is the `counts` dict of compute_counts (bin/kmer_hist.py:34-41) as an
 * open-addressing hash table in HBM.  k <= 255 (the reference's Python integers have no limit: k <= 31 is the fast
 * path, one 64-bit word per key; 32..63, ..127, ..255 take keys of 2, 4, 8 words).  canonical != 0 counts a k-mer and its reverse
 * complement as one key (jellyfish -C; NOT reference behaviour, the reference is forward-strand). */
typedef struct covest_kmer covest_kmer; /* opaque */

int covest_kmer_create(int32_t k, int32_t canonical, int64_t min_slots, int32_t device, covest_kmer **out);
void covest_kmer_destroy(covest_kmer *c);
/* Grow the table to at least min_slots (power of two), re-inserting what it holds.  The caller
 * keeps the table at most half full: slots >= 2 * (DISTINCT k-mers held + those the next batch can
 * add).  Overflow contract: a covest_kmer_add / covest_kmer_histogram that returns COVEST_E_NOMEM has
 * counted PART of its batch; the counter is then only good for covest_kmer_clear (recount with a larger
 * table).  The overflow state is STICKY until covest_kmer_clear: covest_kmer_reserve waits for everything in
 * flight on the device, returns COVEST_E_NOMEM if an earlier (asynchronous) add overflowed, and otherwise
 * re-inserts into the larger table and reports its own outcome. */
int covest_kmer_reserve(covest_kmer *c, int64_t min_slots);
/* compute_counts(seq, prev_counts=counts, k) for n_reads preprocessed reads (bin/kmer_hist.py:44-54
 * already applied: only a/c/g/t in either case).  bases: the reads back to back; offsets[n_reads+1].
 * A read shorter than k contributes the hash of what there is, an empty read k-mer 0 (:36-37).
 * HOST buffers; the call copies them to the device and waits. */
int covest_kmer_add(covest_kmer *c, const uint8_t *bases, const int64_t *offsets, int64_t n_reads);
/* Same with DEVICE buffers, asynchronous on `stream`: d_offsets may be NULL when every read is
 * read_len bases long. */
int covest_kmer_add_device(covest_kmer *c, const uint8_t *d_bases, const int64_t *d_offsets,
                           int64_t n_reads, int64_t read_len, void *stream);
/* compute_histogram(counts) (bin/kmer_hist.py:57-64): out[i] = number of distinct k-mers seen i
 * times, i = 0 .. max count.  Call with out == NULL to learn needed_len (= max count + 1) and the
 * number of distinct k-mers; fails with COVEST_E_NOMEM-like status if the table overflowed. */
int covest_kmer_histogram(covest_kmer *c, int64_t *out, int64_t out_len, int64_t *needed_len,
                          int64_t *distinct);
/* The WHOLE counting loop of main (bin/kmer_hist.py:77-89: compute_counts over every read) for reads resident in
 * HBM, into an EMPTY counter, by the partitioned path (kmer_bulk.hip): the k-mers are grouped by minimizer into
 * buckets of super-k-mer records and counted bucket by bucket in LDS -- an occurrence costs no scattered memory
 * operation.  The counts are not kept as a dict: afterwards the counter answers covest_kmer_histogram (count-of-counts,
 * distinct keys -- exact) and nothing else, until covest_kmer_clear.  d_offsets NULL: every read is read_len bases;
 * else offsets[n_reads + 1] (ascending, reads back to back; fewer than 2^32 reads); n_bases_total is a hint, the
 * offsets decide.  Blocks until done.
 * COVEST_E_UNSUPPORTED: k outside 19..31, or reads of one length (no offsets) shorter than k; COVEST_E_NOMEM: the
 * buckets do not fit the device, or the sample of the reads misjudged them beyond what the overflow list holds --
 * covest_kmer_clear, then count with covest_kmer_add_device (whatever the counter held before the call is not part
 * of the result either way: the call counts into an emptied counter). */
int covest_kmer_count_reads_device(covest_kmer *c, const uint8_t *d_bases, const int64_t *d_offsets, int64_t n_reads,
                                   int64_t read_len, int64_t n_bases_total, void *stream);
/* How the last covest_kmer_count_reads_device went (the counter still holds its result): out[0] buckets, [1] minimizer
 * length, [2] pass 0 sampled one block of reads in this many, [3] records the buckets had room for, [4] records that
 * found their bucket full, [5] buckets counted by a workgroup instead of a wave, [6] buckets counted through the table
 * in HBM, [7] records pass 1 wrote.  Diagnostics for the caller's log; no reference counterpart. */
int covest_kmer_partition_info(const covest_kmer *c, int64_t out[8]);
/* A cap on what the partitioned path may allocate for its buckets' records (bytes; 0 = no cap but the device's free
 * memory): a caller that shares the card names its budget, and covest_kmer_count_reads_device answers COVEST_E_NOMEM
 * where the buckets would pass it -- before it allocates them. */
int covest_kmer_memory_limit(covest_kmer *c, int64_t max_bytes);
/* ... and how long its steps took on the device, milliseconds (HIP events on the caller's stream): out[0] pass 0
 * (sample of the reads, room per bucket, their places), [1] pass 1 (records to their buckets), [2] pass 2 (the buckets
 * counted in LDS), [3] what was left for the table in HBM. */
int covest_kmer_partition_ms(const covest_kmer *c, double out[4]);
/* Measurement only: the rate at which THIS device retires returning 64-bit atomic adds at pseudo-random places of a
 * `slots`-word array (the 64 lanes of a wave instruction on 64 different lines) -- what bounds pass 1 of the
 * partitioned path (one such add per record) and, per occurrence, the table path.  `ops` adds are timed. */
int covest_kmer_scatter_rate(int32_t device, int64_t slots, int64_t ops, double *ops_per_s);
int64_t covest_kmer_slots(const covest_kmer *c);
/* Forget every count (counts = defaultdict(int) again), keeping the table's size; asynchronous on `stream`.  After
 * covest_kmer_count_reads_device it also frees the buckets' records (gigabytes the handle keeps from call to call). */
int covest_kmer_clear(covest_kmer *c, void *stream);

/* ---- FASTA / FASTQ front-end of the k-mer histogram: bin/kmer_hist.py:44-54 preprocess, :67-74 load_reads ----
 * HOST code (the only HIP calls: a device count and the allocation of page-locked buffers): the mapped file
 * is parsed span by span into batches in the packed layout covest_kmer_add takes.  Format by extension, as the reference: ".fq" / ".fastq" = FASTQ (4-line records), anything else FASTA
 * (a record = a '>' header line and the concatenation of the lines up to the next header; the reference leaves
 * the parsing to Bio.SeqIO).  preprocess is applied on the way: lower case; 'n' dropped (n_strategy 0, IGNORE),
 * replaced by 'a' (1, SINGLE) or by a random base (2, RANDOM: a hash of `seed` and of the N's position in the
 * file -- the reference draws from Python's unseeded random, which nothing can reproduce).  Any other letter fails with COVEST_E_INVALID, where
 * the reference's single_hash raises KeyError (:15).  An empty record is a read (it counts k-mer 0, :36-37). */
typedef struct covest_reads covest_reads; /* opaque */
int covest_reads_open(const char *path, int32_t n_strategy, uint64_t seed, covest_reads **out);
void covest_reads_close(covest_reads *r);
/* The next batch: whole reads, about max_bases bases of them (the span of the file that holds that many, up to the
 * next record boundary; one read at least), parsed by several threads (COVEST_READER_THREADS, default: the
 * machine's, 16 at most).  Two batch buffers alternate: *bases / *offsets[*n_reads + 1] stay valid until the call
 * AFTER the next one on `r`, so batch i + 1 can be parsed while batch i is being counted.  The buffers are
 * page-locked when the process has a HIP device (COVEST_READER_PINNED=0: never).  *n_reads == 0: end of file. */
int covest_reads_next(covest_reads *r, int64_t max_bases, const uint8_t **bases, const int64_t **offsets,
                      int64_t *n_reads);
/* file bytes consumed so far */
int64_t covest_reads_bytes(const covest_reads *r);

/* ---- histogram down-sampling: covest/histogram.py:47-70 sample_histogram ----
 * Expected counts of the histogram after keeping every read with probability 1/factor, BEFORE the
 * reference's randomised rounding (:71-74, host side): out[j-1] = sum_i counts_i * pmf_i(j) for
 * j = 1 .. out_len, with pmf_i = binomial(i, 1/factor) for i < 100 and Poisson(i/factor) for i >= 100.
 * keys[n] >= 1 are the (already trimmed) source counts i, counts[n] their multiplicities.  HOST buffers.
 * factor must be > 1.  (Where i/factor > 200 the reference's poisson_dist is itself wrong -- see
 * DESIGN.md -- and this returns the correct Poisson pmf.) */
int covest_thin_histogram(int32_t device, int64_t n, const int32_t *keys, const double *counts, double factor,
                          int64_t out_len, double *out);
/* The same, launched `repeats` times with inputs resident in HBM; *kernel_ms = mean device time of one
 * launch pair (hipEvents on the launch stream).  For benchmarks. */
int covest_thin_histogram_timed(int32_t device, int64_t n, const int32_t *keys, const double *counts, double factor,
                                int64_t out_len, double *out, int32_t repeats, double *kernel_ms);

/* PROFILING AID: with the environment variable COVEST_FACTORED_DIAG set at
 * covest_grid_create, the factored kernel accumulates s_memtime stamps per wave
 * ([workgroup][wave][8] int64: build, contract, log, barrier cycles); this copies
 * up to n of them to the host and returns how many exist.  Not part of the
 * reference's interface. */
int64_t covest_grid_diag(covest_grid *g, int64_t *out, int64_t n);

#ifdef __cplusplus
}
#endif
#endif /* COVEST_AMD_H */
