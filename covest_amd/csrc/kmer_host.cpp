// kmer_host.cpp -- the k-mer counter of the C ABI (include/covest_amd.h): covest_kmer_* over kmer_count.hip,
// kmer_wide.hip and kmer_bulk.hip.
#include "host.h"

using namespace covest;

namespace {

int kmer_alloc_table(covest_kmer *c, int64_t min_slots, KmerTable &t, DevBuf &slots)
{
    int lg = 10;
    while (((int64_t)1 << lg) < min_slots && lg < 40)
        ++lg;
    const size_t n = (size_t)1 << lg;
    HIP_TRY(slots.reserve(n * sizeof(KmerSlot)));
    t.slots = slots.as<KmerSlot>();
    t.mask = n - 1;
    t.log2_slots = lg;
    t.k = c->k;
    HIP_TRY(launch_kmer_fill_empty(t, nullptr));
    return COVEST_OK;
}

int kmer_alloc_wide(covest_kmer *c, int64_t min_slots, KmerWideTable &t, DevBuf &slots)
{
    int lg = 10;
    while (((int64_t)1 << lg) < min_slots && lg < 38)
        ++lg;
    t.w = c->wide;
    t.stride = 2 * c->wide; // 1 + w words, rounded up to a power of two
    t.k = c->k;
    t.log2_slots = lg;
    t.mask = ((unsigned long long)1 << lg) - 1;
    HIP_TRY(slots.reserve(((size_t)1 << lg) * (size_t)t.stride * sizeof(unsigned long long)));
    t.words = slots.as<unsigned long long>();
    HIP_TRY(launch_kmer_wide_clear(t, nullptr));
    return COVEST_OK;
}

int kmer_check_overflow(covest_kmer *c)
{
    int flag = 0;
    HIP_TRY(hipMemcpy(&flag, c->flag.ptr, sizeof(int), hipMemcpyDeviceToHost));
    if (flag)
        return fail(COVEST_E_NOMEM, "k-mer table overflow: call covest_kmer_reserve with more slots");
    return COVEST_OK;
}

} // namespace

extern "C" {

int covest_kmer_create(int32_t k, int32_t canonical, int64_t min_slots, int32_t device, covest_kmer **out)
{
    if (!out)
        return fail(COVEST_E_INVALID, "covest_kmer_create: null argument");
    *out = nullptr;
    if (k < 1 || k > 255)
        return fail(COVEST_E_INVALID, "covest_kmer_create: k must be in 1..255 (keys of up to eight 64-bit words)");
    {
        const int drc = resolve_device(device, "covest_kmer_create", &device);
        if (drc != COVEST_OK)
            return drc;
    }
    covest_kmer *c = new (std::nothrow) covest_kmer();
    if (!c)
        return fail(COVEST_E_NOMEM, "covest_kmer_create: out of host memory");
    c->device = device;
    c->k = k;
    c->canonical = canonical != 0;
    c->wide = k <= 31 ? 0 : k <= 63 ? 2 : k <= 127 ? 4 : 8;
    DeviceGuard dev_guard(device);
    hipError_t e = hipSuccess;
    int rc = dev_guard.status();
    if (rc == COVEST_OK)
        rc = c->wide ? kmer_alloc_wide(c, min_slots, c->wtable, c->slots) : kmer_alloc_table(c, min_slots, c->table, c->slots);
    if (rc == COVEST_OK) {
        e = c->flag.reserve(sizeof(int));
        if (e == hipSuccess)
            e = hipMemset(c->flag.ptr, 0, sizeof(int));
        if (e == hipSuccess)
            e = c->stats.reserve(2 * sizeof(unsigned long long));
        if (e != hipSuccess)
            rc = fail_hip(e, "covest_kmer_create: allocation");
    }
    if (rc != COVEST_OK) {
        covest_kmer_destroy(c);
        return rc;
    }
    *out = c;
    return COVEST_OK;
}

void covest_kmer_destroy(covest_kmer *c)
{
    if (!c)
        return;
    DeviceGuard dev_guard(c->device);
    for (hipEvent_t &e : c->bulk_ev)
        if (e) {
            (void)hipEventDestroy(e);
            e = nullptr;
        }
    (void)hipDeviceSynchronize();
    delete c; // (its buffers go with it: host.h DevBuf)
}

int64_t covest_kmer_slots(const covest_kmer *c)
{
    return c ? (int64_t)((c->wide ? c->wtable.mask : c->table.mask) + 1) : COVEST_E_INVALID;
}

int covest_kmer_clear(covest_kmer *c, void *stream)
{
    if (!c)
        return fail(COVEST_E_INVALID, "covest_kmer_clear: null counter");
    std::lock_guard<std::mutex> guard(c->lock);
    DeviceGuard dev_guard(c->device);
    if (dev_guard.status() != COVEST_OK)
        return dev_guard.status();
    if (c->wide)
        HIP_TRY(launch_kmer_wide_clear(c->wtable, static_cast<hipStream_t>(stream)));
    else
        HIP_TRY(launch_kmer_fill_empty(c->table, static_cast<hipStream_t>(stream)));
    HIP_TRY(hipMemsetAsync(c->flag.ptr, 0, sizeof(int), static_cast<hipStream_t>(stream)));
    // what the partitioned path kept for its next call -- the buckets' records are gigabytes -- goes with the counts,
    // whether its last call succeeded or not (a call that failed after its reserve left `bulk` false and the records
    // allocated: the table path the caller falls back to needs that memory).  covest_kmer_count_reads_device has
    // returned, and hipFree waits for the device: nothing of it is in flight
    if (c->bulk_recs.ptr || c->bulk_ovf.ptr)
        (void)hipDeviceSynchronize();
    c->bulk_recs.release();
    c->bulk_ovf.release();
    c->bulk = false;
    return COVEST_OK;
}

int covest_kmer_memory_limit(covest_kmer *c, int64_t max_bytes)
{
    if (!c || max_bytes < 0)
        return fail(COVEST_E_INVALID, "covest_kmer_memory_limit: bad argument");
    std::lock_guard<std::mutex> guard(c->lock);
    c->bulk_mem_limit = max_bytes;
    return COVEST_OK;
}

int covest_kmer_reserve(covest_kmer *c, int64_t min_slots)
{
    if (!c)
        return fail(COVEST_E_INVALID, "covest_kmer_reserve: null counter");
    std::lock_guard<std::mutex> guard(c->lock);
    if ((int64_t)((c->wide ? c->wtable.mask : c->table.mask) + 1) >= min_slots)
        return COVEST_OK;
    DeviceGuard dev_guard(c->device);
    if (dev_guard.status() != COVEST_OK)
        return dev_guard.status();
    // An overflow of an earlier covest_kmer_add_device (asynchronous: it never looks at the flag itself) is STICKY
    // until covest_kmer_clear: its batch is partly counted, and a rehash of a table with k-mers missing must not make
    // the next covest_kmer_histogram look clean.  So: everything in flight on this device first (the adds may run
    // on a caller's non-blocking stream, the rehash runs on the null stream), then the flag.
    HIP_TRY(hipDeviceSynchronize());
    {
        const int rc = kmer_check_overflow(c);
        if (rc != COVEST_OK)
            return rc;
    }
    KmerTable bigger{};
    KmerWideTable wbigger{};
    DevBuf slots;
    int rc = c->wide ? kmer_alloc_wide(c, min_slots, wbigger, slots) : kmer_alloc_table(c, min_slots, bigger, slots);
    hipError_t e = hipSuccess;
    if (rc == COVEST_OK) { // (the flag is known to be clean here: whatever it holds afterwards is the rehash's)
        e = c->wide ? launch_kmer_wide_rehash(c->wtable, wbigger, c->flag.as<int>(), nullptr)
                    : launch_kmer_rehash(c->table, bigger, c->flag.as<int>(), nullptr);
        if (e == hipSuccess)
            e = hipDeviceSynchronize();
        if (e != hipSuccess)
            rc = fail_hip(e, "covest_kmer_reserve: rehash");
    }
    if (rc != COVEST_OK)
        return rc; // (the new table goes with `slots`)
    c->slots = std::move(slots);
    c->table = bigger;
    c->wtable = wbigger;
    return kmer_check_overflow(c);
}

int covest_kmer_add_device(covest_kmer *c, const uint8_t *d_bases, const int64_t *d_offsets,
                           int64_t n_reads, int64_t read_len, void *stream)
{
    if (!c || n_reads < 0 || (n_reads > 0 && !d_bases) || (!d_offsets && read_len < 0))
        return fail(COVEST_E_INVALID, "covest_kmer_add_device: bad argument");
    std::lock_guard<std::mutex> guard(c->lock);
    if (c->bulk)
        return fail(COVEST_E_INVALID, "covest_kmer_add_device: the counter holds a covest_kmer_count_reads_device result "
                                      "(its keys are not in the table); covest_kmer_clear first");
    DeviceGuard dev_guard(c->device);
    if (dev_guard.status() != COVEST_OK)
        return dev_guard.status();
    if (c->wide)
        HIP_TRY(launch_kmer_wide_count(d_bases, d_offsets, n_reads, read_len, c->canonical, c->wtable, c->flag.as<int>(),
                                       static_cast<hipStream_t>(stream)));
    else
        HIP_TRY(launch_kmer_count(d_bases, d_offsets, n_reads, read_len, c->k, c->canonical, c->table,
                                  c->flag.as<int>(), static_cast<hipStream_t>(stream)));
    return COVEST_OK;
}

int covest_kmer_add(covest_kmer *c, const uint8_t *bases, const int64_t *offsets, int64_t n_reads)
{
    if (!c || n_reads < 0 || (n_reads > 0 && !offsets))
        return fail(COVEST_E_INVALID, "covest_kmer_add: bad argument");
    if (n_reads == 0)
        return COVEST_OK;
    const int64_t n_bytes = offsets[n_reads] - offsets[0];
    if (n_bytes < 0 || (n_bytes > 0 && !bases))
        return fail(COVEST_E_INVALID, "covest_kmer_add: bad offsets");
    DeviceGuard dev_guard(c->device);
    if (dev_guard.status() != COVEST_OK)
        return dev_guard.status();
    {
        std::lock_guard<std::mutex> guard(c->lock);
        HIP_TRY(c->ws_bases.reserve((size_t)(n_bytes > 0 ? n_bytes : 1)));
        HIP_TRY(c->ws_offsets.reserve((size_t)(n_reads + 1) * sizeof(int64_t)));
        if (n_bytes > 0)
            HIP_TRY(hipMemcpy(c->ws_bases.ptr, bases + offsets[0], (size_t)n_bytes, hipMemcpyHostToDevice));
        std::vector<int64_t> rel((size_t)n_reads + 1);
        for (int64_t i = 0; i <= n_reads; ++i)
            rel[(size_t)i] = offsets[i] - offsets[0];
        HIP_TRY(hipMemcpy(c->ws_offsets.ptr, rel.data(), rel.size() * sizeof(int64_t), hipMemcpyHostToDevice));
    }
    int rc = covest_kmer_add_device(c, c->ws_bases.as<uint8_t>(), c->ws_offsets.as<int64_t>(), n_reads, 0, nullptr);
    if (rc != COVEST_OK)
        return rc;
    HIP_TRY(hipDeviceSynchronize());
    return kmer_check_overflow(c);
}

int covest_kmer_histogram(covest_kmer *c, int64_t *out, int64_t out_len, int64_t *needed_len,
                          int64_t *distinct)
{
    if (!c)
        return fail(COVEST_E_INVALID, "covest_kmer_histogram: null counter");
    std::lock_guard<std::mutex> guard(c->lock);
    DeviceGuard dev_guard(c->device);
    if (dev_guard.status() != COVEST_OK)
        return dev_guard.status();
    HIP_TRY(hipDeviceSynchronize());
    int rc = kmer_check_overflow(c);
    if (rc != COVEST_OK)
        return rc;
    unsigned long long stats[2] = {0, 0};
    const bool table_in_use = !c->bulk || c->bulk_table_used; // (a partitioned count may leave nothing in the table)
    if (table_in_use) {
        HIP_TRY(hipMemset(c->stats.ptr, 0, sizeof(stats)));
        if (c->wide)
            HIP_TRY(launch_kmer_wide_stats(c->wtable, c->stats.as<unsigned long long>(), nullptr));
        else
            HIP_TRY(launch_kmer_stats(c->table, c->stats.as<unsigned long long>(), nullptr));
        HIP_TRY(hipMemcpy(stats, c->stats.ptr, sizeof(stats), hipMemcpyDeviceToHost));
    }
    // (after covest_kmer_count_reads_device the table holds only what the partitioned path handed back; the rest of
    // the keys were counted in LDS, and what is left of them is their count-of-counts)
    std::vector<unsigned long long> big;
    if (c->bulk) {
        stats[0] = std::max(stats[0], c->bulk_stats[0]);
        stats[1] += c->bulk_stats[1];
        if (c->bulk_stats[2] > 0) {
            big.resize((size_t)c->bulk_stats[2]);
            HIP_TRY(hipMemcpy(big.data(), c->bulk_big.ptr, big.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        }
    }
    const int64_t need = (int64_t)stats[0] + 1; // index 0 .. max count (bin/kmer_hist.py:64)
    if (needed_len)
        *needed_len = need;
    if (distinct)
        *distinct = (int64_t)stats[1];
    if (!out)
        return COVEST_OK;
    if (out_len < need)
        return fail(COVEST_E_INVALID, "covest_kmer_histogram: output shorter than max count + 1");
    HIP_TRY(c->hist.reserve((size_t)need * sizeof(unsigned long long)));
    HIP_TRY(hipMemset(c->hist.ptr, 0, (size_t)need * sizeof(unsigned long long)));
    if (!table_in_use)
        ;
    else if (c->wide)
        HIP_TRY(launch_kmer_wide_histogram(c->wtable, c->hist.as<unsigned long long>(), (unsigned long long)need, nullptr));
    else
        HIP_TRY(launch_kmer_histogram(c->table, c->hist.as<unsigned long long>(), (unsigned long long)need, nullptr));
    HIP_TRY(hipMemcpy(out, c->hist.ptr, (size_t)need * sizeof(int64_t), hipMemcpyDeviceToHost));
    if (c->bulk) {
        const size_t n_dense = (size_t)std::min<unsigned long long>((unsigned long long)need, kBulkHistLen);
        std::vector<unsigned long long> dense(n_dense);
        HIP_TRY(hipMemcpy(dense.data(), c->bulk_hist.ptr, n_dense * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < n_dense; ++i)
            out[i] += (int64_t)dense[i];
        for (unsigned long long v : big)
            if ((int64_t)v < need)
                out[v] += 1;
    }
    return COVEST_OK;
}

// The whole counting loop of bin/kmer_hist.py:77-89 for reads resident in HBM, by the partitioned path
// (kmer_bulk.hip).  See include/covest_amd.h.
int covest_kmer_count_reads_device(covest_kmer *c, const uint8_t *d_bases, const int64_t *d_offsets, int64_t n_reads,
                                   int64_t read_len, int64_t n_bases_total, void *stream)
{
    if (!c || n_reads < 0 || (n_reads > 0 && !d_bases) || (!d_offsets && read_len < 0))
        return fail(COVEST_E_INVALID, "covest_kmer_count_reads_device: bad argument");
    if (c->wide || c->k < 19 || c->k > 31)
        return fail(COVEST_E_UNSUPPORTED, "covest_kmer_count_reads_device: the partitioned path takes k = 19 .. 31 "
                                          "(use covest_kmer_add_device)");
    if (!d_offsets && (read_len < c->k || read_len >= ((int64_t)1 << 30)))
        return fail(COVEST_E_UNSUPPORTED, "covest_kmer_count_reads_device: reads shorter than k, or of 2^30 bases and more "
                                          "(use covest_kmer_add_device)");
    std::lock_guard<std::mutex> guard(c->lock);
    DeviceGuard dev_guard(c->device);
    if (dev_guard.status() != COVEST_OK)
        return dev_guard.status();
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int k = c->k;
    // Reads that come with offsets but are all of one length (a sequencer's usually are) take the path of reads of one
    // length: its threads need not look up which read their byte belongs to.
    int64_t ragged_base0 = 0, ragged_total = 0;
    if (d_offsets && n_reads > 0) {
        HIP_TRY(c->bulk_ctl.reserve((16 + kOvfShards * kOvfStride) * sizeof(unsigned long long)));
        unsigned long long *flag = c->bulk_ctl.as<unsigned long long>();
        const unsigned long long one = 1;
        int64_t two[2] = {0, 0};
        HIP_TRY(hipMemcpyAsync(flag, &one, sizeof one, hipMemcpyHostToDevice, st));
        HIP_TRY(launch_kmer_one_length(d_offsets, n_reads, flag, st));
        unsigned long long same = 0;
        int64_t last = 0;
        HIP_TRY(hipMemcpyAsync(&same, flag, sizeof same, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(two, d_offsets, sizeof two, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(&last, d_offsets + n_reads, sizeof last, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        const int64_t len0 = two[1] - two[0];
        ragged_base0 = two[0];
        ragged_total = last - two[0];
        if (ragged_total < 0 || n_reads >= ((int64_t)1 << 32))
            return fail(COVEST_E_INVALID, "covest_kmer_count_reads_device: offsets do not ascend, or 2^32 reads and more");
        n_bases_total = ragged_total;
        if (same && len0 >= k && len0 < ((int64_t)1 << 30)) {
            d_bases += two[0];
            d_offsets = nullptr;
            read_len = len0;
        }
    }
    // windows (an upper bound for reads of different lengths: every base starts at most one)
    const double windows = d_offsets ? (double)std::max<int64_t>(n_bases_total, n_reads) : (double)n_reads * (double)(read_len - k + 1);
    KmerBulk p{};
    p.k = k;
    p.m = std::min(k - 8, 13);
    p.canonical = c->canonical;
    p.max_run = 32 - k + 1;
    // 1000-2000 windows per bucket, at least 2^10 buckets, at most an eighth of the minimizers there are.  Measured
    // (diagnostic build, COVEST_KMER_LG): 1 Gbp 2^22 / 2^21 / 2^20 / 2^19 buckets 20.3 / 18.2 / 17.8 / 16.5 ms, 10 Gbp
    // 2^25 / 2^24 / 2^23 / 2^22 / 2^21 205 / 174 / 141-155 / 143-145 / 146 ms: fewer, fuller buckets keep the sectors
    // that pass 1 writes into within the caches' reach and the sample of pass 0 thin; a bucket of 2000 windows still
    // fits a workgroup's LDS table when every one of them is a different key.
    int lg = 10;
    while (lg < 2 * p.m - 3 && (double)((int64_t)1 << lg) * 2048.0 < windows)
        ++lg;
#ifdef COVEST_DIAG // diagnostic builds only: the shipped library has no knobs
    if (const char *e = std::getenv("COVEST_KMER_M"))
        p.m = std::max(8, std::min(std::atoi(e), std::min(k - 1, 15)));
    if (const char *e = std::getenv("COVEST_KMER_LG"))
        lg = std::max(10, std::min(std::atoi(e), 26));
#endif
    p.w = k - p.m + 1;
    p.log2_buckets = lg;
    const size_t n_buckets = (size_t)1 << lg;
    // pass 0 looks at everything when that is little, else at one block of tiles (one read) in 2 .. 16: as thin a
    // sample as leaves the average bucket six sampled records (a record per ~5 windows) -- the room is the estimate
    // plus three of its standard deviations, and below that the estimate is mostly deviation (1 Gbp with one block in
    // 16: 2 % of the buckets overflowed their room and went through the table in HBM)
    {
        const double per_bucket = windows / 5.0 / (double)n_buckets;
        int thin = 1;
        while (thin < 16 && (double)(2 * thin) * 6.0 <= per_bucket)
            thin *= 2;
        const double bytes = d_offsets ? (double)ragged_total : (double)n_reads * (double)read_len;
        const bool large = bytes / (double)kmer_bulk_block_bytes(p) >= 4096.0;
        p.sample = large ? thin : 1;
    }
#ifdef COVEST_DIAG
    if (const char *e = std::getenv("COVEST_KMER_SAMPLE"))
        p.sample = std::max(1, std::atoi(e));
#endif
    HIP_TRY(c->bulk_sampled.reserve(n_buckets * sizeof(unsigned)));
    HIP_TRY(c->bulk_cursor.reserve(n_buckets * sizeof(ulonglong2)));
    HIP_TRY(c->bulk_fill.reserve(n_buckets * sizeof(unsigned long long)));
    HIP_TRY(c->bulk_later.reserve((2 * n_buckets + 8) * sizeof(unsigned)));
    HIP_TRY(c->bulk_partial.reserve((n_buckets / 1024 + 1) * sizeof(unsigned long long)));
    HIP_TRY(c->bulk_ctl.reserve((16 + kOvfShards * kOvfStride) * sizeof(unsigned long long)));
    HIP_TRY(c->bulk_hist.reserve((size_t)kBulkHistLen * sizeof(unsigned long long)));
    HIP_TRY(c->bulk_big.reserve((size_t)kBulkBigCap * sizeof(unsigned long long)));
    p.sampled = c->bulk_sampled.as<unsigned>();
    p.ctl = c->bulk_cursor.as<ulonglong2>();
    p.fill = c->bulk_fill.as<KmerBulk::fill_t>();
    // [2] room for records in all, [4..7] stats, [16 ..] the overflow list's counters (one per 128-byte line)
    unsigned long long *ctl = c->bulk_ctl.as<unsigned long long>();
    p.ovf_count = ctl + 16;
    c->bulk = false;
    for (hipEvent_t &e : c->bulk_ev)
        if (!e)
            HIP_TRY(hipEventCreate(&e));
    HIP_TRY(hipEventRecord(c->bulk_ev[0], st));
    // pass 0: room per bucket from the sample, the buckets' places
    HIP_TRY(hipMemsetAsync(p.sampled, 0, n_buckets * sizeof(unsigned), st));
    HIP_TRY(hipMemsetAsync(ctl, 0, (16 + kOvfShards * kOvfStride) * sizeof(unsigned long long), st));
    unsigned *first_read = nullptr;
    if (d_offsets && n_reads > 0) { // reads of different lengths: the read of every tile's first byte (kmer_bulk.hip)
        HIP_TRY(c->bulk_tile_reads.reserve((size_t)kmer_bulk_ragged_tiles(p, ragged_total) * sizeof(unsigned)));
        first_read = c->bulk_tile_reads.as<unsigned>();
    }
    HIP_TRY(launch_kmer_scatter(d_bases, d_offsets, n_reads, read_len, ragged_base0, ragged_total, first_read, p, true, st));
    HIP_TRY(launch_kmer_place_buckets(p, c->bulk_partial.as<unsigned long long>(), ctl + 2, st));
    HIP_TRY(hipEventRecord(c->bulk_ev[1], st));
    unsigned long long room = 0;
    HIP_TRY(hipMemcpyAsync(&room, ctl + 2, sizeof(room), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    p.overflow_cap = std::max<unsigned long long>(4096ull, room / 8ull) / kOvfShards; // (per part of the list)
    {
        size_t free_b = 0, total_b = 0;
        HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        const size_t have = c->bulk_recs.cap + c->bulk_ovf.cap;
        const double want = ((double)room + (double)p.overflow_cap * kOvfShards) * 16.0;
        if (want > 0.85 * (double)(free_b + have))
            return fail(COVEST_E_NOMEM, "covest_kmer_count_reads_device: the buckets do not fit the free device memory");
        if (c->bulk_mem_limit > 0 && want > (double)c->bulk_mem_limit)
            return fail(COVEST_E_NOMEM, "covest_kmer_count_reads_device: the buckets do not fit the caller's limit "
                                        "(covest_kmer_memory_limit)");
    }
    HIP_TRY(c->bulk_recs.reserve(std::max<size_t>((size_t)room, 1) * sizeof(ulonglong2)));
    HIP_TRY(c->bulk_ovf.reserve((size_t)p.overflow_cap * kOvfShards * sizeof(ulonglong2)));
    p.recs = c->bulk_recs.as<ulonglong2>();
    p.overflow = c->bulk_ovf.as<ulonglong2>();
    // [0] buckets left to a workgroup, [2..3] buckets left to the table and (64-bit) their k-mers; the lists behind
    unsigned *later = c->bulk_later.as<unsigned>();
    unsigned long long *to_table = reinterpret_cast<unsigned long long *>(later + 2);
    unsigned *later_list = later + 8, *to_table_list = later + 8 + n_buckets;
    HIP_TRY(hipMemsetAsync(later, 0, 8 * sizeof(unsigned), st));
    HIP_TRY(hipMemsetAsync(p.fill, 0, n_buckets * sizeof(KmerBulk::fill_t), st));
    HIP_TRY(hipMemsetAsync(c->bulk_hist.ptr, 0, (size_t)kBulkHistLen * sizeof(unsigned long long), st));
    // pass 1, pass 2
    int n_cu = 256;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, c->device) == hipSuccess && prop.multiProcessorCount > 0)
            n_cu = prop.multiProcessorCount;
    }
    HIP_TRY(launch_kmer_scatter(d_bases, d_offsets, n_reads, read_len, ragged_base0, ragged_total, first_read, p, false, st));
    HIP_TRY(hipEventRecord(c->bulk_ev[2], st));
    HIP_TRY(launch_kmer_bucket_count(p, c->bulk_hist.as<unsigned long long>(), kBulkHistLen, ctl + 4,
                                     c->bulk_big.as<unsigned long long>(), kBulkBigCap, later, later_list, to_table, to_table_list,
                                     /*small_buckets=*/(double)room <= 128.0 * (double)n_buckets, n_cu, st));
    HIP_TRY(hipEventRecord(c->bulk_ev[3], st));
    unsigned long long n_overflowed = 0, listed[2] = {0, 0};
    std::vector<unsigned long long> parts((size_t)kOvfShards * kOvfStride);
    HIP_TRY(hipMemcpyAsync(parts.data(), p.ovf_count, parts.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(c->bulk_stats, ctl + 4, sizeof(c->bulk_stats), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(&c->bulk_later_n, later, sizeof(unsigned), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(listed, to_table, sizeof(listed), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    bool part_full = false;
    for (int i = 0; i < kOvfShards; ++i) {
        n_overflowed += parts[(size_t)i * kOvfStride];
        part_full = part_full || parts[(size_t)i * kOvfStride] > p.overflow_cap;
    }
    if (part_full)
        return fail(COVEST_E_NOMEM, "covest_kmer_count_reads_device: the overflow list is full (the sample of the reads "
                                    "misjudged the buckets); use covest_kmer_add_device");
    if (c->bulk_stats[2] > kBulkBigCap)
        return fail(COVEST_E_NOMEM, "covest_kmer_count_reads_device: more than 4096 keys with counts beyond 2^20");
    // what no LDS table could hold -- the buckets that overflowed their room (all their records: a key is counted in one
    // place), those with too many distinct keys -- goes to the table in HBM, sized now that the need is known
    c->bulk_table_used = n_overflowed > 0 || listed[0] > 0;
    if (c->bulk_table_used) {
        const double want = 2.0 * ((double)listed[1] + (double)n_overflowed * (double)p.max_run) + 1024.0;
        int tlg = 10;
        while (tlg < 40 && (double)((int64_t)1 << tlg) < want)
            ++tlg;
        if ((int64_t)(c->table.mask + 1) < ((int64_t)1 << tlg)) { // (nothing to keep: the counter was to be emptied)
            KmerTable bigger{};
            DevBuf slots;
            const int rc = kmer_alloc_table(c, (int64_t)1 << tlg, bigger, slots);
            if (rc != COVEST_OK)
                return rc;
            c->slots = std::move(slots);
            c->table = bigger;
        }
        HIP_TRY(launch_kmer_fill_empty(c->table, st));
        HIP_TRY(hipMemsetAsync(c->flag.ptr, 0, sizeof(int), st));
        HIP_TRY(launch_kmer_to_table(p, n_overflowed > 0, c->table, c->flag.as<int>(), to_table, to_table_list, st));
        HIP_TRY(hipStreamSynchronize(st));
        const int frc = kmer_check_overflow(c);
        if (frc != COVEST_OK)
            return frc;
    }
    HIP_TRY(hipEventRecord(c->bulk_ev[4], st));
    HIP_TRY(hipEventSynchronize(c->bulk_ev[4]));
    for (int i = 0; i < 4; ++i)
        HIP_TRY(hipEventElapsedTime(&c->bulk_ms[i], c->bulk_ev[i], c->bulk_ev[i + 1]));
    c->bulk_info[0] = (int64_t)n_buckets;
    c->bulk_info[1] = p.m;
    c->bulk_info[2] = p.sample;
    c->bulk_info[3] = (int64_t)room;
    c->bulk_info[4] = (int64_t)n_overflowed;
    c->bulk_to_table_n = listed[0];
    c->bulk = true;
    return COVEST_OK;
}

int covest_kmer_partition_info(const covest_kmer *c, int64_t out[8])
{
    if (!c || !out)
        return fail(COVEST_E_INVALID, "covest_kmer_partition_info: bad argument");
    if (!c->bulk)
        return fail(COVEST_E_INVALID, "covest_kmer_partition_info: the counter holds no covest_kmer_count_reads_device result");
    for (int i = 0; i < 5; ++i)
        out[i] = c->bulk_info[i];
    out[5] = (int64_t)c->bulk_later_n;
    out[6] = (int64_t)c->bulk_to_table_n;
    out[7] = (int64_t)c->bulk_stats[3];
    return COVEST_OK;
}

int covest_kmer_partition_ms(const covest_kmer *c, double out[4])
{
    if (!c || !out)
        return fail(COVEST_E_INVALID, "covest_kmer_partition_ms: bad argument");
    if (!c->bulk)
        return fail(COVEST_E_INVALID, "covest_kmer_partition_ms: the counter holds no covest_kmer_count_reads_device result");
    for (int i = 0; i < 4; ++i)
        out[i] = (double)c->bulk_ms[i];
    return COVEST_OK;
}

int covest_kmer_scatter_rate(int32_t device, int64_t slots, int64_t ops, double *ops_per_s)
{
    if (slots < 1 || ops < 1 || !ops_per_s)
        return fail(COVEST_E_INVALID, "covest_kmer_scatter_rate: bad argument");
    {
        const int drc = resolve_device(device, "covest_kmer_scatter_rate", &device);
        if (drc != COVEST_OK)
            return drc;
    }
    DeviceGuard dev_guard(device);
    if (dev_guard.status() != COVEST_OK)
        return dev_guard.status();
    DevBuf words;
    HIP_TRY(words.reserve(((size_t)slots + 1) * sizeof(unsigned long long)));
    hipEvent_t a = nullptr, b = nullptr;
    hipError_t e = hipMemset(words.ptr, 0, ((size_t)slots + 1) * sizeof(unsigned long long));
    if (e == hipSuccess)
        e = hipEventCreate(&a);
    if (e == hipSuccess)
        e = hipEventCreate(&b);
    unsigned long long *w = words.as<unsigned long long>();
    if (e == hipSuccess) // (once untimed: the pages are touched, the clocks are up)
        e = launch_kmer_scatter_rate(w, (unsigned long long)slots, std::min<int64_t>(ops, 1 << 24), w + slots, nullptr);
    if (e == hipSuccess)
        e = hipEventRecord(a, nullptr);
    if (e == hipSuccess)
        e = launch_kmer_scatter_rate(w, (unsigned long long)slots, ops, w + slots, nullptr);
    if (e == hipSuccess)
        e = hipEventRecord(b, nullptr);
    if (e == hipSuccess)
        e = hipEventSynchronize(b);
    float ms = 0.0f;
    if (e == hipSuccess)
        e = hipEventElapsedTime(&ms, a, b);
    if (a)
        (void)hipEventDestroy(a);
    if (b)
        (void)hipEventDestroy(b);
    if (e != hipSuccess)
        return fail_hip(e, "covest_kmer_scatter_rate");
    const double done = (double)(((ops + 63) / 64) * 64);
    *ops_per_s = done / ((double)ms * 1e-3);
    return COVEST_OK;
}


} // extern "C"
