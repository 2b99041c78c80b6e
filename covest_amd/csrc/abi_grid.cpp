// abi_grid.cpp -- the grid handle of the C ABI (include/covest_amd.h): covest_grid_*.
#include "host.h"

using namespace covest;

namespace covest {

// Where a grid handle stages an upload, and how the copy is issued (host.h covest_grid::stage).
int grid_stage_begin(covest_grid *g, size_t bytes, StageSlot &slot)
{
    if (!g->async_uploads) { // covest_grid_create: the process's staging buffer, a blocking copy
        SharedStage &ss = shared_stage();
        slot.shared_hold = std::unique_lock<std::mutex>(ss.mu);
        HIP_TRY(ss.buf.reserve(bytes));
        slot.ptr = ss.buf.as<char>();
        return COVEST_OK;
    }
    const size_t at = (g->stage_off + 63) / 64 * 64;
    if (at + bytes > g->stage.cap) {
        // (the block is outgrown: what it holds may still be read -- copies in flight, a small grid's axes read in
        // place -- so it is set aside until the next reset, and a larger one takes over)
        if (g->stage.ptr)
            g->stage_retired.push_back(std::move(g->stage));
        HIP_TRY(g->stage.reserve(std::max<size_t>(2 * (at + bytes), 1 << 18)));
        slot.ptr = g->stage.as<char>();
        g->stage_off = bytes;
        return COVEST_OK;
    }
    slot.ptr = g->stage.as<char>() + at;
    g->stage_off = at + bytes;
    return COVEST_OK;
}

int grid_stage_commit(covest_grid *g, StageSlot &slot, void *dst, size_t bytes)
{
    if (!g->async_uploads) {
        HIP_TRY(hipMemcpy(dst, slot.ptr, bytes, hipMemcpyHostToDevice));
        slot.shared_hold = std::unique_lock<std::mutex>();
        return COVEST_OK;
    }
    HIP_TRY(hipMemcpyAsync(dst, slot.ptr, bytes, hipMemcpyHostToDevice, nullptr));
    g->upload_pending = true;
    return COVEST_OK;
}

} // namespace covest

// optimize_grid re-configures a handle every iteration (21 times 0.2 ms of allocations and copies otherwise).
static int grid_configure(covest_grid *g, int32_t n_axes, const double *const *axes, const int64_t *axis_len,
                          int64_t flat_begin, int64_t flat_end, const char *who)
{
    covest_model *m = g->model;
    if (!axes || !axis_len)
        return fail(COVEST_E_INVALID, std::string(who) + ": null argument");
    if (n_axes != m->n_par)
        return fail(COVEST_E_INVALID, std::string(who) + ": n_axes must equal the model's param_count");
    int64_t total = 1, n_values = 0;
    for (int d = 0; d < n_axes; ++d) {
        if (axis_len[d] < 1 || !axes[d])
            return fail(COVEST_E_INVALID, std::string(who) + ": every axis needs at least one value");
        if (total > (int64_t)1 << 40)
            return fail(COVEST_E_INVALID, std::string(who) + ": grid too large");
        total *= axis_len[d];
        n_values += axis_len[d];
    }
    if (flat_end < 0)
        flat_end = total;
    if (flat_begin < 0 || flat_begin > flat_end || flat_end > total)
        return fail(COVEST_E_INVALID, std::string(who) + ": bad flat index range");
    g->configured = false; // (set again at the very end: a failure below leaves views into a freed arena behind)
    g->flat_begin = flat_begin;
    g->flat_end = flat_end;
    g->evaluated = false;
    g->scan_valid = false;
    g->ev_used = 0;
    g->stage_off = 0; // (a reset waited for the stream of the last evaluation: what the last reset staged has been copied)
    const int64_t n = flat_end - flat_begin;
    const int64_t n1 = n_axes == 5 ? axis_len[2] : 1, n2 = n_axes == 5 ? axis_len[3] : 1, n3 = n_axes == 5 ? axis_len[4] : 1;
    const int64_t nq = n_axes == 5 ? n1 * n2 * n3 : 0;

    // threshold_o over the (q1, q2, q) sub-grid (host, libm)
    std::vector<int32_t> table((size_t)nq);
    if (nq)
        threshold_table(m, axes[2], n1, axes[3], n2, axes[4], n3, table.data());

    // arena layout: [queue counter (8 B, always at the start: its place does not move from grid to grid) | axes | t_table]
    // uploaded together, then the outputs
    auto up8 = [](size_t v) { return (v + 7) / 8 * 8; };
    const size_t o_ctl = 0, o_axes = 8, o_table = o_axes + (size_t)n_values * sizeof(double);
    const size_t o_ll = up8(o_table + (size_t)nq * sizeof(int32_t)), n_pts = (size_t)(n > 0 ? n : 1);
    const size_t o_idx = o_ll + n_pts * sizeof(double), o_word = o_idx + n_pts * sizeof(int64_t);
    const size_t o_pv = o_word + n_pts * sizeof(unsigned long long), o_pi = o_pv + kArgminBlocks * sizeof(double);
    const size_t o_res = o_pi + kArgminBlocks * sizeof(int64_t), bytes = o_res + sizeof(ArgminResult);
    const void *arena_before = g->arena.ptr;
    const bool counter_clean = g->counter_clean; // (the queue counter is 0: set below, kept by every arg-min launch)
    HIP_TRY(g->arena.reserve(bytes));
    char *base = g->arena.as<char>();
    char *host_base = nullptr;
    bool in_place = false;
    {
        StageSlot slot;
        const int src = grid_stage_begin(g, o_ll, slot);
        if (src != COVEST_OK)
            return src;
        char *stage = slot.ptr;
        std::memset(stage, 0, o_ll);
        double *sa = reinterpret_cast<double *>(stage + o_axes);
        for (int d = 0; d < n_axes; ++d) {
            g->len[d] = axis_len[d];
            std::copy(axes[d], axes[d] + axis_len[d], sa);
            sa += axis_len[d];
        }
        for (int d = n_axes; d < kMaxParams; ++d)
            g->len[d] = 1;
        if (nq)
            std::memcpy(stage + o_table, table.data(), (size_t)nq * sizeof(int32_t));
        // A SMALL grid of a handle that is re-configured (optimize_grid's: a few thousand points, a kilobyte of axes and
        // threshold_o) reads its axes and the table WHERE THEY ARE STAGED -- page-locked host memory mapped into the
        // device's address space: every workgroup reads a handful of values, and the copy engine's start-up (11 us by
        // the trace of a search, a sixth of an iteration) is not paid.  The queue counter behind the axes is device
        // memory all the same (atomics): cleared by a memset on the stream.  Large grids -- every point of 10^6 reads
        // its axes -- keep the copy.
        in_place = g->async_uploads && n <= kArgminSmall;
        if (in_place) {
            host_base = stage;
            // (the counter's place depends on the axes' lengths: cleared unless it is where a clean one was left)
            if (!(counter_clean && g->arena.ptr == arena_before && g->sub_ctl.ptr == base + o_ctl)) {
                HIP_TRY(hipMemsetAsync(base + o_ctl, 0, 8, nullptr));
                g->upload_pending = true;
            }
        } else {
            const int crc = grid_stage_commit(g, slot, base, o_ll);
            if (crc != COVEST_OK)
                return crc;
        }
    }
    g->axes.ptr = (in_place ? host_base : base) + o_axes;
    g->sub_ctl.ptr = base + o_ctl;
    g->t_table.ptr = (in_place ? host_base : base) + o_table;
    g->ll.ptr = base + o_ll;
    g->sub_index.ptr = base + o_idx;
    g->sub_word.ptr = base + o_word;
    g->partial_val.ptr = base + o_pv;
    g->partial_idx.ptr = base + o_pi;
    g->result.ptr = base + o_res;
    PointSource &src = g->src;
    src = PointSource{};
    src.is_grid = 1;
    src.flat_begin = flat_begin;
    {
        int64_t off = 0;
        for (int d = 0; d < kMaxParams; ++d) {
            src.len[d] = d < n_axes ? axis_len[d] : 1;
            src.axis[d] = d < n_axes ? g->axes.as<double>() + off : nullptr;
            if (d < n_axes)
                off += axis_len[d];
        }
    }

    // the block's sum of (T - 1), and the K-factored plan
    g->sum_t_minus_1 = (double)n; // basic: T = 2 everywhere
    g->q_sum_t_minus_1 = 0.0;
    g->has_plan = false;
    if (m->n_par == 5) {
        src.t_table = g->t_table.as<int32_t>();
        // sum of (T-1) over flat indices [begin, end): whole (c,e) rows plus two ragged ends
        std::vector<double> prefix((size_t)nq + 1, 0.0);
        for (int64_t i = 0; i < nq; ++i)
            prefix[(size_t)i + 1] = prefix[(size_t)i] + (double)(table[(size_t)i] > 1 ? table[(size_t)i] - 1 : 0);
        auto upto = [&](int64_t flat) { // sum over flat indices [0, flat)
            return (double)(flat / nq) * prefix[(size_t)nq] + prefix[(size_t)(flat % nq)];
        };
        g->sum_t_minus_1 = upto(flat_end) - upto(flat_begin);
        g->q_sum_t_minus_1 = prefix[(size_t)nq];
        const int prc = build_factored_plan(g, axes, axis_len, table);
        if (prc != COVEST_OK)
            return prc;
    }
    if (g->upload_pending) { // an evaluation on another stream than the copies' waits for this
        if (!g->upload_ev)
            HIP_TRY(hipEventCreateWithFlags(&g->upload_ev, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(g->upload_ev, nullptr));
    }
    g->counter_clean = true;
    g->configured = true;
    return COVEST_OK;
}

static void grid_release(covest_grid *g)
{
    (void)hipDeviceSynchronize(); // (the buffers go back to the process's cache, host.h: nothing may still work on them)
    DeviceIdleScope idle;
    if (g->upload_ev) {
        (void)hipEventDestroy(g->upload_ev);
        g->upload_ev = nullptr;
    }
    g->stage.release();
    g->stage_retired.clear();
    if (g->result_host) {
        pinned_block_give(g->result_host);
        g->result_host = nullptr;
    }
    g->arena.release();
    g->plan_buf.release();
    for (covest_grid::Part &part : g->long_parts)
        part.buf.release();
    g->long_parts.clear();
    g->long_q_orig.release();
    g->long_partial.release();
    for (hipEvent_t e : g->ev_begin)
        (void)hipEventDestroy(e);
    for (hipEvent_t e : g->ev_end)
        (void)hipEventDestroy(e);
    g->ev_begin.clear();
    g->ev_end.clear();
}

extern "C" {

int covest_grid_create(covest_model *m, int32_t n_axes, const double *const *axes,
                       const int64_t *axis_len, int64_t flat_begin, int64_t flat_end,
                       covest_grid **out)
{
    if (!m || !out)
        return fail(COVEST_E_INVALID, "covest_grid_create: null argument");
    *out = nullptr;
    covest_grid *g = new (std::nothrow) covest_grid();
    if (!g)
        return fail(COVEST_E_NOMEM, "covest_grid_create: out of host memory");
    g->model = m;
    std::lock_guard<std::mutex> guard(m->lock);
    DeviceGuard dev_guard(m->device);
    int rc = dev_guard.status();
    if (rc == COVEST_OK)
        rc = grid_configure(g, n_axes, axes, axis_len, flat_begin, flat_end, "covest_grid_create");
    if (rc != COVEST_OK) {
        grid_release(g);
        delete g;
        return rc;
    }
    *out = g;
    return COVEST_OK;
}

int covest_grid_reset(covest_grid *g, int32_t n_axes, const double *const *axes, const int64_t *axis_len,
                      int64_t flat_begin, int64_t flat_end)
{
    if (!g)
        return fail(COVEST_E_INVALID, "covest_grid_reset: null grid");
    covest_model *m = g->model;
    std::lock_guard<std::mutex> guard(m->lock);
    DeviceGuard dev_guard(m->device);
    if (dev_guard.status() != COVEST_OK)
        return dev_guard.status();
    if (g->last_stream || g->evaluated)
        HIP_TRY(hipStreamSynchronize(g->last_stream)); // nothing of the last evaluation may still be in flight
    if (g->upload_pending && !g->evaluated)
        HIP_TRY(hipStreamSynchronize(nullptr)); // (a reset that was never evaluated: its copies read the staging memory)
    g->upload_pending = false;
    g->stage_retired.clear(); // (nothing of the last configuration is read any more)
    g->async_uploads = true; // (from the first reset on: a handle that is re-configured is re-configured often)
    return grid_configure(g, n_axes, axes, axis_len, flat_begin, flat_end, "covest_grid_reset");
}

void covest_grid_destroy(covest_grid *g)
{
    if (!g)
        return;
    DeviceGuard dev_guard(g->model->device);
    grid_release(g);
    delete g;
}

int covest_grid_profile(covest_grid *g, int32_t enable)
{
    if (!g)
        return fail(COVEST_E_INVALID, "covest_grid_profile: null grid");
    g->profiling = enable != 0;
    g->ev_used = 0;
    return COVEST_OK;
}

int covest_grid_kernel_ms(covest_grid *g, double *total_ms, int64_t *launches)
{
    if (!g || !total_ms || !launches)
        return fail(COVEST_E_INVALID, "covest_grid_kernel_ms: null argument");
    DeviceGuard dev_guard(g->model->device);
    int rc = dev_guard.status();
    if (rc != COVEST_OK)
        return rc;
    double sum = 0.0;
    for (size_t i = 0; i < g->ev_used; ++i) {
        HIP_TRY(hipEventSynchronize(g->ev_end[i]));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, g->ev_begin[i], g->ev_end[i]));
        sum += (double)ms;
    }
    *total_ms = sum;
    *launches = (int64_t)g->ev_used;
    return COVEST_OK;
}

int64_t covest_grid_size(const covest_grid *g) { return g ? g->flat_end - g->flat_begin : COVEST_E_INVALID; }

static int grid_eval(covest_grid *g, int32_t kernel, void *stream, bool scan, double scan_start)
{
    if (!g)
        return fail(COVEST_E_INVALID, "covest_grid_eval: null grid");
    if (!g->configured)
        return fail(COVEST_E_INVALID, "covest_grid_eval: the last covest_grid_reset of this handle failed; reset it again");
    covest_model *m = g->model;
    const int kern = resolve_kernel(m, kernel, g);
    if (kern < 0)
        return kern;
    std::lock_guard<std::mutex> guard(m->lock);
    DeviceGuard dev_guard(m->device);
    int rc = dev_guard.status();
    if (rc != COVEST_OK)
        return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t n = g->flat_end - g->flat_begin;
    if (g->upload_pending && st != nullptr && g->upload_ev)
        HIP_TRY(hipStreamWaitEvent(st, g->upload_ev, 0)); // (covest_grid_reset's copies run on the null stream)
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (g->profiling) {
        if (g->ev_used == g->ev_begin.size()) {
            hipEvent_t a, b;
            HIP_TRY(hipEventCreate(&a));
            HIP_TRY(hipEventCreate(&b));
            g->ev_begin.push_back(a);
            g->ev_end.push_back(b);
        }
        e0 = g->ev_begin[g->ev_used];
        e1 = g->ev_end[g->ev_used];
        g->ev_used++;
        HIP_TRY(hipEventRecord(e0, st));
    }
    HIP_TRY(launch_ll(m, kern, g->src, n, g->ll.as<double>(), sub_list_of(m, g->has_plan ? g->t_max : 2, g->sub_index.ptr, g->sub_word.ptr, g->sub_ctl.ptr), st,
                      &g->last_kernel, g));
    g->last_kernel_id = kern;
    if (e1)
        HIP_TRY(hipEventRecord(e1, st));
#ifdef COVEST_DIAG // diagnostic builds only: how many points the recurrence kernel handed back for the strict evaluation
    if (std::getenv("COVEST_DIAG_QUEUE")) {
        unsigned queued = 0;
        HIP_TRY(hipStreamSynchronize(st));
        HIP_TRY(hipMemcpy(&queued, g->sub_ctl.ptr, sizeof queued, hipMemcpyDeviceToHost));
        std::fprintf(stderr, "covest_grid_eval: %u of %lld points handed back\n", queued, (long long)n);
        if (queued > 0) { // the row ranges named: how long they are
            std::vector<unsigned long long> words(queued);
            HIP_TRY(hipMemcpy(words.data(), g->sub_word.ptr, (size_t)queued * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            double sum = 0;
            long long mx = 0, mn = 1 << 30, first_sum = 0;
            for (unsigned long long w : words) {
                const long long unit = sub_units16(w) ? 16 : 1;
                const long long len = ((long long)sub_last(w) - (long long)sub_first(w) + 1) * unit;
                sum += (double)len;
                mx = std::max(mx, len);
                mn = std::min(mn, len);
                first_sum += (long long)sub_first(w) * unit;
            }
            std::fprintf(stderr, "covest_grid_eval: rows named per point: min %lld mean %.1f max %lld; mean first row %.1f\n", mn,
                         sum / queued, mx, (double)first_sum / queued);
            if (g->src.t_table) { // the queued points' threshold_o
                const int64_t n_q = g->src.len[2] * g->src.len[3] * g->src.len[4];
                std::vector<int32_t> tt((size_t)n_q);
                std::vector<int64_t> idx(queued);
                HIP_TRY(hipMemcpy(tt.data(), g->src.t_table, (size_t)n_q * sizeof(int32_t), hipMemcpyDeviceToHost));
                HIP_TRY(hipMemcpy(idx.data(), g->sub_index.ptr, (size_t)queued * sizeof(int64_t), hipMemcpyDeviceToHost));
                long long tmin = 1 << 30, tmax = 0;
                double tsum = 0;
                long long hist[6] = {0, 0, 0, 0, 0, 0}; // T <= 16, 32, 64, 128, 256, more
                for (int64_t i : idx) {
                    const long long T = tt[(size_t)((g->src.flat_begin + i) % n_q)];
                    tmin = std::min(tmin, T);
                    tmax = std::max(tmax, T);
                    tsum += (double)T;
                    hist[T <= 16 ? 0 : T <= 32 ? 1 : T <= 64 ? 2 : T <= 128 ? 3 : T <= 256 ? 4 : 5]++;
                }
                std::fprintf(stderr, "covest_grid_eval: threshold_o of the queued points: min %lld mean %.1f max %lld; <=16 %lld <=32 %lld <=64 %lld <=128 %lld <=256 %lld more %lld\n",
                             tmin, tsum / queued, tmax, hist[0], hist[1], hist[2], hist[3], hist[4], hist[5]);
            }
        }
    }
#endif
    if (!g->result_host) { // (page-locked, mapped: argmin_stage2 stores the winner there itself)
        static_assert(sizeof(ArgminResult) <= 64 && 64 + sizeof(ScanRecords) <= kPinnedBlockBytes, "pinned block");
        g->result_host = static_cast<ArgminResult *>(pinned_block_take());
        if (!g->result_host)
            return fail(COVEST_E_NOMEM, "covest_grid_eval: no page-locked memory for the result");
    }
    g->scan_valid = scan && n >= 1 && n <= kArgminSmall;
    if (g->scan_valid)
        HIP_TRY(launch_argmin_scan(g->ll.as<double>(), n, g->flat_begin, scan_start, g->result.as<ArgminResult>(), g->result_host,
                                   g->scan_host(), g->sub_ctl.as<unsigned>(), st));
    else
        HIP_TRY(launch_argmin(g->ll.as<double>(), n, g->flat_begin, g->partial_val.as<double>(),
                              g->partial_idx.as<int64_t>(), g->result.as<ArgminResult>(), g->result_host, g->sub_ctl.as<unsigned>(), st));
    g->last_stream = st;
    g->evaluated = true;
    return COVEST_OK;
}

int covest_grid_eval(covest_grid *g, int32_t kernel, void *stream) { return grid_eval(g, kernel, stream, false, 0.0); }

int covest_grid_eval_scan(covest_grid *g, int32_t kernel, void *stream, double start_min)
{
    return grid_eval(g, kernel, stream, true, start_min);
}

int covest_grid_scan(covest_grid *g, int32_t cap, int64_t *index, double *negll, int32_t *n_records, int32_t *truncated)
{
    if (!g || !n_records || !truncated || cap < 0 || (cap > 0 && (!index || !negll)))
        return fail(COVEST_E_INVALID, "covest_grid_scan: bad argument");
    if (!g->evaluated)
        return fail(COVEST_E_INVALID, "covest_grid_scan: covest_grid_eval_scan has not run");
    DeviceGuard dev_guard(g->model->device);
    int rc = dev_guard.status();
    if (rc != COVEST_OK)
        return rc;
    HIP_TRY(hipStreamSynchronize(g->last_stream));
    *n_records = 0;
    *truncated = 1; // (a grid beyond one workgroup's reach, or a plain covest_grid_eval: no list -- read the values back)
    if (!g->scan_valid)
        return COVEST_OK;
    const ScanRecords *sr = g->scan_host(); // (the kernel's own stores to page-locked memory)
    const int32_t have = sr->n;
    *truncated = (sr->truncated || have > cap) ? 1 : 0;
    const int32_t take = std::min(have, cap);
    for (int32_t i = 0; i < take; ++i) {
        index[i] = sr->rec[i].index;
        negll[i] = sr->rec[i].negll;
    }
    *n_records = take;
    return COVEST_OK;
}

int covest_grid_argmin(covest_grid *g, double *min_negll, int64_t *argmin_flat)
{
    if (!g || !min_negll || !argmin_flat)
        return fail(COVEST_E_INVALID, "covest_grid_argmin: null argument");
    if (!g->evaluated)
        return fail(COVEST_E_INVALID, "covest_grid_argmin: covest_grid_eval has not run");
    DeviceGuard dev_guard(g->model->device);
    int rc = dev_guard.status();
    if (rc != COVEST_OK)
        return rc;
    HIP_TRY(hipStreamSynchronize(g->last_stream));
    const ArgminResult r = *g->result_host; // (the arg-min kernel's own store: covest_grid_eval)
    *min_negll = r.min_negll;
    *argmin_flat = r.index < 0 ? -1 : g->flat_begin + r.index;
    return COVEST_OK;
}

const double *covest_grid_ll_device(const covest_grid *g) { return g ? g->ll.as<double>() : nullptr; }

const double *covest_grid_argmin_pair_device(const covest_grid *g)
{
    return g ? g->result.as<ArgminResult>()->pair : nullptr; // (address arithmetic only: nothing is read here)
}

int covest_grid_ll_host(covest_grid *g, double *out_ll)
{
    if (!g || !out_ll)
        return fail(COVEST_E_INVALID, "covest_grid_ll_host: null argument");
    if (!g->evaluated)
        return fail(COVEST_E_INVALID, "covest_grid_ll_host: covest_grid_eval has not run");
    DeviceGuard dev_guard(g->model->device);
    int rc = dev_guard.status();
    if (rc != COVEST_OK)
        return rc;
    const int64_t n = g->flat_end - g->flat_begin;
    HIP_TRY(hipStreamSynchronize(g->last_stream));
    if (n > 0)
        HIP_TRY(hipMemcpy(out_ll, g->ll.ptr, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    return COVEST_OK;
}


int64_t covest_grid_diag(covest_grid *g, int64_t *out, int64_t n)
{
    if (!g || !g->has_plan || !g->plan.diag)
        return 0;
    const int nw = g->plan.n_threads / 64;
    const int64_t total = (g->plan.ce_end - g->plan.ce_begin) * g->plan.n_qblocks * nw * 8;
    if (out && n > 0) {
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(out, g->plan.diag, (size_t)std::min(n, total) * sizeof(int64_t), hipMemcpyDeviceToHost);
    }
    return total;
}

int covest_grid_work(const covest_grid *g, double *pmf_terms, double *flops, const char **kernel)
{
    if (!g)
        return fail(COVEST_E_INVALID, "covest_grid_work: null grid");
    const covest_model *m = g->model;
    const double n = (double)(g->flat_end - g->flat_begin);
    const double bins = (double)m->dm.bins.n;
    const double S = (double)m->dm.n_err;
    // SURVEY 8(d) unit: pmf terms of the per-point formulation, bins * S * sum(T - 1)
    const double terms = bins * S * g->sum_t_minus_1;
    if (pmf_terms)
        *pmf_terms = terms;
    if (flops) {
        if (g->last_kernel_id == COVEST_KERNEL_FACTORED) {
            // algorithmic minimum of the factored formulation (ll_factored.hip header):
            // per (c,e): G build 2 flop per (key, o, s), contraction 2 flop per (row, q, o < T_q) -- a row is a key,
            // or the sum of a whole count-less tile (tail != 0, tiles.h); where the plan shares steps between the
            // columns of a q-tile (tiles.h) the shared sums count once --, one log (25 flop) per
            // (counted key, q), prologue exps 25 per (o, s)
            const double n_ce = (double)(g->plan.ce_end - g->plan.ce_begin);
            const double max_o = (double)(g->t_max - 1);
            const double rows = m->tail_is_zero ? bins : m->rows_contracted;
            const double logged = m->tail_is_zero ? bins : m->keys_logged;
            *flops = n_ce * (bins * S * max_o * 2.0 + rows * g->contract_flops_per_row +
                             logged * (double)g->plan.n_q * 25.0 + 25.0 * S * max_o);
        } else {
            // 4 flop per pmf term + 25 per log + 25 per exp of the prologue
            *flops = 4.0 * terms + 25.0 * bins * n + 25.0 * S * g->sum_t_minus_1;
        }
    }
    if (kernel)
        *kernel = g->last_kernel;
    return COVEST_OK;
}


} // extern "C"
