// abi_model.cpp -- the model handle of the C ABI (include/covest_amd.h): covest_model_*, covest_eval_points,
// covest_probabilities, covest_reference_overflow, and the kernel dispatch the grid entry points share.
#include "host.h"

using namespace covest;

extern "C" {

int covest_model_create(const covest_model_desc *d, covest_model **out)
{
    if (!d || !out)
        return fail(COVEST_E_INVALID, "covest_model_create: null argument");
    *out = nullptr;
    if (d->kind != COVEST_MODEL_BASIC && d->kind != COVEST_MODEL_REPEATS)
        return fail(COVEST_E_INVALID, "covest_model_create: unknown model kind");
    if (d->n_err < 1 || d->n_err > COVEST_MAX_ERROR_CLASSES || d->n_err > d->k + 1)
        return fail(COVEST_E_INVALID, "covest_model_create: n_err must be in 1..min(k+1, 64)");
    if (d->r <= 0 || d->k <= 0)
        return fail(COVEST_E_INVALID, "covest_model_create: k and r must be positive");
    if (d->n_keys < 0 || (d->n_keys > 0 && (!d->keys || !d->counts)) || !d->comb)
        return fail(COVEST_E_INVALID, "covest_model_create: null histogram or comb");
    if (d->kind == COVEST_MODEL_REPEATS && d->n_keys == 0)
        return fail(COVEST_E_INVALID,
                    "covest_model_create: repeats model needs a non-empty histogram "
                    "(max() of an empty dict raises in covest/models.py:186)");

    int device = 0;
    {
        const int drc = resolve_device(d->device, "covest_model_create", &device);
        if (drc != COVEST_OK)
            return drc;
    }

    covest_model *m = new (std::nothrow) covest_model();
    if (!m)
        return fail(COVEST_E_NOMEM, "covest_model_create: out of host memory");
    m->device = device;
    m->n_par = d->kind == COVEST_MODEL_BASIC ? 2 : 5;
    m->n_keys = d->n_keys;
    m->threshold = d->threshold;
    m->has_threshold = d->has_threshold != 0;
    DevModel &dm = m->dm;
    dm.kind = d->kind;
    dm.k = d->k;
    dm.r = d->r;
    dm.n_err = d->n_err;
    for (int s = 0; s < kMaxErr; ++s) {
        dm.comb[s] = s < d->n_err ? d->comb[s] : 0.0;
        dm.pow3neg[s] = std::pow(3.0, (double)-s); // 3 ** -s, covest/models.py:77
        dm.ln_comb[s] = dm.comb[s] > 0.0 ? std::log(dm.comb[s]) : -INFINITY;
    }
    for (int i = 0; i < kMaxParams; ++i) {
        dm.lo[i] = i < m->n_par ? d->lo[i] : std::numeric_limits<double>::quiet_NaN();
        dm.hi[i] = i < m->n_par ? d->hi[i] : std::numeric_limits<double>::quiet_NaN();
    }
    dm.tail = d->tail;

    // Bin views.  When tail == 0 the tail term of covest/models.py:104 is exactly
    // 0 whatever sp_j is (0 * log of a positive number, or the else-branch), so
    // bins with h_j == 0 influence nothing and are dropped from the evaluated view.
    std::vector<double> key_a, lg_a, cnt_a, key_e, lg_e, cnt_e;
    std::vector<HostBin> eval_bins;
    const bool keep_all = d->tail == 0.0; // (with a tail the evaluated view IS the full view)
    if (keep_all) {
        key_a.reserve((size_t)d->n_keys);
        lg_a.reserve((size_t)d->n_keys);
        cnt_a.reserve((size_t)d->n_keys);
    }
    int hist_max = std::numeric_limits<int>::min();
    for (int64_t b = 0; b < d->n_keys; ++b)
        hist_max = std::max(hist_max, (int)d->keys[b]);
    lgamma_ensure(hist_max); // one lock for the whole histogram
    for (int64_t b = 0; b < d->n_keys; ++b) {
        const int j = d->keys[b];
        const int je = j > 0 ? j : 0; // the product loop of the C extension is empty for j <= 0
        const double kd = (double)je;
        const double lg = lgamma_at(je);
        const double h = d->counts[b];
        if (keep_all) {
            key_a.push_back(kd);
            lg_a.push_back(lg);
            cnt_a.push_back(h);
        }
        if (d->tail != 0.0 || h != 0.0) {
            key_e.push_back(kd);
            lg_e.push_back(lg);
            cnt_e.push_back(h);
            eval_bins.push_back({j, h, (int32_t)key_e.size() - 1});
        }
    }
    m->hist_max = d->n_keys > 0 ? hist_max : 0;
    m->tail_is_zero = d->tail == 0.0;
    m->key_max = hist_max > 0 ? hist_max : 0;

    DeviceGuard dev_guard(m->device);
    int rc = dev_guard.status();
    if (d->tail != 0.0) { // the evaluated view IS the full view
        m->host_all_key.clear();
    } else {
        m->host_all_key = std::move(key_a);
        m->host_all_lgam = std::move(lg_a);
        m->host_all_cnt = std::move(cnt_a);
    }
    if (rc == COVEST_OK)
        rc = upload_bins(m->bins_eval, dm.bins, key_e, lg_e, cnt_e);
    if (rc == COVEST_OK && d->tail != 0.0) {
        m->all_bins = dm.bins;
        m->all_bins_ready = true;
    }
    if (rc == COVEST_OK)
        rc = build_tiles(m, std::move(eval_bins));
    if (rc != COVEST_OK) {
        covest_model_destroy(m);
        return rc;
    }
    *out = m;
    return COVEST_OK;
}

void covest_model_destroy(covest_model *m)
{
    if (!m)
        return;
    DeviceGuard dev_guard(m->device);
    (void)hipDeviceSynchronize(); // (its small buffers go back to the process's cache: nothing may still work on them)
    DeviceIdleScope idle;
    delete m; // (its buffers go with it: host.h DevBuf / HostBuf)
}

int covest_model_param_count(const covest_model *m) { return m ? m->n_par : COVEST_E_INVALID; }

int64_t covest_model_bins_evaluated(const covest_model *m) { return m ? m->dm.bins.n : COVEST_E_INVALID; }


} // extern "C"

// scaling, so truncated_poisson(l, j) is +inf as soon as the running product passes LDBL_MAX -- its largest
// value is reached at i = min(j, floor(l)): l^i / i!.  A likelihood evaluation calls it for every key j of the
// histogram and every l = o * l_s, o < threshold_o (covest/models.py:92-97, :235-241); the largest l against the
// largest key decides.  The kernels return the finite value the formula defines; this reports, per point,
// whether the reference itself would have returned inf / NaN there (and optimize_grid, covest/grid.py:65-70,
// would have selected it).
static bool reference_product_overflows(long double l, int64_t j_max)
{
    if (!(l > 0.0L) || j_max < 1)
        return false;
    const long double i_top = std::min<long double>((long double)j_max, floorl(l));
    if (i_top < 1.0L)
        return false;
    const long double ln_ldbl_max = 11356.523406294143949492L;
    return i_top * logl(l) - lgammal(i_top + 1.0L) > ln_ldbl_max;
}

static bool reference_overflows_at(const DevModel &dm, int n_par, const double *par_in, int T, int64_t key_max)
{
    double par[kMaxParams] = {0, 0, 0, 0, 0};
    for (int d = 0; d < n_par; ++d)
        par[d] = clamp_one(dm, d, par_in[d]);
    const double ck = par[0] * (double)(dm.r - dm.k + 1) / (double)dm.r; // covest/models.py:71-72
    double l_max = 0.0;
    for (int sidx = 0; sidx < dm.n_err; ++sidx) { // covest/models.py:76-79, same evaluation order
        double v = ck * dm.pow3neg[sidx];
        v = v * std::pow(1.0 - par[1], (double)(dm.k - sidx));
        v = v * std::pow(par[1], (double)sidx);
        if (v > l_max)
            l_max = v;
    }
    const int o_max = n_par == 5 ? T - 1 : 1;
    return o_max >= 1 && reference_product_overflows((long double)((double)o_max * l_max), key_max);
}


namespace covest {

// Resolve COVEST_KERNEL_* for a request (g == nullptr: a point list).  Returns the
// kernel to run or a negative error.
int resolve_kernel(const covest_model *m, int32_t kernel, const covest_grid *g)
{
    const bool basic_fast = m->has_tiles && m->dm.kind == COVEST_MODEL_BASIC;
    const bool factored_ok = g && g->has_plan;
    switch (kernel) {
    case COVEST_KERNEL_AUTO:
        if (basic_fast)
            return COVEST_KERNEL_RECUR;
        // the factored kernel pays when many weight vectors share each (c, e)
        if (factored_ok && g->plan.n_q >= 32)
            return COVEST_KERNEL_FACTORED;
        // a repeats-model point list: one workgroup per (point, key segment) (list mode) instead of one wave --
        // the latency path of refinements.  Long lists are throughput work and go to K-direct (and the list
        // mode's per-point tables, 13 KB each, stay small): see covest_eval_points.
        if (!g && m->has_tiles && m->dm.kind == COVEST_MODEL_REPEATS && m->dm.n_err <= 8)
            return COVEST_KERNEL_FACTORED;
        return COVEST_KERNEL_DIRECT;
    case COVEST_KERNEL_DIRECT:
        return COVEST_KERNEL_DIRECT;
    case COVEST_KERNEL_DIRECT_REF:
        return COVEST_KERNEL_DIRECT_REF;
    case COVEST_KERNEL_RECUR:
        if (basic_fast)
            return COVEST_KERNEL_RECUR;
        return fail(COVEST_E_INVALID, "recurrence kernel needs the basic model, max_error <= 32 and keys in 1..16384");
    case COVEST_KERNEL_FACTORED:
        if (factored_ok || (!g && m->has_tiles && m->dm.kind == COVEST_MODEL_REPEATS && m->dm.n_err <= 8))
            return COVEST_KERNEL_FACTORED;
        return fail(COVEST_E_INVALID, "factored kernel needs the repeats model, keys in 1..16384 and max_error <= 32 "
                                      "(<= 8 for a point list)");
    default:
        return fail(COVEST_E_INVALID, "unknown kernel");
    }
}

// K-factored on a dense grid: the part whose weight vectors fit a workgroup's lanes writes log-likelihoods (and is
// followed by the pass that patches what it handed back); the long weight vectors go chunk by chunk of copy numbers
// into an HBM buffer of p_j, one batch of (c, e) rows at a time, and ll_finish_dense takes their logs.
static hipError_t launch_factored_grid(covest_grid *g, double *out, const SubList &sub, hipStream_t st)
{
    const covest_model *m = g->model;
    if (g->has_short_part) {
        hipError_t e = launch_ll_factored(m->dm, m->tv, g->plan, out, sub, st);
        if (e != hipSuccess)
            return e;
        e = launch_ll_fix_list(m->dm, m->tv, g->src, out, sub, st, g->flat_end - g->flat_begin);
        if (e != hipSuccess)
            return e;
    }
    if (g->n_long_tiles > 0) {
        const int64_t n_cols = (int64_t)g->n_long_tiles * 16, n_rows = (int64_t)m->tv.n_items * kTileBins;
        const int64_t ce_begin = g->plan.ce_begin, ce_end = g->plan.ce_end;
        const int64_t per_ce = n_cols * n_rows * (int64_t)sizeof(double);
        const int64_t batch = std::max<int64_t>(1, std::min<int64_t>(ce_end - ce_begin, ((int64_t)1 << 30) / per_ce));
        hipError_t e = g->long_partial.reserve((size_t)(batch * per_ce));
        if (e != hipSuccess)
            return e;
        for (int64_t first = ce_begin; first < ce_end; first += batch) {
            const int64_t last = std::min(ce_end, first + batch);
            for (covest_grid::Part &part : g->long_parts) {
                FactoredPlan pl = part.plan;
                pl.ce_begin = first;
                pl.ce_end = last;
                pl.ce_first = first;
                pl.n_cols_partial = n_cols;
                pl.partial = g->long_partial.as<double>();
                e = launch_ll_factored(m->dm, m->tv, pl, out, sub, st);
                if (e != hipSuccess)
                    return e;
            }
            e = launch_ll_finish_dense(m->dm, m->tv, g->src, g->long_partial.as<double>(), first, last - first, n_cols,
                                       g->long_q_orig.as<int32_t>(), g->plan.n_q, g->flat_end, out, st);
            if (e != hipSuccess)
                return e;
        }
    }
    return hipSuccess;
}

SubList sub_list_of(const covest_model *m, int t_max, void *index, void *word, void *ctl)
{
    SubList l{};
    l.p_clamp = clamp_for(m, t_max);
    l.log_p_clamp = std::log(l.p_clamp);
    l.count = static_cast<unsigned *>(ctl);
    l.index = static_cast<int64_t *>(index);
    l.word = static_cast<unsigned long long *>(word);
    l.index_offset = 0;
#ifdef COVEST_DIAG // diagnostic builds only (direct_point.h SubList::diag_class): the shipped library has no knobs
    const char *dc = std::getenv("COVEST_DIAG_BASIC_CLASS");
    l.diag_class = dc ? std::atoi(dc) : 0;
#endif
    return l;
}


// `sub`: the queue the recurrence kernels append the points they hand back to (direct_point.h) -- drained right
// behind them by the fix pass; K-direct has nothing to hand back.  The queue must be empty (counter 0) on entry.
hipError_t launch_ll(const covest_model *m, int kernel, const PointSource &src, int64_t n,
                            double *out, const SubList &sub, hipStream_t st, const char **name,
                            const covest_grid *g)
{
    if (kernel == COVEST_KERNEL_FACTORED) {
        if (name)
            *name = "ll_factored";
        return launch_factored_grid(const_cast<covest_grid *>(g), out, sub, st);
    }
    if (kernel == COVEST_KERNEL_RECUR) {
        if (name)
            *name = "ll_basic";
        hipError_t e = launch_ll_basic(m->dm, m->tv, src, n, out, sub, st);
        return e != hipSuccess ? e : launch_ll_fix_list(m->dm, m->tv, src, out, sub, st, n);
    }
    if (name)
        *name = kernel == COVEST_KERNEL_DIRECT_REF ? "ll_direct_ref" : "ll_direct";
    return launch_ll_direct(m->dm, src, n, out, nullptr, st, kernel == COVEST_KERNEL_DIRECT_REF);
}

// Workspace of a point-list launch's queue (direct_point.h): room for n entries, counters zeroed on first use.

} // namespace covest

constexpr int64_t kInPlaceMaxPoints = 256;   // point lists up to this size: parameters and values in mapped host memory
constexpr int64_t kInPlaceMaxListPoints = 4; // repeats model, list mode: tables read in place (13 KB a point, 8 workgroups each)

static int reserve_point_queue(covest_model *m, int64_t n)
{
    HIP_TRY(m->ws_sub_index.reserve((size_t)n * sizeof(int64_t)));
    HIP_TRY(m->ws_sub_word.reserve((size_t)n * sizeof(unsigned long long)));
    HIP_TRY(m->ws_sub_ctl.reserve(sizeof(unsigned)));
    HIP_TRY(hipMemset(m->ws_sub_ctl.ptr, 0, sizeof(unsigned))); // the queue starts empty (every point-list call)
    return COVEST_OK;
}

// Point lists through K-factored's list mode: add the strict evaluation of the rows the kernel handed back
// (words[i] != 0, direct_point.h) to out_ll[i].  Called with the model locked.
static int fix_points_host(covest_model *m, int64_t n, const double *params, double *out_ll,
                           const std::vector<unsigned long long> &words)
{
    std::vector<int64_t> again;
    for (int64_t i = 0; i < n; ++i)
        if (words[(size_t)i] != 0 && std::isfinite(out_ll[i]))
            again.push_back(i);
    if (again.empty())
        return COVEST_OK;
    const int P = m->n_par;
    const size_t na = again.size();
    std::vector<double> sub_par(na * (size_t)P), sub_ll(na);
    std::vector<int32_t> sub_t(na, 2);
    std::vector<unsigned long long> sub_w(na);
    std::vector<int64_t> sub_i(na);
    for (size_t k = 0; k < na; ++k) {
        std::memcpy(&sub_par[k * (size_t)P], params + again[k] * P, (size_t)P * sizeof(double));
        if (P == 5)
            sub_t[k] = threshold_for_point(m, params + again[k] * P);
        sub_ll[k] = out_ll[again[k]];
        sub_w[k] = words[(size_t)again[k]];
        sub_i[k] = (int64_t)k;
    }
    int rc = reserve_point_queue(m, (int64_t)na);
    if (rc != COVEST_OK)
        return rc;
    HIP_TRY(m->ws_params.reserve(sub_par.size() * sizeof(double)));
    HIP_TRY(m->ws_t.reserve(na * sizeof(int32_t)));
    HIP_TRY(m->ws_out.reserve(na * sizeof(double)));
    HIP_TRY(hipMemcpy(m->ws_params.ptr, sub_par.data(), sub_par.size() * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(m->ws_t.ptr, sub_t.data(), na * sizeof(int32_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(m->ws_out.ptr, sub_ll.data(), na * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(m->ws_sub_index.ptr, sub_i.data(), na * sizeof(int64_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(m->ws_sub_word.ptr, sub_w.data(), na * sizeof(unsigned long long), hipMemcpyHostToDevice));
    const unsigned count = (unsigned)na;
    HIP_TRY(hipMemcpy(m->ws_sub_ctl.ptr, &count, sizeof count, hipMemcpyHostToDevice));
    PointSource src{};
    src.is_grid = 0;
    src.params = m->ws_params.as<double>();
    src.t_list = P == 5 ? m->ws_t.as<int32_t>() : nullptr;
    HIP_TRY(launch_ll_fix_list(m->dm, m->tv, src, m->ws_out.as<double>(),
                               sub_list_of(m, m->n_par == 5 ? 513 : 2, m->ws_sub_index.ptr, m->ws_sub_word.ptr, m->ws_sub_ctl.ptr), nullptr,
                               (int64_t)na));
    HIP_TRY(hipMemcpy(sub_ll.data(), m->ws_out.ptr, na * sizeof(double), hipMemcpyDeviceToHost));
    for (size_t k = 0; k < na; ++k)
        out_ll[again[k]] = sub_ll[k];
    return COVEST_OK;
}


extern "C" {

int covest_eval_points(covest_model *m, int64_t n, const double *params, double *out_ll,
                       int32_t kernel)
{
    if (!m || n < 0 || (n > 0 && (!params || !out_ll)))
        return fail(COVEST_E_INVALID, "covest_eval_points: bad argument");
    if (n == 0)
        return COVEST_OK;
    int kern = resolve_kernel(m, kernel, nullptr);
    if (kern < 0)
        return kern;
    if (kern == COVEST_KERNEL_FACTORED && n > kListModeMaxPoints) {
        if (kernel == COVEST_KERNEL_FACTORED)
            return fail(COVEST_E_INVALID, "factored kernel: a point list of more than 4096 points (use a grid, or K-direct)");
        kern = COVEST_KERNEL_DIRECT; // AUTO: throughput work
    }
    std::lock_guard<std::mutex> guard(m->lock);
    DeviceGuard dev_guard(m->device);
    int rc = dev_guard.status();
    if (rc != COVEST_OK)
        return rc;
    const int P = m->n_par;
    // A SMALL list (what scipy's refinement issues: a point and its P finite-difference neighbours) moves nothing through
    // the copy engine (round 5): its parameters are read, and its values written, IN PLACE in page-locked host memory
    // mapped into the device's address space -- one launch (two with the strict pass) and one wait for the stream; a
    // blocking copy either side of the launch was two thirds of a basic-model evaluation's 48 us.
    const bool in_place = n <= kInPlaceMaxPoints;
    PointSource src{};
    src.is_grid = 0;
    double *out_dev = nullptr; // where the kernels leave the values: HBM, or (in_place) the mapped host block
    SubList queue = sub_list_of(m, m->n_par == 5 ? 513 : 2, nullptr, nullptr, nullptr); // (list mode and K-direct hand nothing back)
    if (kern != COVEST_KERNEL_FACTORED) {
        std::vector<int32_t> t;
        if (P == 5) {
            t.resize((size_t)n);
            for (int64_t i = 0; i < n; ++i)
                t[(size_t)i] = threshold_for_point(m, params + i * P);
        }
        const size_t par_bytes = (size_t)n * P * sizeof(double), t_bytes = ((size_t)n * sizeof(int32_t) + 7) / 8 * 8;
        if (in_place) {
            HIP_TRY(m->ws_stage.reserve(par_bytes + t_bytes));
            HIP_TRY(m->ws_result.reserve((size_t)n * sizeof(double)));
            char *st = m->ws_stage.as<char>();
            std::memcpy(st, params, par_bytes);
            if (P == 5)
                std::memcpy(st + par_bytes, t.data(), (size_t)n * sizeof(int32_t));
            src.params = reinterpret_cast<const double *>(st);
            src.t_list = P == 5 ? reinterpret_cast<const int32_t *>(st + par_bytes) : nullptr;
            out_dev = m->ws_result.as<double>();
        } else {
            HIP_TRY(m->ws_params.reserve(par_bytes));
            HIP_TRY(m->ws_out.reserve((size_t)n * sizeof(double)));
            HIP_TRY(hipMemcpy(m->ws_params.ptr, params, par_bytes, hipMemcpyHostToDevice));
            src.params = m->ws_params.as<double>();
            if (P == 5) {
                HIP_TRY(m->ws_t.reserve((size_t)n * sizeof(int32_t)));
                HIP_TRY(hipMemcpy(m->ws_t.ptr, t.data(), (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice));
                src.t_list = m->ws_t.as<int32_t>();
            }
            out_dev = m->ws_out.as<double>();
        }
        if (kern == COVEST_KERNEL_RECUR) { // K-basic hands points back through the queue (direct_point.h): empty on entry
            HIP_TRY(m->ws_sub_index.reserve((size_t)n * sizeof(int64_t)));
            HIP_TRY(m->ws_sub_word.reserve((size_t)n * sizeof(unsigned long long)));
            HIP_TRY(m->ws_sub_ctl.reserve(sizeof(unsigned)));
            HIP_TRY(hipMemsetAsync(m->ws_sub_ctl.ptr, 0, sizeof(unsigned), nullptr));
            queue = sub_list_of(m, 2, m->ws_sub_index.ptr, m->ws_sub_word.ptr, m->ws_sub_ctl.ptr);
        }
    } else {
        HIP_TRY(m->ws_out.reserve((size_t)n * sizeof(double)));
        HIP_TRY(m->ws_params.reserve((size_t)n * P * sizeof(double)));
        HIP_TRY(m->ws_t.reserve((size_t)n * sizeof(int32_t)));
        src.params = m->ws_params.as<double>();
        src.t_list = m->ws_t.as<int32_t>();
    }
    if (kern == COVEST_KERNEL_FACTORED) {
        // repeats model point list: one workgroup per point (build_list_plan); a point whose threshold_o
        // exceeds a workgroup's 512 lanes is cut into chunks of 512 copy numbers, one workgroup each, and
        // finished by ll_finish_partials.  A point's route depends on its own threshold_o only -- never on
        // what else is in the call (refinements compare values across calls).  threshold_o == 1 (nothing to
        // sum) goes to K-direct.
        std::vector<int32_t> t((size_t)n);
        std::vector<int64_t> fits, big, rest;
        std::vector<unsigned long long> words((size_t)n, 0ull); // keys handed back per point (direct_point.h)
        for (int64_t i = 0; i < n; ++i) {
            t[(size_t)i] = threshold_for_point(m, params + i * P);
            const int o_max = t[(size_t)i] - 1;
            (o_max < 1 ? rest : o_max <= 512 ? fits : big).push_back(i);
        }
        if (!fits.empty()) {
            std::vector<double> sub_par(fits.size() * 5);
            std::vector<int32_t> sub_t(fits.size());
            for (size_t k = 0; k < fits.size(); ++k) {
                std::memcpy(&sub_par[k * 5], params + fits[k] * 5, 5 * sizeof(double));
                sub_t[k] = t[(size_t)fits[k]];
            }
            FactoredPlan pl;
            rc = build_list_plan(m, (int64_t)fits.size(), sub_par.data(), sub_t, nullptr, m->ws_plan, pl,
                                 (int64_t)fits.size() <= kInPlaceMaxListPoints);
            if (rc != COVEST_OK)
                return rc;
            // {LL part, sp_j part (hi, lo), side word} per (point, key segment); the segments are added here, in order
            // (round 5: the kernel stores them straight into page-locked, device-mapped host memory -- 256 bytes a point --
            // and the list's tables go up asynchronously: ONE wait for the stream per call instead of a blocking copy
            // either side of the launch, a third of a single evaluation's 75 us)
            const size_t n_parts = fits.size() * (size_t)pl.n_seg;
            m->ws_result.flags = hipHostMallocPortable | hipHostMallocMapped;
            HIP_TRY(m->ws_result.reserve(n_parts * 4 * sizeof(double)));
            pl.partial = m->ws_result.as<double>();
            HIP_TRY(launch_ll_factored(m->dm, m->tv, pl, m->ws_out.as<double>(), queue, nullptr));
            HIP_TRY(hipStreamSynchronize(nullptr));
            const double *got = m->ws_result.as<double>();
            for (size_t k = 0; k < fits.size(); ++k) {
                double ll = 0.0, hi = 0.0, lo = 0.0;
                unsigned u_first = 0xFFFFFFFFu, u_last = 0; // the segments' handed-back units, merged
                bool any_unit = false;
                for (int sg = 0; sg < pl.n_seg; ++sg) {
                    const double *o = &got[(k * (size_t)pl.n_seg + (size_t)sg) * 4];
                    ll += o[0];
                    unsigned long long w;
                    std::memcpy(&w, &o[3], sizeof w);
                    if (w != 0) {
                        any_unit = true;
                        u_first = std::min(u_first, sub_first(w));
                        u_last = std::max(u_last, sub_last(w));
                    }
                    const double sum = hi + o[1], bb = sum - hi; // two-sum, as the kernels' CompSum
                    lo += ((hi - (sum - bb)) + (o[1] - bb)) + o[2];
                    hi = sum;
                }
                double tail_term = 0.0;
                if (m->dm.tail != 0.0) { // tail * log(1 - min(1, sp)), covest/models.py:103-105
                    double sp = hi + lo;
                    if (!(sp < 1.0))
                        sp = 1.0;
                    if (sp < 1.0)
                        tail_term = m->dm.tail * std::log(1.0 - sp);
                }
                out_ll[fits[k]] = ll + tail_term;
                words[(size_t)fits[k]] = any_unit ? sub_word(u_first, u_last, true) : 0ull;
            }
        }
        if (!big.empty()) {
            std::vector<double> item_par, point_par(5 * big.size());
            std::vector<int32_t> item_t, item_ob, first_item(big.size() + 1, 0), point_t(big.size());
            for (size_t k = 0; k < big.size(); ++k) {
                const double *par = params + big[k] * 5;
                std::memcpy(&point_par[5 * k], par, 5 * sizeof(double));
                point_t[k] = t[(size_t)big[k]];
                for (int ob = 0; ob < t[(size_t)big[k]] - 1; ob += 512) {
                    item_par.insert(item_par.end(), par, par + 5);
                    item_t.push_back(t[(size_t)big[k]]);
                    item_ob.push_back(ob);
                }
                first_item[k + 1] = (int32_t)item_t.size();
            }
            const int64_t n_items = (int64_t)item_t.size();
            const size_t n_keys = (size_t)m->tv.n_items * kTileBins; // rows of the items (tiles.h)
            FactoredPlan pl;
            rc = build_list_plan(m, n_items, item_par.data(), item_t, &item_ob, m->ws_plan2, pl);
            if (rc != COVEST_OK)
                return rc;
            HIP_TRY(m->ws_partial.reserve((size_t)n_items * n_keys * sizeof(double)));
            const size_t items_bytes = (size_t)n_items * sizeof(int32_t), first_bytes = first_item.size() * sizeof(int32_t);
            const size_t pt_bytes = point_t.size() * sizeof(int32_t);
            const size_t int_bytes = ((items_bytes + first_bytes + pt_bytes + 7) / 8) * 8;
            HIP_TRY(m->ws_items.reserve(int_bytes + point_par.size() * sizeof(double)));
            char *ib = m->ws_items.as<char>();
            {
                std::vector<char> stage(int_bytes + point_par.size() * sizeof(double)); // one copy
                std::memcpy(stage.data(), item_ob.data(), items_bytes);
                std::memcpy(stage.data() + items_bytes, first_item.data(), first_bytes);
                std::memcpy(stage.data() + items_bytes + first_bytes, point_t.data(), pt_bytes);
                std::memcpy(stage.data() + int_bytes, point_par.data(), point_par.size() * sizeof(double));
                HIP_TRY(hipMemcpy(ib, stage.data(), stage.size(), hipMemcpyHostToDevice));
            }
            pl.list_mode = 2;
            pl.item_obase = reinterpret_cast<const int32_t *>(ib);
            pl.partial = m->ws_partial.as<double>();
            HIP_TRY(launch_ll_factored(m->dm, m->tv, pl, m->ws_out.as<double>(), queue, nullptr));
            HIP_TRY(launch_ll_finish_partials(m->dm, m->tv, pl.partial, reinterpret_cast<const int32_t *>(ib + items_bytes),
                                              reinterpret_cast<const double *>(ib + int_bytes),
                                              reinterpret_cast<const int32_t *>(ib + items_bytes + first_bytes),
                                              (int64_t)big.size(), m->ws_out.as<double>(), nullptr));
            std::vector<double> got(big.size());
            HIP_TRY(hipMemcpy(got.data(), m->ws_out.ptr, big.size() * sizeof(double), hipMemcpyDeviceToHost));
            for (size_t k = 0; k < big.size(); ++k)
                out_ll[big[k]] = got[k];
        }
        if (!rest.empty()) {
            std::vector<double> sub_par(rest.size() * 5);
            std::vector<int32_t> sub_t(rest.size());
            for (size_t k = 0; k < rest.size(); ++k) {
                std::memcpy(&sub_par[k * 5], params + rest[k] * 5, 5 * sizeof(double));
                sub_t[k] = t[(size_t)rest[k]];
            }
            HIP_TRY(hipMemcpy(m->ws_params.ptr, sub_par.data(), sub_par.size() * sizeof(double), hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(m->ws_t.ptr, sub_t.data(), sub_t.size() * sizeof(int32_t), hipMemcpyHostToDevice));
            HIP_TRY(launch_ll(m, COVEST_KERNEL_DIRECT, src, (int64_t)rest.size(), m->ws_out.as<double>(), queue, nullptr, nullptr));
            std::vector<double> got(rest.size());
            HIP_TRY(hipMemcpy(got.data(), m->ws_out.ptr, rest.size() * sizeof(double), hipMemcpyDeviceToHost));
            for (size_t k = 0; k < rest.size(); ++k)
                out_ll[rest[k]] = got[k];
        }
        return fix_points_host(m, n, params, out_ll, words);
    }
    // (K-basic is followed by the pass that patches the points it handed back: launch_ll)
    HIP_TRY(launch_ll(m, kern, src, n, out_dev, queue, nullptr, nullptr));
    if (in_place) {
        HIP_TRY(hipStreamSynchronize(nullptr));
        std::memcpy(out_ll, out_dev, (size_t)n * sizeof(double));
    } else {
        HIP_TRY(hipMemcpy(out_ll, out_dev, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    }
    return COVEST_OK;
}

int covest_reference_overflow(const covest_model *m, int64_t n, const double *params, uint8_t *flags)
{
    if (!m || n < 0 || (n > 0 && (!params || !flags)))
        return fail(COVEST_E_INVALID, "covest_reference_overflow: bad argument");
    const int P = m->n_par;
    for (int64_t i = 0; i < n; ++i) {
        const int T = P == 5 ? threshold_for_point(m, params + i * P) : 2;
        flags[i] = reference_overflows_at(m->dm, P, params + i * P, T, m->key_max) ? 1 : 0;
    }
    return COVEST_OK;
}

int covest_probabilities(covest_model *m, const double *params, int32_t clamp, double *out_p)
{
    if (!m || !params || (m->n_keys > 0 && !out_p))
        return fail(COVEST_E_INVALID, "covest_probabilities: bad argument");
    if (m->n_keys == 0)
        return COVEST_OK;
    std::lock_guard<std::mutex> guard(m->lock);
    DeviceGuard dev_guard(m->device);
    int rc = dev_guard.status();
    if (rc != COVEST_OK)
        return rc;
    const int P = m->n_par;
    HIP_TRY(m->ws_params.reserve((size_t)P * sizeof(double)));
    HIP_TRY(m->ws_out.reserve(sizeof(double)));
    HIP_TRY(m->ws_p.reserve((size_t)m->n_keys * sizeof(double)));
    HIP_TRY(hipMemcpy(m->ws_params.ptr, params, (size_t)P * sizeof(double), hipMemcpyHostToDevice));
    PointSource src{};
    src.params = m->ws_params.as<double>();
    if (P == 5) {
        const int32_t t = clamp ? threshold_for_point(m, params)
                                : threshold_o_host(params[2], params[3], params[4], m->threshold,
                                                   m->has_threshold, m->hist_max);
        HIP_TRY(m->ws_t.reserve(sizeof(int32_t)));
        HIP_TRY(hipMemcpy(m->ws_t.ptr, &t, sizeof(int32_t), hipMemcpyHostToDevice));
        src.t_list = m->ws_t.as<int32_t>();
    }
    if (!m->all_bins_ready) { // the view over EVERY key is only needed here: uploaded on first use
        rc = upload_bins(m->bins_all, m->all_bins, m->host_all_key, m->host_all_lgam, m->host_all_cnt);
        if (rc != COVEST_OK)
            return rc;
        m->all_bins_ready = true;
    }
    DevModel full = m->dm;
    full.bins = m->all_bins;
    if (!clamp)
        for (int i = 0; i < kMaxParams; ++i)
            full.lo[i] = full.hi[i] = std::numeric_limits<double>::quiet_NaN();
    HIP_TRY(launch_ll_direct(full, src, 1, m->ws_out.as<double>(), m->ws_p.as<double>(), nullptr));
    HIP_TRY(hipMemcpy(out_p, m->ws_p.ptr, (size_t)m->n_keys * sizeof(double), hipMemcpyDeviceToHost));
    return COVEST_OK;
}

// Everything a grid handle holds besides its identity: called by covest_grid_create and covest_grid_reset.  Device
// memory is only ever grown, and the small inputs (axes, threshold table, the queue's counter) go up in ONE copy:

} // extern "C"
