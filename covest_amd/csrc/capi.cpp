// capi.cpp -- the C ABI of include/covest_amd.h: host-side model preparation,
// threshold_o, device buffers and kernel dispatch.  Compiled with hipcc, links
// only the HIP runtime.  There is no CPU compute path in this library: every
// likelihood value comes out of a gfx950 kernel.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <limits>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/covest_amd.h"
#include "device_model.h"
#include "direct_point.h"
#include "kernels.h"

using namespace covest;

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string &msg)
{
    g_last_error = msg;
    return code;
}

} // namespace

namespace covest {
// for the library's other translation units (reads_io.cpp): record the message covest_last_error returns
int set_error(int code, const std::string &msg) { return fail(code, msg); }
} // namespace covest

namespace {

int fail_hip(hipError_t e, const char *what)
{
    g_last_error = std::string(what) + ": " + hipGetErrorString(e);
    return (e == hipErrorNoDevice || e == hipErrorInvalidDevice || e == hipErrorInsufficientDriver)
               ? COVEST_E_NO_DEVICE
               : COVEST_E_HIP;
}

#define HIP_TRY(expr)                                \
    do {                                             \
        hipError_t e__ = (expr);                     \
        if (e__ != hipSuccess)                       \
            return fail_hip(e__, #expr);             \
    } while (0)

// A device allocation that grows on demand and is released with its owner.
struct DevBuf {
    void *ptr = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes)
    {
        if (bytes <= cap)
            return hipSuccess;
        if (ptr)
            (void)hipFree(ptr);
        ptr = nullptr;
        cap = 0;
        hipError_t e = hipMalloc(&ptr, bytes);
        if (e == hipSuccess)
            cap = bytes;
        return e;
    }
    void release()
    {
        if (ptr)
            (void)hipFree(ptr);
        ptr = nullptr;
        cap = 0;
    }
    template <class T> T *as() const { return static_cast<T *>(ptr); }
};

// Page-locked host memory that is kept from call to call: what a call stages for the device goes through here.  (A
// pageable source makes hipMemcpy pin and unpin it, or bounce it, per call -- from a few hundred KB on that is most
// of a point-list evaluation's time, and how much depends on the machine: 64 points took 214 us on one box of the pool
// and 430 on another, 128 points 0.4 and 27 ms.)
struct HostBuf {
    void *ptr = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes)
    {
        if (bytes <= cap)
            return hipSuccess;
        if (ptr)
            (void)hipHostFree(ptr);
        ptr = nullptr;
        cap = 0;
        const size_t want = std::max<size_t>(bytes + bytes / 2, 1 << 16);
        hipError_t e = hipHostMalloc(&ptr, want, hipHostMallocPortable); // (any device of the process may copy from it)
        if (e == hipSuccess)
            cap = want;
        return e;
    }
    void release()
    {
        if (ptr)
            (void)hipHostFree(ptr);
        ptr = nullptr;
        cap = 0;
    }
    template <class T> T *as() const { return static_cast<T *>(ptr); }
};

// ONE page-locked staging buffer for the uploads that handles make when they are created or re-configured (bins, tile
// tables, axes, plans: tens to hundreds of KB, each copy blocking): kept for the life of the process, so that a
// handle's creation pays neither a pageable copy nor a page-locked allocation of its own.
struct SharedStage {
    std::mutex mu;
    HostBuf buf;
};
SharedStage &shared_stage()
{
    static SharedStage *s = new SharedStage; // (never freed: the runtime may be gone when statics are destroyed)
    return *s;
}

} // namespace

struct covest_model {
    int device = 0;
    int n_par = 2;
    DevModel dm{};       // bins = the evaluated view
    BinView all_bins{};  // every key, in dict order (compute_probabilities); uploaded on first use
    std::vector<double> host_all_key, host_all_lgam, host_all_cnt;
    bool all_bins_ready = false;
    int64_t n_keys = 0;
    int hist_max = 0;    // max(self.hist)
    int64_t key_max = 0; // the largest key the reference evaluates a pmf term for (0 if none is positive)
    // work accounting of K-factored over the item table (tiles.h): rows that are contracted (32 per item: a sum item
    // stands for up to 1024 keys) and keys that take a log
    double rows_contracted = 0.0, keys_logged = 0.0;
    bool tail_is_zero = true;
    double threshold = 0.0;
    bool has_threshold = true;
    // device storage of the two bin views
    DevBuf bins_eval, bins_all;
    // tile table of the pmf recurrence (fast kernels); has_tiles == false -> direct kernel only
    DevBuf tiles_buf;
    TileView tv{};
    bool has_tiles = false;
    // scratch for covest_eval_points / covest_probabilities
    DevBuf ws_params, ws_t, ws_out, ws_p, ws_plan, ws_plan2, ws_partial, ws_items;
    HostBuf ws_stage; // staging of a point list's tables (build_list_plan)
    DevBuf ws_sub_index, ws_sub_word, ws_sub_ctl; // the queue of handed-back points of a point-list launch (direct_point.h)
    std::mutex lock;
};

struct covest_grid {
    covest_model *model = nullptr;
    int64_t len[kMaxParams] = {1, 1, 1, 1, 1};
    int64_t flat_begin = 0, flat_end = 0;
    PointSource src{};
    // one allocation (grown on demand, kept across covest_grid_reset) behind the fixed-purpose views below
    DevBuf arena, plan_buf;
    struct View {
        void *ptr = nullptr;
        template <class T> T *as() const { return static_cast<T *>(ptr); }
        void release() { ptr = nullptr; }
    };
    View axes, t_table, ll, sub_index, sub_word, sub_ctl, partial_val, partial_idx, result;
    FactoredPlan plan{};        // K-factored work description (repeats model, dense grid): the weight vectors whose
                                //   threshold_o fits a workgroup's lanes (build_factored_plan)
    bool has_plan = false;
    bool has_short_part = false;
    // the weight vectors beyond that: one part per chunk of copy numbers, p_j summed in HBM, logs by ll_finish_dense
    struct Part {
        DevBuf buf;
        FactoredPlan plan{};
    };
    std::vector<Part> long_parts;
    int32_t n_long_tiles = 0;
    std::vector<int32_t> long_q_orig_host;
    DevBuf long_q_orig, long_partial;
    int t_max = 2; // largest threshold_o of the (q1, q2, q) product
    double q_sum_t_minus_1 = 0.0; // sum over the Q weight vectors of (threshold_o - 1)
    double contract_flops_per_row = 0.0; // K-factored: useful flops of the contraction per row (build_factored_plan)
    double sum_t_minus_1 = 0.0; // sum over the block's points of (threshold_o - 1)
    const char *last_kernel = "none";
    int last_kernel_id = 0;
    hipStream_t last_stream = nullptr;
    ArgminResult *result_host = nullptr; // page-locked mirror of `result`
    bool evaluated = false;
    bool configured = false; // false while (and after) a covest_grid_reset failed half way: the views may dangle
    // optional hipEvent bracketing of the likelihood kernel
    bool profiling = false;
    std::vector<hipEvent_t> ev_begin, ev_end;
    size_t ev_used = 0;
};

namespace {

// RepeatsModel.get_b_o / get_hist_threshold, covest/models.py:185-208, with libm
// pow as CPython's float ** int.  b_o is non-increasing in o for o >= 3 when
// 0 <= 1-q <= 1, so the first crossing is found by bisection and then confirmed
// against its left neighbours with the very same pow calls the linear scan of
// the reference would make; outside that domain the scan itself is used.
double weight_ge3(double head, double one_minus_q, int o)
{
    return head * std::pow(one_minus_q, (double)(o - 3));
}

int threshold_o_host(double q1, double q2, double q, double thr, bool has_thr, int hist_max)
{
    if (!has_thr)
        return hist_max;
    if (hist_max > 1 && q1 <= thr)
        return 1;
    if (hist_max > 2 && (1 - q1) * q2 <= thr)
        return 2;
    if (hist_max <= 3)
        return hist_max;
    const double head = (1 - q1) * (1 - q2) * q;
    const double base = 1 - q;
    const int last = hist_max - 1; // o ranges over 3..last
    if (!(base >= 0.0 && base <= 1.0) || !(head == head)) {
        for (int o = 3; o <= last; ++o)
            if (weight_ge3(head, base, o) <= thr)
                return o;
        return hist_max;
    }
    if (weight_ge3(head, base, 3) <= thr)
        return 3;
    if (!(weight_ge3(head, base, last) <= thr))
        return hist_max;
    int lo = 3, hi = last; // f(lo) > thr, f(hi) <= thr
    while (hi - lo > 1) {
        const int mid = lo + (hi - lo) / 2;
        if (weight_ge3(head, base, mid) <= thr)
            hi = mid;
        else
            lo = mid;
    }
    while (hi > 3 && weight_ge3(head, base, hi - 1) <= thr)
        --hi;
    return hi;
}

double clamp_one(const DevModel &dm, int d, double v)
{
    const double lo = dm.lo[d], hi = dm.hi[d];
    if (lo == lo && v < lo)
        return lo;
    if (hi == hi && v > hi)
        return hi;
    return v;
}

int threshold_for_point(const covest_model *m, const double *par)
{
    return threshold_o_host(clamp_one(m->dm, 2, par[2]), clamp_one(m->dm, 3, par[3]),
                            clamp_one(m->dm, 4, par[4]), m->threshold, m->has_threshold, m->hist_max);
}

// Every entry point works on ITS handle's device and leaves the calling thread's current device as it found
// it: the caller (torch, another library, a rank bound to another GPU) never sees its device change underneath.
class DeviceGuard {
  public:
    explicit DeviceGuard(int device)
    {
        // (HIP keeps the last error of the thread until somebody reads it: whatever an earlier, unrelated call left
        // behind must not be taken for a failure of the launches this entry point is about to make)
        (void)hipGetLastError();
        had_prev_ = hipGetDevice(&prev_) == hipSuccess;
        if (had_prev_ && prev_ == device)
            return; // nothing to switch, nothing to restore
        const hipError_t e = hipSetDevice(device);
        if (e != hipSuccess)
            status_ = fail_hip(e, "hipSetDevice");
        else
            switched_ = true;
    }
    ~DeviceGuard()
    {
        if (switched_ && had_prev_)
            (void)hipSetDevice(prev_);
    }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
    int status() const { return status_; }

  private:
    int prev_ = 0;
    bool had_prev_ = false, switched_ = false;
    int status_ = COVEST_OK;
};

// device < 0 = the calling thread's current device (include/covest_amd.h); checked against the device count.
int resolve_device(int device, const char *who, int *out)
{
    if (device < 0) {
        hipError_t e = hipGetDevice(&device);
        if (e != hipSuccess)
            return fail_hip(e, "hipGetDevice");
    }
    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess)
        return fail_hip(e, "hipGetDeviceCount");
    if (n_dev <= 0 || device >= n_dev)
        return fail(COVEST_E_NO_DEVICE, std::string(who) + ": no such HIP device");
    *out = device;
    return COVEST_OK;
}

// ln j! = lgamma(j + 1) rounded from long double, for j = 0, 1, 2, ...: a process-wide table grown on demand
// (lgammal costs ~100 ns; a 10 000-key histogram paid 1 ms of it per model handle).  A deque: growing it never moves
// the entries already there, so a caller that has made sure of the first n (lgamma_ensure, under the lock) may read
// them without it (lgamma_at) -- one lock per histogram instead of one per key.
std::mutex g_lgamma_lock;
std::deque<double> g_lgamma_table;
constexpr int64_t kLgammaTableMax = (int64_t)1 << 22; // beyond any histogram the fast paths accept: not cached

void lgamma_ensure(int64_t j_max)
{
    if (j_max > kLgammaTableMax)
        j_max = kLgammaTableMax;
    std::lock_guard<std::mutex> guard(g_lgamma_lock);
    for (size_t v = g_lgamma_table.size(); v <= (size_t)std::max<int64_t>(j_max, 0); ++v)
        g_lgamma_table.push_back((double)lgammal((long double)v + 1.0L));
}

// (after lgamma_ensure(j) or larger)
inline double lgamma_at(int64_t j)
{
    if (j < 0)
        j = 0;
    if (j > kLgammaTableMax)
        return (double)lgammal((long double)j + 1.0L);
    return g_lgamma_table[(size_t)j];
}

double lgamma_of_factorial(int64_t j)
{
    lgamma_ensure(j);
    return lgamma_at(j);
}

// One buffer, one copy: [key | lgam | cnt].
int upload_bins(DevBuf &buf, BinView &view, const std::vector<double> &key,
                const std::vector<double> &lgam, const std::vector<double> &cnt)
{
    const size_t n = key.size();
    view.n = (int64_t)n;
    view.key = view.lgam = view.cnt = nullptr;
    if (n == 0)
        return COVEST_OK;
    HIP_TRY(buf.reserve(3 * n * sizeof(double)));
    double *base = buf.as<double>();
    {
        SharedStage &ss = shared_stage();
        std::lock_guard<std::mutex> hold(ss.mu);
        HIP_TRY(ss.buf.reserve(3 * n * sizeof(double)));
        double *stage = ss.buf.as<double>();
        std::copy(key.begin(), key.end(), stage);
        std::copy(lgam.begin(), lgam.end(), stage + n);
        std::copy(cnt.begin(), cnt.end(), stage + 2 * n);
        HIP_TRY(hipMemcpy(base, stage, 3 * n * sizeof(double), hipMemcpyHostToDevice));
    }
    view.key = base;
    view.lgam = base + n;
    view.cnt = base + 2 * n;
    return COVEST_OK;
}

// Tile table of streams.h over the evaluated bins: keys sorted ascending, split
// into runs of consecutive keys (gaps of up to kGapFill keys are bridged with
// filler keys that are stepped over but neither logged nor summed), each run cut
// into tiles of <= 32 keys.  Returns false when the fast kernels do not apply.
constexpr int64_t kListModeMaxPoints = 4096; // longer point lists are throughput work: K-direct
constexpr int kGapFill = 12;
constexpr int kMaxFastKey = 16384;

struct HostBin {
    int key;
    double cnt;
    int32_t index; // in the evaluated bin view (DevModel::bins)
};

int build_tiles(covest_model *m, std::vector<HostBin> bins)
{
    m->has_tiles = false;
    if (m->dm.n_err > 32 || bins.empty()) // (the recurrence kernels hold max_error <= 32 error classes)
        return COVEST_OK;
    std::sort(bins.begin(), bins.end(), [](const HostBin &a, const HostBin &b) { return a.key < b.key; });
    if (bins.front().key < 1 || bins.back().key > kMaxFastKey)
        return COVEST_OK;
    struct Tile {
        int k0, nb, run_start;
    };
    std::vector<Tile> tiles;
    std::vector<double> scal, cnt;
    std::vector<int32_t> row_bin;
    size_t i = 0;
    while (i < bins.size()) {
        // one run: keys bins[i..j) with gaps <= kGapFill
        size_t j = i + 1;
        while (j < bins.size() && bins[j].key - bins[j - 1].key <= kGapFill + 1)
            ++j;
        const int first = bins[i].key, last = bins[j - 1].key;
        size_t cur = i;
        for (int k0 = first; k0 <= last; k0 += kTileBins) {
            const int nb = std::min(kTileBins, last - k0 + 1);
            tiles.push_back({k0, nb, k0 == first ? 1 : 0});
            long double sc = ldexpl(1.0L, -kScaleBits);
            for (int b = 0; b < kTileBins; ++b) {
                double sv = 0.0, cv = 0.0;
                int32_t which = -1;
                if (b < nb) {
                    const int key = k0 + b;
                    sc /= (long double)key;
                    if (cur < j && bins[cur].key == key) {
                        sv = (double)sc;
                        cv = bins[cur].cnt;
                        which = bins[cur].index;
                        ++cur;
                    } // else a FILLER key (a gap of the histogram the recurrence walks through): scale 0, so
                      // that its p_j is exactly 0 -- it is no key of the reference's p_j dict, and must add
                      // nothing to sp_j (covest/models.py:103) and take no log
                }
                scal.push_back(sv);
                cnt.push_back(cv);
                row_bin.push_back(which);
            }
        }
        i = j;
    }
    const size_t nt = tiles.size();
    // layout: [first_key | lgam_prev | lgam_last | renorm] doubles, then scal/cnt, then int32 n_bins/run_start
    std::vector<double> dbl(4 * nt);
    std::vector<int32_t> ints(2 * nt);
    for (size_t t = 0; t < nt; ++t) {
        const Tile &tl = tiles[t];
        dbl[t] = (double)tl.k0;
        dbl[nt + t] = lgamma_of_factorial((int64_t)tl.k0 - 1);
        dbl[2 * nt + t] = lgamma_of_factorial((int64_t)(tl.k0 + tl.nb) - 1);
        long double rn = 1.0L;
        for (int b = 0; b < tl.nb; ++b)
            rn /= (long double)(tl.k0 + b);
        dbl[3 * nt + t] = (double)rn;
        ints[t] = tl.nb;
        ints[nt + t] = tl.run_start;
    }
    // items (tiles.h): runs of all-zero-count tiles (they exist only with a tail) are grouped, up to 32 per item
    std::vector<int32_t> item_first, item_ntiles, item_sum;
    std::vector<double> item_cnt;
#ifdef COVEST_DIAG
    const bool no_sum_items = std::getenv("COVEST_NO_SUM_ITEMS") != nullptr; // diagnostic builds: every tile a plain item
#else
    const bool no_sum_items = false;
#endif
    for (size_t t = 0; t < nt;) {
        auto all_zero = [&](size_t tt) {
            for (int b = 0; b < kTileBins; ++b)
                if (cnt[tt * kTileBins + (size_t)b] != 0.0)
                    return false;
            return true;
        };
        if (!all_zero(t) || no_sum_items) {
            item_first.push_back((int32_t)t);
            item_ntiles.push_back(1);
            item_sum.push_back(0);
            item_cnt.insert(item_cnt.end(), cnt.begin() + (std::ptrdiff_t)(t * kTileBins),
                            cnt.begin() + (std::ptrdiff_t)((t + 1) * kTileBins));
            ++t;
            continue;
        }
        size_t e = t + 1;
        while (e < nt && e - t < (size_t)kTileBins && all_zero(e))
            ++e;
        item_first.push_back((int32_t)t);
        item_ntiles.push_back((int32_t)(e - t));
        item_sum.push_back(1);
        item_cnt.insert(item_cnt.end(), (size_t)kTileBins, 0.0);
        t = e;
    }
    const size_t ni = item_first.size();
    m->rows_contracted = (double)ni * kTileBins;
    m->keys_logged = 0.0;
    for (double c : cnt)
        m->keys_logged += c != 0.0 ? 1.0 : 0.0;
    // K-factored's view of the rows (tiles.h): the scale of a plain item's rows as a factor (and its reciprocal, for
    // the clamp in the row's units) and as the constant it adds to the item's sum of h_j log p_j
    std::vector<double> item_scal(ni * kTileBins, 0.0), item_iscal(ni * kTileBins, 0.0), item_lconst(ni, 0.0);
    for (size_t i2 = 0; i2 < ni; ++i2) {
        if (item_sum[i2]) {
            for (int b = 0; b < kTileBins; ++b)
                item_scal[i2 * kTileBins + (size_t)b] = 1.0;
            continue;
        }
        const size_t t = (size_t)item_first[i2];
        long double lc = 0.0L, lratio = 0.0L; // ln((k0-1)!/(k0+b)!) = -sum_{i=k0}^{k0+b} ln i
        for (int b = 0; b < tiles[t].nb; ++b) {
            lratio -= logl((long double)(tiles[t].k0 + b));
            const double sv = scal[t * kTileBins + (size_t)b];
            if (sv == 0.0)
                continue; // filler key
            item_scal[i2 * kTileBins + (size_t)b] = sv;
            if (cnt[t * kTileBins + (size_t)b] != 0.0) // (a row without a count takes no log and is never "low": 0)
                item_iscal[i2 * kTileBins + (size_t)b] = 1.0 / sv;
            lc += (long double)cnt[t * kTileBins + (size_t)b] * lratio;
        }
        item_lconst[i2] = (double)lc;
    }
    // K-basic's closed form (ll_basic.hip): suffix sums over the counted keys of the tiles t .. nt - 1
    std::vector<double> suf(5 * (nt + 1) + 2, 0.0);
    {
        long double s_h = 0.0L, s_jh = 0.0L, s_lgh = 0.0L;
        double first_key = 0.0, first_lg = 0.0;
        for (size_t t = nt; t-- > 0;) {
            for (int b = tiles[t].nb - 1; b >= 0; --b) {
                const double h = cnt[t * kTileBins + (size_t)b];
                if (h == 0.0)
                    continue;
                const int key = tiles[t].k0 + b;
                const double lg = lgamma_at(key);
                s_h += (long double)h;
                s_jh += (long double)h * (long double)key;
                s_lgh += (long double)h * (long double)lg;
                first_key = (double)key;
                first_lg = lg;
                if (suf[5 * (nt + 1)] == 0.0) { // the first one met from the end: the last counted key
                    suf[5 * (nt + 1)] = (double)key;
                    suf[5 * (nt + 1) + 1] = lg;
                }
            }
            suf[t] = (double)s_h;
            suf[(nt + 1) + t] = (double)s_jh;
            suf[2 * (nt + 1) + t] = (double)s_lgh;
            suf[3 * (nt + 1) + t] = first_key;
            suf[4 * (nt + 1) + t] = first_lg;
        }
    }
    const size_t n_dbl = 4 * nt + 2 * nt * kTileBins + 3 * ni * kTileBins + ni + suf.size();
    std::vector<int32_t> tile_zero(nt, 0);
    for (size_t i2 = 0; i2 < ni; ++i2)
        if (item_sum[i2])
            for (int32_t r = 0; r < item_ntiles[i2]; ++r)
                tile_zero[(size_t)(item_first[i2] + r)] = 1;
    const size_t bytes = n_dbl * sizeof(double) + (3 * nt + 3 * ni + nt * kTileBins) * sizeof(int32_t);
    HIP_TRY(m->tiles_buf.reserve(bytes));
    double *base = m->tiles_buf.as<double>();
    int32_t *ibase = reinterpret_cast<int32_t *>(base + n_dbl);
    SharedStage &ss = shared_stage();
    std::lock_guard<std::mutex> hold(ss.mu);
    HIP_TRY(ss.buf.reserve(bytes)); // one copy instead of five
    char *const stage = ss.buf.as<char>();
    char *sp = stage;
    auto put = [&](const void *src, size_t n) {
        std::memcpy(sp, src, n);
        sp += n;
    };
    put(dbl.data(), 4 * nt * sizeof(double));
    put(scal.data(), nt * kTileBins * sizeof(double));
    put(cnt.data(), nt * kTileBins * sizeof(double));
    put(item_cnt.data(), ni * kTileBins * sizeof(double));
    put(item_scal.data(), ni * kTileBins * sizeof(double));
    put(item_iscal.data(), ni * kTileBins * sizeof(double));
    put(item_lconst.data(), ni * sizeof(double));
    put(suf.data(), suf.size() * sizeof(double));
    put(ints.data(), 2 * nt * sizeof(int32_t));
    put(tile_zero.data(), nt * sizeof(int32_t));
    put(item_first.data(), ni * sizeof(int32_t));
    put(item_ntiles.data(), ni * sizeof(int32_t));
    put(item_sum.data(), ni * sizeof(int32_t));
    put(row_bin.data(), nt * kTileBins * sizeof(int32_t));
    HIP_TRY(hipMemcpy(base, stage, bytes, hipMemcpyHostToDevice));
    m->tv = tile_view_from((int32_t)nt, (int32_t)ni, base, ibase);
    m->has_tiles = true;
    return COVEST_OK;
}

// p_clamp of direct_point.h for a launch whose largest threshold_o is t_max.
double clamp_for(const covest_model *m, int t_max)
{
    return (double)(m->dm.n_err + std::max(t_max, 2)) * kClampPerTerm;
}

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
        r4[ls] = std::pow(1 - q, 4.0);
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
    const int want_blocks = n_ce_grid >= 192 ? 1 : (int)std::min<int64_t>(n_qtiles, (256 + n_ce_grid - 1) / n_ce_grid);
    const int n_qblocks = std::max(std::max(1, (n_units + cap_block - 1) / cap_block), want_blocks);
    // cost model of the assignment, in MFMA steps: a unit costs its steps (in every pass) plus its share of the
    // logs; a builder wave starts with the cost of phase A (tuned on C3 with the in-kernel stamps)
    // (the assignment fixes the order of a point's sums: the shipped library takes the constants of tiles.h, only a
    // diagnostic build -- tiles.h -- lets the environment override them for tuning sweeps)
    int unit_overhead = kUnitOverhead, build_cost = kBuildCost, shared_div = kSharedStepsPerMfma;
    int last_builder_extra = kLastBuilderExtra;
    // (with a tail an item may stand for up to 32 count-less tiles, tiles.h: the builders walk every one of them
    // while the contraction sees one item -- charge them for the tiles an item holds on average)
    if (m->has_tiles && m->tv.n_items > 0)
        build_cost = (int)std::lround((double)kBuildCost * (double)m->tv.n_tiles / (double)m->tv.n_items);
#ifdef COVEST_DIAG
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
        for (int colx = 0; colx < 16; ++colx) {
            const size_t gs = ((size_t)tile_lo + (size_t)qt) * 16 + (size_t)colx;
            if (qo.order[gs] < 0)
                continue; // padding column
            double q1, q2, q;
            weights_of(gs, q1, q2, q);
            const int o_first = o_base + unit_o0[at];
            const double head = (1 - q1) * (1 - q2) * q, base = 1 - q;
            double geo = o_first >= 3 ? std::pow(base, (double)(o_first - 3)) : 1.0; // base^(o - 3) at o = max(o_first, 3)
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
                unit_rho[4 * at] = std::pow(base, 16.0);
                unit_rho[4 * at + 2] = std::pow(base, 4.0);
                unit_rho[4 * at + 3] = 1.0 / std::pow(base, 4.0 * (double)unit_nsh[at]);
            }
            if (unit_nsh[at] > 0) { // the first step after the shared ones: o = 5 + 4 nsh .. 8 + 4 nsh
                const int o_after = o_first + 4 * (unit_nsh[at] + 1);
                double g2 = std::pow(base, (double)(o_after - 3));
                for (int d = 0; d < 4; ++d, g2 *= base)
                    piece_w[(at * 64 + (size_t)(d * 16 + colx)) * 2 + 1] = head * g2;
            }
        }
    }
    // one buffer: doubles first (r4 | piece_w), then int32 (q_T | q_orig | unit tables)
    const size_t n_dbl = n_slots + piece_w.size() + unit_rho.size();
    const size_t n_int = 2 * n_slots + 7 * n_unit;
    HIP_TRY(buf.reserve(n_dbl * sizeof(double) + n_int * sizeof(int32_t)));
    double *dbase = buf.as<double>();
    int32_t *ibase = reinterpret_cast<int32_t *>(dbase + n_dbl);
    {
        // staged on the host in the device layout, ONE copy (a dozen small copies cost ~150 us of the plan build)
        const size_t stage_bytes = n_dbl * sizeof(double) + n_int * sizeof(int32_t);
        SharedStage &ss = shared_stage();
        std::lock_guard<std::mutex> hold(ss.mu);
        HIP_TRY(ss.buf.reserve(stage_bytes));
        double *sd = ss.buf.as<double>();
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
        HIP_TRY(hipMemcpy(buf.ptr, sd, stage_bytes, hipMemcpyHostToDevice));
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
                std::fprintf(stderr, "COVEST_FACTORED_DIAG %p %zu\n", dp, bytes);
            }
        }
    }
#endif
    return COVEST_OK;
}

// All the parts of a dense repeats grid (see above).  g->has_plan stays false where K-factored does not apply:
// no tile table (keys beyond 16384 ...), more than 32 error classes, or more weight vectors than 2^24.
int build_factored_plan(covest_grid *g, const double *const *axes, const int64_t *axis_len,
                        const std::vector<int32_t> &t_table)
{
    covest_model *m = g->model;
    g->has_plan = false;
    for (covest_grid::Part &part : g->long_parts)
        part.buf.release();
    g->long_parts.clear();
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
#ifdef COVEST_DIAG
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
int build_list_plan(covest_model *m, int64_t n, const double *params, const std::vector<int32_t> &t_list,
                    const std::vector<int32_t> *o_base_list, DevBuf &buf, FactoredPlan &pl)
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
        unit_s0(n_unit, 0), unit_o0(n_unit, 1), unit_len(n_unit, 0), unit_cont(n_unit, 0);
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
        {unit_cont.data(), unit_cont.size()}, {unit_o0.data(), unit_o0.size()}};
    size_t n_dbl = 0, n_int = 0;
    for (auto &pr : dparts)
        n_dbl += pr.second;
    for (auto &pr : iparts)
        n_int += pr.second;
    HIP_TRY(buf.reserve(n_dbl * sizeof(double) + n_int * sizeof(int32_t)));
    // one staging buffer (page-locked, the model's), one copy
    const size_t stage_bytes = n_dbl * sizeof(double) + n_int * sizeof(int32_t);
    HIP_TRY(m->ws_stage.reserve(stage_bytes));
    char *stage = m->ws_stage.as<char>();
    double *dbase = buf.as<double>();
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
    HIP_TRY(hipMemcpy(buf.ptr, stage, stage_bytes, hipMemcpyHostToDevice));
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

} // namespace

extern "C" {

int covest_abi_version(void) { return COVEST_ABI_VERSION; }

const char *covest_last_error(void) { return g_last_error.c_str(); }

int covest_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess)
        return fail_hip(e, "hipGetDeviceCount");
    return n;
}

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
    m->bins_eval.release();
    m->bins_all.release();
    m->tiles_buf.release();
    m->ws_params.release();
    m->ws_t.release();
    m->ws_out.release();
    m->ws_p.release();
    m->ws_plan.release();
    m->ws_plan2.release();
    m->ws_stage.release();
    m->ws_partial.release();
    m->ws_items.release();
    m->ws_sub_index.release();
    m->ws_sub_word.release();
    m->ws_sub_ctl.release();
    delete m;
}

int covest_model_param_count(const covest_model *m) { return m ? m->n_par : COVEST_E_INVALID; }

int64_t covest_model_bins_evaluated(const covest_model *m) { return m ? m->dm.bins.n : COVEST_E_INVALID; }

int covest_threshold_o(int64_t n, const double *q123, double threshold, int32_t has_threshold,
                       int32_t hist_max, int32_t *out)
{
    if (n < 0 || (n > 0 && (!q123 || !out)))
        return fail(COVEST_E_INVALID, "covest_threshold_o: bad argument");
    for (int64_t i = 0; i < n; ++i)
        out[i] = threshold_o_host(q123[3 * i], q123[3 * i + 1], q123[3 * i + 2], threshold,
                                  has_threshold != 0, hist_max);
    return COVEST_OK;
}

// ---- where the REFERENCE overflows (documented divergence) ----
// c_src/covest_poissonmodule.c:19-24 forms the whole product prod_{i<=j} (l / i) in x87 long double BEFORE any
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

// Resolve COVEST_KERNEL_* for a request (g == nullptr: a point list).  Returns the
// kernel to run or a negative error.
static int resolve_kernel(const covest_model *m, int32_t kernel, const covest_grid *g)
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
        e = launch_ll_fix_list(m->dm, m->tv, g->src, out, sub, st);
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

static SubList sub_list_of(const covest_model *m, int t_max, void *index, void *word, void *ctl)
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
static hipError_t launch_ll(const covest_model *m, int kernel, const PointSource &src, int64_t n,
                            double *out, const SubList &sub, hipStream_t st, const char **name,
                            const covest_grid *g = nullptr)
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
        return e != hipSuccess ? e : launch_ll_fix_list(m->dm, m->tv, src, out, sub, st);
    }
    if (name)
        *name = kernel == COVEST_KERNEL_DIRECT_REF ? "ll_direct_ref" : "ll_direct";
    return launch_ll_direct(m->dm, src, n, out, nullptr, st, kernel == COVEST_KERNEL_DIRECT_REF);
}

// Workspace of a point-list launch's queue (direct_point.h): room for n entries, counters zeroed on first use.
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
                               sub_list_of(m, m->n_par == 5 ? 513 : 2, m->ws_sub_index.ptr, m->ws_sub_word.ptr, m->ws_sub_ctl.ptr), nullptr));
    HIP_TRY(hipMemcpy(sub_ll.data(), m->ws_out.ptr, na * sizeof(double), hipMemcpyDeviceToHost));
    for (size_t k = 0; k < na; ++k)
        out_ll[again[k]] = sub_ll[k];
    return COVEST_OK;
}

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
    HIP_TRY(m->ws_params.reserve((size_t)n * P * sizeof(double)));
    HIP_TRY(m->ws_out.reserve((size_t)n * sizeof(double)));
    HIP_TRY(m->ws_t.reserve((size_t)n * sizeof(int32_t)));
    {
        const int qrc = reserve_point_queue(m, n);
        if (qrc != COVEST_OK)
            return qrc;
    }
    const SubList queue = sub_list_of(m, m->n_par == 5 ? 513 : 2, m->ws_sub_index.ptr, m->ws_sub_word.ptr, m->ws_sub_ctl.ptr);
    PointSource src{};
    src.is_grid = 0;
    src.params = m->ws_params.as<double>();
    src.t_list = P == 5 ? m->ws_t.as<int32_t>() : nullptr;
    if (kern != COVEST_KERNEL_FACTORED) // (the list mode uploads its own tables: every copy is ~10 us of latency)
        HIP_TRY(hipMemcpy(m->ws_params.ptr, params, (size_t)n * P * sizeof(double), hipMemcpyHostToDevice));
    if (P == 5 && kern != COVEST_KERNEL_FACTORED) {
        std::vector<int32_t> t((size_t)n);
        for (int64_t i = 0; i < n; ++i)
            t[(size_t)i] = threshold_for_point(m, params + i * P);
        HIP_TRY(m->ws_t.reserve((size_t)n * sizeof(int32_t)));
        HIP_TRY(hipMemcpy(m->ws_t.ptr, t.data(), (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice));
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
            rc = build_list_plan(m, (int64_t)fits.size(), sub_par.data(), sub_t, nullptr, m->ws_plan, pl);
            if (rc != COVEST_OK)
                return rc;
            // {LL part, sp_j part (hi, lo), side word} per (point, key segment); the segments are added here, in order
            const size_t n_parts = fits.size() * (size_t)pl.n_seg;
            HIP_TRY(m->ws_partial.reserve(n_parts * 4 * sizeof(double)));
            pl.partial = m->ws_partial.as<double>();
            HIP_TRY(launch_ll_factored(m->dm, m->tv, pl, m->ws_out.as<double>(), queue, nullptr));
            std::vector<double> got(n_parts * 4);
            HIP_TRY(hipMemcpy(got.data(), m->ws_partial.ptr, got.size() * sizeof(double), hipMemcpyDeviceToHost));
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
    HIP_TRY(launch_ll(m, kern, src, n, m->ws_out.as<double>(), queue, nullptr, nullptr));
    HIP_TRY(hipMemcpy(out_ll, m->ws_out.ptr, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
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
    g->ev_used = 0;
    const int64_t n = flat_end - flat_begin;
    const int64_t n1 = n_axes == 5 ? axis_len[2] : 1, n2 = n_axes == 5 ? axis_len[3] : 1, n3 = n_axes == 5 ? axis_len[4] : 1;
    const int64_t nq = n_axes == 5 ? n1 * n2 * n3 : 0;

    // threshold_o over the (q1, q2, q) sub-grid (host, libm)
    std::vector<int32_t> table((size_t)nq);
    for (int64_t a = 0; a < n1 && nq; ++a)
        for (int64_t b = 0; b < n2; ++b)
            for (int64_t c = 0; c < n3; ++c) {
                const double par[5] = {0, 0, axes[2][a], axes[3][b], axes[4][c]};
                table[(size_t)((a * n2 + b) * n3 + c)] = threshold_for_point(m, par);
            }

    // arena layout: [axes | queue counter (8 B) | t_table] uploaded together, then the outputs
    auto up8 = [](size_t v) { return (v + 7) / 8 * 8; };
    const size_t o_axes = 0, o_ctl = o_axes + (size_t)n_values * sizeof(double), o_table = o_ctl + 8;
    const size_t o_ll = up8(o_table + (size_t)nq * sizeof(int32_t)), n_pts = (size_t)(n > 0 ? n : 1);
    const size_t o_idx = o_ll + n_pts * sizeof(double), o_word = o_idx + n_pts * sizeof(int64_t);
    const size_t o_pv = o_word + n_pts * sizeof(unsigned long long), o_pi = o_pv + kArgminBlocks * sizeof(double);
    const size_t o_res = o_pi + kArgminBlocks * sizeof(int64_t), bytes = o_res + sizeof(ArgminResult);
    HIP_TRY(g->arena.reserve(bytes));
    char *base = g->arena.as<char>();
    {
        SharedStage &ss = shared_stage();
        std::lock_guard<std::mutex> hold(ss.mu);
        HIP_TRY(ss.buf.reserve(o_ll));
        char *stage = ss.buf.as<char>();
        std::memset(stage, 0, o_ll);
        double *sa = reinterpret_cast<double *>(stage);
        for (int d = 0; d < n_axes; ++d) {
            g->len[d] = axis_len[d];
            std::copy(axes[d], axes[d] + axis_len[d], sa);
            sa += axis_len[d];
        }
        for (int d = n_axes; d < kMaxParams; ++d)
            g->len[d] = 1;
        if (nq)
            std::memcpy(stage + o_table, table.data(), (size_t)nq * sizeof(int32_t));
        HIP_TRY(hipMemcpy(base, stage, o_ll, hipMemcpyHostToDevice));
    }
    g->axes.ptr = base + o_axes;
    g->sub_ctl.ptr = base + o_ctl;
    g->t_table.ptr = base + o_table;
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
    g->configured = true;
    return COVEST_OK;
}

static void grid_release(covest_grid *g)
{
    if (g->result_host) {
        (void)hipHostFree(g->result_host);
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

int covest_grid_eval(covest_grid *g, int32_t kernel, void *stream)
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
        }
    }
#endif
    if (!g->result_host) // (page-locked, mapped: argmin_stage2 stores the winner there itself)
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&g->result_host), sizeof(ArgminResult), hipHostMallocMapped));
    HIP_TRY(launch_argmin(g->ll.as<double>(), n, g->flat_begin, g->partial_val.as<double>(),
                          g->partial_idx.as<int64_t>(), g->result.as<ArgminResult>(), g->result_host, g->sub_ctl.as<unsigned>(), st));
    g->last_stream = st;
    g->evaluated = true;
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

// ------------------------------------------------------------------ k-mer counter
} // extern "C" (reopened below: the handle type needs C++ members)

struct covest_kmer {
    int device = 0;
    int k = 20;
    int canonical = 0;
    int wide = 0;          // 0: k <= 31, `table`; else the words per key of `wtable` (kmer_wide.hip)
    KmerWideTable wtable{};
    KmerTable table{};
    DevBuf slots, flag, stats, hist, ws_bases, ws_offsets;
    // the partitioned path (kmer_bulk.hip): its buffers, kept from call to call, and what it found
    bool bulk = false; // the counter holds the result of covest_kmer_count_reads_device (until covest_kmer_clear)
    DevBuf bulk_sampled, bulk_cursor, bulk_fill, bulk_tile_reads, bulk_later, bulk_partial, bulk_recs, bulk_ovf, bulk_ctl, bulk_hist, bulk_big;
    unsigned long long bulk_stats[4] = {0, 0, 0, 0};
    int64_t bulk_info[5] = {0, 0, 0, 0, 0}; // buckets, m, sample, records there was room for, records that found none
    unsigned bulk_later_n = 0;              // buckets a workgroup (not a wave) counted
    unsigned long long bulk_to_table_n = 0; // buckets counted through the table in HBM
    bool bulk_table_used = false;           // ... and whether the table holds anything of the result
    hipEvent_t bulk_ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr}; // start, placed, scattered, counted, done
    float bulk_ms[4] = {0, 0, 0, 0};
    std::mutex lock;
};

constexpr unsigned long long kBulkHistLen = 1ull << 20; // dense count-of-counts bins of the partitioned path
constexpr unsigned long long kBulkBigCap = 4096;        // counts beyond them, listed one by one

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
    c->slots.release();
    c->flag.release();
    c->stats.release();
    c->hist.release();
    c->ws_bases.release();
    c->ws_offsets.release();
    for (hipEvent_t &e : c->bulk_ev)
        if (e) {
            (void)hipEventDestroy(e);
            e = nullptr;
        }
    c->bulk_sampled.release();
    c->bulk_fill.release();
    c->bulk_tile_reads.release();
    c->bulk_later.release();
    c->bulk_partial.release();
    c->bulk_cursor.release();
    c->bulk_recs.release();
    c->bulk_ovf.release();
    c->bulk_ctl.release();
    c->bulk_hist.release();
    c->bulk_big.release();
    delete c;
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
    if (c->bulk) {
        // what the partitioned path kept for its next call -- the buckets' records are gigabytes -- goes with the counts
        // (covest_kmer_count_reads_device has returned: nothing of it is in flight)
        c->bulk_recs.release();
        c->bulk_ovf.release();
        c->bulk = false;
    }
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
    if (rc != COVEST_OK) {
        slots.release(); // (DevBuf has no destructor: every error path lets the new table go)
        return rc;
    }
    c->slots.release();
    c->slots = slots;
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
        if (((double)room + (double)p.overflow_cap * kOvfShards) * 16.0 > 0.85 * (double)(free_b + have))
            return fail(COVEST_E_NOMEM, "covest_kmer_count_reads_device: the buckets do not fit the free device memory");
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
            if (rc != COVEST_OK) {
                slots.release();
                return rc;
            }
            c->slots.release();
            c->slots = slots;
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
    words.release();
    if (e != hipSuccess)
        return fail_hip(e, "covest_kmer_scatter_rate");
    const double done = (double)(((ops + 63) / 64) * 64);
    *ops_per_s = done / ((double)ms * 1e-3);
    return COVEST_OK;
}

static int thin_histogram_impl(int32_t device, int64_t n, const int32_t *keys, const double *counts, double factor,
                               int64_t out_len, double *out, int32_t repeats, double *kernel_ms)
{
    if (n < 0 || out_len < 0 || (n > 0 && (!keys || !counts)) || (out_len > 0 && !out))
        return fail(COVEST_E_INVALID, "covest_thin_histogram: null argument");
    if (!(factor > 1.0))
        return fail(COVEST_E_INVALID, "covest_thin_histogram: factor must be > 1");
    int32_t max_key = 0;
    for (int64_t s = 0; s < n; ++s) {
        if (keys[s] < 1)
            return fail(COVEST_E_INVALID, "covest_thin_histogram: keys must be >= 1");
        max_key = std::max(max_key, keys[s]);
    }
    if (kernel_ms)
        *kernel_ms = 0.0;
    if (out_len == 0)
        return COVEST_OK;
    {
        const int drc = resolve_device(device, "covest_thin_histogram", &device);
        if (drc != COVEST_OK)
            return drc;
    }
    DeviceGuard dev_guard(device);
    if (dev_guard.status() != COVEST_OK)
        return dev_guard.status();
    const int64_t top = std::max<int64_t>(max_key, out_len);
    std::vector<double> lgam((size_t)top + 1);
    for (int64_t v = 0; v <= top; ++v)
        lgam[(size_t)v] = std::lgamma((double)v + 1.0);
    std::vector<ThinSource> src((size_t)std::max<int64_t>(n, 1));
    for (int64_t s = 0; s < n; ++s) {
        ThinSource &e = src[(size_t)s];
        e.i = keys[s];
        e.pad = 0;
        e.count = counts[s];
        const double l = (double)keys[s] * (1.0 / factor); // `i * prob`, covest/histogram.py:64
        e.a = keys[s] < 100 ? lgam[(size_t)keys[s]] : std::log(l);
        e.b = l;
    }
    DevBuf d_src, d_lgam, d_partial, d_out;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    auto done = [&](int code) {
        d_src.release();
        d_lgam.release();
        d_partial.release();
        d_out.release();
        if (e0)
            (void)hipEventDestroy(e0);
        if (e1)
            (void)hipEventDestroy(e1);
        return code;
    };
#define THIN_TRY(expr)                          \
    do {                                        \
        hipError_t e__ = (expr);                \
        if (e__ != hipSuccess)                  \
            return done(fail_hip(e__, #expr));  \
    } while (0)
    THIN_TRY(d_src.reserve(src.size() * sizeof(ThinSource)));
    THIN_TRY(d_lgam.reserve(lgam.size() * sizeof(double)));
    THIN_TRY(d_partial.reserve((size_t)thin_hist_chunks() * (size_t)out_len * sizeof(double)));
    THIN_TRY(d_out.reserve((size_t)out_len * sizeof(double)));
    THIN_TRY(hipMemcpy(d_src.ptr, src.data(), src.size() * sizeof(ThinSource), hipMemcpyHostToDevice));
    THIN_TRY(hipMemcpy(d_lgam.ptr, lgam.data(), lgam.size() * sizeof(double), hipMemcpyHostToDevice));
    if (kernel_ms) {
        THIN_TRY(hipEventCreate(&e0));
        THIN_TRY(hipEventCreate(&e1));
        THIN_TRY(hipEventRecord(e0, nullptr));
    }
    for (int32_t rep = 0; rep < std::max(repeats, 1); ++rep)
        THIN_TRY(launch_thin_hist(d_src.as<ThinSource>(), n, d_lgam.as<double>(), factor, out_len,
                                  d_partial.as<double>(), d_out.as<double>(), nullptr));
    if (kernel_ms) {
        THIN_TRY(hipEventRecord(e1, nullptr));
        THIN_TRY(hipEventSynchronize(e1));
        float ms = 0.f;
        THIN_TRY(hipEventElapsedTime(&ms, e0, e1));
        *kernel_ms = (double)ms / std::max(repeats, 1);
    }
    THIN_TRY(hipMemcpy(out, d_out.ptr, (size_t)out_len * sizeof(double), hipMemcpyDeviceToHost));
#undef THIN_TRY
    return done(COVEST_OK);
}

int covest_thin_histogram(int32_t device, int64_t n, const int32_t *keys, const double *counts, double factor,
                          int64_t out_len, double *out)
{
    return thin_histogram_impl(device, n, keys, counts, factor, out_len, out, 1, nullptr);
}

int covest_thin_histogram_timed(int32_t device, int64_t n, const int32_t *keys, const double *counts, double factor,
                                int64_t out_len, double *out, int32_t repeats, double *kernel_ms)
{
    return thin_histogram_impl(device, n, keys, counts, factor, out_len, out, repeats, kernel_ms);
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
