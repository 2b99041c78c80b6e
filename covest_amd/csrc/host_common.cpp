// host_common.cpp -- error state, device selection, the ln j! table and threshold_o (host arithmetic: libm,
// as CPython) of libcovest_amd.so.  Compiled with hipcc, links only the HIP runtime.  There is no CPU compute path in
// this library: every likelihood value comes out of a gfx950 kernel.
#include "host.h"

using namespace covest;

namespace {
thread_local std::string g_last_error;
} // namespace

namespace covest {

// (also for the library's other translation units, reads_io.cpp: record the message covest_last_error returns)
int set_error(int code, const std::string &msg)
{
    g_last_error = msg;
    return code;
}

int fail_hip(hipError_t e, const char *what)
{
    g_last_error = std::string(what) + ": " + hipGetErrorString(e);
    if (e == hipErrorOutOfMemory) { // an allocation that did not fit: the caller may fall back (include/covest_amd.h)
        (void)hipGetLastError();    // (the runtime keeps the error until somebody reads it)
        return COVEST_E_NOMEM;
    }
    return (e == hipErrorNoDevice || e == hipErrorInvalidDevice || e == hipErrorInsufficientDriver)
               ? COVEST_E_NO_DEVICE
               : COVEST_E_HIP;
}

namespace {
struct DevCache {
    std::mutex mu;
    struct Entry {
        int device;
        void *ptr;
        size_t cap;
    };
    std::vector<Entry> free_list;
    size_t bytes = 0;
    std::vector<std::pair<int, void *>> pinned; // (device, block)
};
DevCache &dev_cache()
{
    static DevCache *c = new DevCache; // (never freed: the runtime may be gone when statics are destroyed)
    return *c;
}
constexpr size_t kDevCacheEntryMax = (size_t)8 << 20, kDevCacheTotalMax = (size_t)64 << 20;
} // namespace

namespace {
thread_local int tls_device_idle = 0;
}
DeviceIdleScope::DeviceIdleScope() { ++tls_device_idle; }
DeviceIdleScope::~DeviceIdleScope() { --tls_device_idle; }
bool DeviceIdleScope::active() { return tls_device_idle > 0; }

bool dev_cache_take(size_t bytes, void **ptr, size_t *cap, int *device)
{
    if (bytes == 0 || bytes > kDevCacheEntryMax)
        return false;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess)
        return false;
    DevCache &c = dev_cache();
    std::lock_guard<std::mutex> hold(c.mu);
    size_t best = c.free_list.size();
    for (size_t i = 0; i < c.free_list.size(); ++i) {
        const DevCache::Entry &e = c.free_list[i];
        // best fit, and no buffer more than four times what is asked for (a 4 MB buffer for 200 bytes would starve the
        // next large request)
        if (e.device == dev && e.cap >= bytes && e.cap <= std::max<size_t>(4 * bytes, 4096) &&
            (best == c.free_list.size() || e.cap < c.free_list[best].cap))
            best = i;
    }
    if (best == c.free_list.size())
        return false;
    *ptr = c.free_list[best].ptr;
    *cap = c.free_list[best].cap;
    *device = dev;
    c.bytes -= c.free_list[best].cap;
    c.free_list.erase(c.free_list.begin() + (std::ptrdiff_t)best);
    return true;
}

bool dev_cache_give(void *ptr, size_t cap, int dev)
{
    if (!ptr || cap == 0 || cap > kDevCacheEntryMax || dev < 0)
        return false;
    DevCache &c = dev_cache();
    std::lock_guard<std::mutex> hold(c.mu);
    if (c.bytes + cap > kDevCacheTotalMax || c.free_list.size() >= 256)
        return false;
    c.free_list.push_back({dev, ptr, cap});
    c.bytes += cap;
    return true;
}

void *pinned_block_take()
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    {
        DevCache &c = dev_cache();
        std::lock_guard<std::mutex> hold(c.mu);
        for (size_t i = 0; i < c.pinned.size(); ++i)
            if (c.pinned[i].first == dev) {
                void *p = c.pinned[i].second;
                c.pinned.erase(c.pinned.begin() + (std::ptrdiff_t)i);
                return p;
            }
    }
    void *p = nullptr;
    if (hipHostMalloc(&p, kPinnedBlockBytes, hipHostMallocMapped) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return p;
}

void pinned_block_give(void *p)
{
    if (!p)
        return;
    int dev = 0;
    (void)hipGetDevice(&dev);
    DevCache &c = dev_cache();
    std::lock_guard<std::mutex> hold(c.mu);
    if (c.pinned.size() < 64) {
        c.pinned.push_back({dev, p});
        return;
    }
    (void)hipHostFree(p);
}

SharedStage &shared_stage()
{
    static SharedStage *s = new SharedStage; // (never freed: the runtime may be gone when statics are destroyed)
    return *s;
}


// RepeatsModel.get_b_o / get_hist_threshold, covest/models.py:185-208, with libm
// pow as CPython's float ** int.  b_o is non-increasing in o for o >= 3 when
// 0 <= 1-q <= 1, so the first crossing is found by bisection and then confirmed
// against its left neighbours with the very same pow calls the linear scan of
// the reference would make; outside that domain the scan itself is used.
// (pw(o) = pow(1 - q, o - 3): libm's, called directly or looked up in a table of the very same calls)
template <class PowFn>
static int threshold_o_impl(double q1, double q2, double q, double thr, bool has_thr, int hist_max, PowFn pw)
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
    auto weight = [&](int o) { return head * pw(o); };
    if (!(base >= 0.0 && base <= 1.0) || !(head == head)) {
        for (int o = 3; o <= last; ++o)
            if (weight(o) <= thr)
                return o;
        return hist_max;
    }
    if (weight(3) <= thr)
        return 3;
    if (!(weight(last) <= thr))
        return hist_max;
    int lo = 3, hi = last; // f(lo) > thr, f(hi) <= thr
    while (hi - lo > 1) {
        const int mid = lo + (hi - lo) / 2;
        if (weight(mid) <= thr)
            hi = mid;
        else
            lo = mid;
    }
    while (hi > 3 && weight(hi - 1) <= thr)
        --hi;
    return hi;
}

int threshold_o_host(double q1, double q2, double q, double thr, bool has_thr, int hist_max)
{
    const double base = 1 - q;
    return threshold_o_impl(q1, q2, q, thr, has_thr, hist_max, [&](int o) { return std::pow(base, (double)(o - 3)); });
}

// threshold_o over the (q1, q2, q) product of three axes (clamped to the model's bounds), last axis fastest: the same
// decisions as threshold_o_host point by point, but the powers of one q are computed ONCE for all its (q1, q2) -- an
// optimize_grid iteration asks for 216 thresholds of 6 different q, 1 300 calls of pow otherwise (round 4).
void threshold_table(const covest_model *m, const double *a1, int64_t n1, const double *a2, int64_t n2, const double *a3,
                     int64_t n3, int32_t *out)
{
    // pw[o] = pow(1 - q, o - 3) for the q in hand; NaN: not asked for yet (the entries a q touched are set back, not the
    // whole table: 10 000 keys and 16 values of q would be a megabyte of fills)
    std::vector<double> pw((size_t)std::max(m->hist_max, 4) + 1, std::numeric_limits<double>::quiet_NaN());
    std::vector<int> touched;
    for (int64_t c = 0; c < n3; ++c) {
        const double q = clamp_one(m->dm, 4, a3[c]);
        const double base = 1 - q;
        for (int o : touched)
            pw[(size_t)o] = std::numeric_limits<double>::quiet_NaN();
        touched.clear();
        auto pow_of = [&](int o) {
            double &v = pw[(size_t)o];
            if (v != v) {
                v = std::pow(base, (double)(o - 3));
                touched.push_back(o); // (a power that IS NaN -- q is -- is computed again each time: still the same value)
            }
            return v;
        };
        for (int64_t a = 0; a < n1; ++a) {
            const double q1 = clamp_one(m->dm, 2, a1[a]);
            for (int64_t b = 0; b < n2; ++b)
                out[(a * n2 + b) * n3 + c] = threshold_o_impl(q1, clamp_one(m->dm, 3, a2[b]), q, m->threshold, m->has_threshold,
                                                              m->hist_max, pow_of);
        }
    }
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
// (after lgamma_ensure(j) or larger)
double lgamma_at(int64_t j)
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

} // namespace covest

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

} // extern "C"
