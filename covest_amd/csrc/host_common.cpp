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
