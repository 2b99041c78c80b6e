// thin_host.cpp -- covest_thin_histogram* of the C ABI over thin_hist.hip.
#include "host.h"

using namespace covest;

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
    auto done = [&](int code) { // (the buffers go with their DevBuf)
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

extern "C" {

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

} // extern "C"
