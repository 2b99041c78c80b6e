// microbench_f64.hip -- what does gfx950 give fp64 work?  (1) v_fma_f64 rate,
// (2) v_mfma_f64_16x16x4_f64 rate, (3) both interleaved in one wave: do the
// matrix pipe and the vector pipe overlap for fp64?  (4) exp/log cost.
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench_f64.hip -o gpurun_out/microbench_f64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 4096;

__global__ __launch_bounds__(256) void k_fma(double *out, double a, double b)
{
    double v[8];
    for (int i = 0; i < 8; ++i) v[i] = a + threadIdx.x * 1e-9 + i;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = fma(v[i], b, a);
    }
    double s = 0; for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_mulmuladd(double *out, double a, double b)
{
    // the 2-op-per-term recurrence shape: t *= x; g += t
    double t[8], x[8], g = 0;
    for (int i = 0; i < 8; ++i) { t[i] = a + threadIdx.x * 1e-9 + i; x[i] = b + i * 1e-12; }
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { t[i] *= x[i]; g += t[i]; }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = g;
}

__global__ __launch_bounds__(256) void k_mfma(double *out, double a, double b)
{
    d4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (d4){0, 0, 0, 0};
    double av = a + threadIdx.x * 1e-9, bv = b;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[i], 0, 0, 0);
    }
    double s = 0; for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// dependent-accumulator chains: NACC independent accumulators per wave, WAVES_PER_SIMD set by the launch
template <int NACC>
__global__ __launch_bounds__(512) void k_mfma_chain(double *out, double a, double b)
{
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
    double av = a + threadIdx.x * 1e-9, bv = b;
    for (int it = 0; it < ITERS * 4 / NACC; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[i], 0, 0, 0);
    }
    double s = 0; for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NFMA>
__global__ __launch_bounds__(256) void k_both(double *out, double a, double b)
{
    d4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (d4){0, 0, 0, 0};
    double v[8];
    for (int i = 0; i < 8; ++i) v[i] = a + threadIdx.x * 1e-9 + i;
    double av = a + threadIdx.x * 1e-9, bv = b;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[i], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NFMA; ++j) v[(i * NFMA + j) & 7] = fma(v[(i * NFMA + j) & 7], b, a);
        }
    }
    double s = 0; for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_exp(double *out, double a)
{
    double v[4];
    for (int i = 0; i < 4; ++i) v[i] = -a - threadIdx.x * 1e-3 - i;
    double s = 0;
    for (int it = 0; it < ITERS / 8; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { s += exp(v[i]); v[i] -= 1e-6; }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_log(double *out, double a)
{
    double v[4];
    for (int i = 0; i < 4; ++i) v[i] = a + threadIdx.x * 1e-3 + i;
    double s = 0;
    for (int it = 0; it < ITERS / 8; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { s += log(v[i]); v[i] += 1e-6; }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class F>
static float time_kernel(F launch, int reps = 5)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(e0, 0);
        launch();
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}

int main()
{
    const int blocks = 256 * 8, threads = 256; // 8 workgroups per CU: 8 waves per SIMD
    double *out;
    CHECK(hipMalloc(&out, sizeof(double) * blocks * threads));
    const double lanes = (double)blocks * threads;
    float ms;
    ms = time_kernel([&] { hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(threads), 0, 0, out, 1.0, 0.999999); });
    printf("v_fma_f64        : %8.3f ms  %7.2f TFLOP/s\n", ms, lanes * ITERS * 8 * 2 / ms / 1e9);
    ms = time_kernel([&] { hipLaunchKernelGGL(k_mulmuladd, dim3(blocks), dim3(threads), 0, 0, out, 1.0, 0.999999); });
    printf("mul+add per term : %8.3f ms  %7.2f Tterm/s (2 instr/term)\n", ms, lanes * ITERS * 8 / ms / 1e9);
    ms = time_kernel([&] { hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(threads), 0, 0, out, 1.0, 0.999999); });
    const double mfma_flop = (double)blocks * (threads / 64) * ITERS * 4 * (16 * 16 * 4 * 2);
    printf("mfma_f64_16x16x4 : %8.3f ms  %7.2f TFLOP/s\n", ms, mfma_flop / ms / 1e9);
    float ms_m = ms;
    ms = time_kernel([&] { hipLaunchKernelGGL(k_both<4>, dim3(blocks), dim3(threads), 0, 0, out, 1.0, 0.999999); });
    printf("mfma + 4 fma each: %8.3f ms  (mfma alone %.3f) mfma %7.2f + valu %7.2f TFLOP/s\n", ms, ms_m, mfma_flop / ms / 1e9,
           lanes * ITERS * 16 * 2 / ms / 1e9);
    ms = time_kernel([&] { hipLaunchKernelGGL(k_both<8>, dim3(blocks), dim3(threads), 0, 0, out, 1.0, 0.999999); });
    printf("mfma + 8 fma each: %8.3f ms  mfma %7.2f + valu %7.2f TFLOP/s\n", ms, mfma_flop / ms / 1e9, lanes * ITERS * 32 * 2 / ms / 1e9);
    ms = time_kernel([&] { hipLaunchKernelGGL(k_both<16>, dim3(blocks), dim3(threads), 0, 0, out, 1.0, 0.999999); });
    printf("mfma +16 fma each: %8.3f ms  mfma %7.2f + valu %7.2f TFLOP/s\n", ms, mfma_flop / ms / 1e9, lanes * ITERS * 64 * 2 / ms / 1e9);
    for (int wps = 1; wps <= 8; wps *= 2) { // fp64 VALU rate against waves per SIMD (256-thread workgroups)
        const int nb = 256 * wps;
        const double ln = (double)nb * threads;
        ms = time_kernel([&] { hipLaunchKernelGGL(k_fma, dim3(nb), dim3(threads), 0, 0, out, 1.0, 0.999999); });
        printf("v_fma_f64, %d wave(s)/SIMD      : %8.3f ms  %7.2f TFLOP/s\n", wps, ms, ln * ITERS * 8 * 2 / ms / 1e9);
        ms = time_kernel([&] { hipLaunchKernelGGL(k_mulmuladd, dim3(nb), dim3(threads), 0, 0, out, 1.0, 0.999999); });
        printf("mul+add chain, %d wave(s)/SIMD  : %8.3f ms  %7.2f Tterm/s\n", wps, ms, ln * ITERS * 8 / ms / 1e9);
    }
    {   // one 512-thread workgroup per CU = 2 waves per SIMD, as K-factored runs
        const double fl = 256.0 * 8 * ITERS * 4 * (16 * 16 * 4 * 2);
        ms = time_kernel([&] { hipLaunchKernelGGL(k_mfma_chain<1>, dim3(256), dim3(512), 0, 0, out, 1.0, 0.999999); });
        printf("mfma chain x1, 2 waves/SIMD: %8.3f ms %7.2f TFLOP/s\n", ms, fl / ms / 1e9);
        ms = time_kernel([&] { hipLaunchKernelGGL(k_mfma_chain<2>, dim3(256), dim3(512), 0, 0, out, 1.0, 0.999999); });
        printf("mfma chain x2, 2 waves/SIMD: %8.3f ms %7.2f TFLOP/s\n", ms, fl / ms / 1e9);
        ms = time_kernel([&] { hipLaunchKernelGGL(k_mfma_chain<4>, dim3(256), dim3(512), 0, 0, out, 1.0, 0.999999); });
        printf("mfma chain x4, 2 waves/SIMD: %8.3f ms %7.2f TFLOP/s\n", ms, fl / ms / 1e9);
        const double fl1 = 256.0 * 4 * ITERS * 4 * (16 * 16 * 4 * 2);
        ms = time_kernel([&] { hipLaunchKernelGGL(k_mfma_chain<1>, dim3(256), dim3(256), 0, 0, out, 1.0, 0.999999); });
        printf("mfma chain x1, 1 wave/SIMD : %8.3f ms %7.2f TFLOP/s\n", ms, fl1 / ms / 1e9);
        ms = time_kernel([&] { hipLaunchKernelGGL(k_mfma_chain<2>, dim3(256), dim3(256), 0, 0, out, 1.0, 0.999999); });
        printf("mfma chain x2, 1 wave/SIMD : %8.3f ms %7.2f TFLOP/s\n", ms, fl1 / ms / 1e9);
        ms = time_kernel([&] { hipLaunchKernelGGL(k_mfma_chain<4>, dim3(256), dim3(256), 0, 0, out, 1.0, 0.999999); });
        printf("mfma chain x4, 1 wave/SIMD : %8.3f ms %7.2f TFLOP/s\n", ms, fl1 / ms / 1e9);
    }
    ms = time_kernel([&] { hipLaunchKernelGGL(k_exp, dim3(blocks), dim3(threads), 0, 0, out, 1.0); });
    printf("exp(f64)         : %8.3f ms  %7.2f Gexp/s  (= %.1f fma-equivalents each)\n", ms, lanes * (ITERS / 8) * 4 / ms / 1e6, 0.0);
    float ms_e = ms;
    ms = time_kernel([&] { hipLaunchKernelGGL(k_log, dim3(blocks), dim3(threads), 0, 0, out, 1.0); });
    printf("log(f64)         : %8.3f ms  %7.2f Glog/s\n", ms, lanes * (ITERS / 8) * 4 / ms / 1e6);
    (void)ms_e;
    hipFree(out);
    return 0;
}
