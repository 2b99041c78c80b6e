// microbench_phaseb.hip -- the MFMA step loop of ll_factored.hip in isolation: what limits it?
// Variants: NSLOT accumulators per wave; weight produced by VALU (mul + select) or constant;
// A fragment from LDS (prefetched) or from a register.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int STEPS = 64, TILES = 512;

template <int NSLOT, bool VALU_W, bool LDS_A>
__global__ __launch_bounds__(512) void k_loop(double *out, const int *tq_in, double r4v)
{
    __shared__ double G[32 * 514];
    for (int i = threadIdx.x; i < 32 * 514; i += blockDim.x) G[i] = 1.0 + i * 1e-9;
    __syncthreads();
    const int lane = threadIdx.x & 63, col = lane & 15, kq = lane >> 4;
    d4 acc[NSLOT];
    double wrun[NSLOT], r4[NSLOT];
    int tq[NSLOT];
    for (int k = 0; k < NSLOT; ++k) { acc[k] = (d4){0, 0, 0, 0}; wrun[k] = 1.0 + k; r4[k] = r4v; tq[k] = tq_in[k * 64 + lane]; }
    const double *arow0 = G + col * 514 + kq, *arow1 = G + (16 + col) * 514 + kq;
    for (int t = 0; t < TILES; ++t) {
        double a0n = arow0[0], a1n = arow1[0];
        for (int step = 0; step < STEPS; ++step) {
            double a0 = a0n, a1 = a1n;
            if (LDS_A) { a0n = arow0[4 * step + 4]; a1n = arow1[4 * step + 4]; }
            const int o_here = 1 + 4 * step + kq;
#pragma unroll
            for (int k = 0; k < NSLOT; ++k) {
                double w = wrun[k];
                if (VALU_W) { wrun[k] *= r4[k]; w = (o_here < tq[k]) ? w : 0.0; }
                acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64((k & 1) ? a1 : a0, w, acc[k], 0, 0, 0);
            }
        }
    }
    double s = 0;
    for (int k = 0; k < NSLOT; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3] + wrun[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class F> float timeit(F f)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    f(); (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) { (void)hipEventRecord(e0, 0); f(); (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
    return best;
}

template <int NSLOT, bool VW, bool LA> void run(double *out, int *tq, const char *name)
{
    float ms = timeit([&] { hipLaunchKernelGGL((k_loop<NSLOT, VW, LA>), dim3(256), dim3(512), 0, 0, out, tq, 0.999999); });
    double mfmas = 256.0 * 8 * TILES * STEPS * NSLOT;
    printf("%-34s slots=%d: %8.3f ms  %6.1f ns/MFMA/SIMD  %6.2f TFLOP/s\n", name, NSLOT, ms, ms * 1e6 / (mfmas / 1024), mfmas * 2048 / ms / 1e9);
}

int main()
{
    double *out; int *tq; (void)hipMalloc(&out, 8 * 256 * 512); (void)hipMalloc(&tq, 4 * 64 * 8);
    int h[64 * 8]; for (int i = 0; i < 512; ++i) h[i] = 1000000; (void)hipMemcpy(tq, h, sizeof(h), hipMemcpyHostToDevice);
    run<1, false, false>(out, tq, "const w, reg A");
    run<1, true, false>(out, tq, "VALU w, reg A");
    run<1, false, true>(out, tq, "const w, LDS A");
    run<1, true, true>(out, tq, "VALU w, LDS A");
    run<2, true, true>(out, tq, "VALU w, LDS A");
    run<3, true, true>(out, tq, "VALU w, LDS A");
    run<6, true, true>(out, tq, "VALU w, LDS A");
    run<6, false, true>(out, tq, "const w, LDS A");
    run<6, true, false>(out, tq, "VALU w, reg A");
    return 0;
}
