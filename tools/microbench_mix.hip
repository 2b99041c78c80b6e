// microbench_mix.hip -- one SIMD, two waves: wave A runs fp64 VALU (the recurrence shape), wave B runs
// dependent fp64 MFMAs.  How does the SIMD arbitrate them?  Per-wave elapsed cycles via s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int N = 20000;

// mode: 0 = both roles co-run (waves 0-3 VALU, 4-7 MFMA), 1 = only VALU waves work, 2 = only MFMA waves work
__global__ __launch_bounds__(512) void k_mix(double *out, long long *cyc, int mode, double b)
{
    const int wave = threadIdx.x >> 6;
    const bool valu_role = wave < 4;
    double v[8], x[8], g = 0;
    for (int i = 0; i < 8; ++i) { v[i] = 1.0 + threadIdx.x * 1e-9 + i; x[i] = b + i * 1e-12; }
    d4 acc = (d4){0, 0, 0, 0};
    double w = 1.0 + threadIdx.x * 1e-9;
    __syncthreads();
    const long long t0 = clock64();
    if (valu_role) {
        if (mode != 2)
            for (int it = 0; it < N; ++it) {
#pragma unroll
                for (int i = 0; i < 8; ++i) { v[i] *= x[i]; g += v[i]; }
            }
    } else {
        if (mode != 1)
            for (int it = 0; it < N; ++it) {
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(w, b, acc, 0, 0, 0);
            }
    }
    const long long t1 = clock64();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = g + acc[0] + acc[1] + acc[2] + acc[3];
}

int main()
{
    double *out; long long *cyc; (void)hipMalloc(&out, 8 * 256 * 512); (void)hipMalloc(&cyc, 8 * 256 * 8);
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(k_mix, dim3(256), dim3(512), 0, 0, out, cyc, mode, 0.9999999);
        (void)hipDeviceSynchronize();
        long long h[256 * 8]; (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        double sv = 0, sm = 0;
        for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? sv : sm) += (double)h[b * 8 + w];
        sv /= 256 * 4; sm /= 256 * 4;
        printf("mode %d (%s): VALU wave %9.0f cycles = %6.2f cyc per f64 instr (16/iter); MFMA wave %9.0f cycles = %6.1f cyc per MFMA\n",
               mode, mode == 0 ? "co-run" : mode == 1 ? "VALU only" : "MFMA only", sv, sv / (N * 16.0), sm, sm / (double)N);
    }
    return 0;
}
