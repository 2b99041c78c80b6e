// microbench_ops.hip -- issue cost of the individual instructions of the fast kernels'
// per-key loop on gfx950, relative to v_fma_f64: is v_frexp_mant_f64 / v_cvt_f64_i32 /
// v_mov_b64 a full-rate slot?  Each kernel runs 8 independent chains of ONE instruction,
// 4 waves per SIMD, every CU busy.  Output: ps per wave-instruction per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench_ops.hip -o tools/bin/microbench_ops
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITERS = 2048;

#define KERNEL(name, DECL, BODY)                                                   \
    __global__ __launch_bounds__(256) void name(double *out, double a, double b)   \
    {                                                                              \
        double v[8];                                                               \
        int w[8];                                                                  \
        for (int i = 0; i < 8; ++i) { v[i] = a + threadIdx.x * 1e-9 + i; w[i] = threadIdx.x + i; } \
        DECL;                                                                      \
        for (int it = 0; it < ITERS; ++it) {                                       \
            _Pragma("unroll") for (int i = 0; i < 8; ++i) { BODY; }                \
        }                                                                          \
        double s = 0;                                                              \
        for (int i = 0; i < 8; ++i) s += v[i] + w[i];                              \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                            \
    }

KERNEL(k_fma, , asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v[i]) : "v"(b), "v"(a)))
KERNEL(k_mul, , asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v[i]) : "v"(b)))
KERNEL(k_add, , asm volatile("v_add_f64 %0, %0, %1" : "+v"(v[i]) : "v"(b)))
KERNEL(k_fmac, , asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(v[i]) : "v"(b), "v"(a)))
KERNEL(k_fma_s, , asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v[i]) : "v"(b), "s"(a)))
KERNEL(k_mov64, double t[8], asm volatile("v_mov_b64 %0, %1" : "=v"(t[i]) : "v"(v[i])); asm volatile("v_mov_b64 %0, %1" : "=v"(v[i]) : "v"(t[i])))
KERNEL(k_mov32, int t[8], asm volatile("v_mov_b32 %0, %1" : "=v"(t[i]) : "v"(w[i])); asm volatile("v_mov_b32 %0, %1" : "=v"(w[i]) : "v"(t[i])))
KERNEL(k_frexp_mant, , asm volatile("v_frexp_mant_f64 %0, %0" : "+v"(v[i])))
KERNEL(k_frexp_exp, , asm volatile("v_frexp_exp_i32_f64 %0, %1" : "=v"(w[i]) : "v"(v[i])))
KERNEL(k_cvt, , asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(v[i]) : "v"(w[i])))
KERNEL(k_and, , asm volatile("v_and_b32 %0, %0, %1" : "+v"(w[i]) : "v"(w[(i + 1) & 7])))
KERNEL(k_cmp, , asm volatile("v_cmp_ge_f64 vcc, %0, %1" : : "v"(v[i]), "v"(b) : "vcc"))
KERNEL(k_ldexp, , asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(v[i]) : "v"(w[i])))
KERNEL(k_rcp, , asm volatile("v_rcp_f64 %0, %0" : "+v"(v[i])))
KERNEL(k_cndmask, , asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(w[i]) : "v"(w[(i + 1) & 7]) : "vcc"))

__global__ __launch_bounds__(256) void k_ldsread(double *out, double a, double b)
{
    __shared__ __attribute__((aligned(16))) double tab[128];
    if (threadIdx.x < 128) tab[threadIdx.x] = a + threadIdx.x;
    __syncthreads();
    double s = 0;
    unsigned off = (threadIdx.x * 7 & 63) * 16;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const double2 e = *reinterpret_cast<const double2 *>(reinterpret_cast<const char *>(tab) + off);
            s += e.x;
            off = (off + 16 * (i + 1) + (unsigned)(e.y > 1e300)) & 1008;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class K>
int run(const char *name, K kern, double *d_out, double per_iter_instr, double base_ps)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int blocks = 256 * 4; // 4 workgroups of 4 waves per CU: 4 waves per SIMD
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, 1.000001, 0.999999);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r)
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, 1.000001, 0.999999);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    // per SIMD: 4 waves x ITERS x per_iter_instr instructions per launch
    const double instr = 5.0 * 4 * ITERS * per_iter_instr;
    const double ps = ms * 1e9 / instr;
    printf("%-14s %8.1f ps per wave-instruction  (%.2f x v_fma_f64)\n", name, ps, base_ps > 0 ? ps / base_ps : 1.0);
    return 0;
}

int main()
{
    double *d_out;
    CHECK(hipMalloc(&d_out, sizeof(double) * 256 * 1024 * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    // base
    hipLaunchKernelGGL(k_fma, dim3(1024), dim3(256), 0, 0, d_out, 1.000001, 0.999999);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k_fma, dim3(1024), dim3(256), 0, 0, d_out, 1.000001, 0.999999);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double base = ms * 1e9 / (5.0 * 4 * ITERS * 8);
    printf("v_fma_f64 base: %.1f ps per wave-instruction per SIMD = %.2f TFLOP/s\n", base,
           1024.0 * 128 / (base * 1e-12) / 1e12);
    run("v_fma_f64", k_fma, d_out, 8, base);
    run("v_mul_f64", k_mul, d_out, 8, base);
    run("v_add_f64", k_add, d_out, 8, base);
    run("v_fmac_f64", k_fmac, d_out, 8, base);
    run("v_fma_f64 sgpr", k_fma_s, d_out, 8, base);
    run("v_mov_b64", k_mov64, d_out, 16, base);
    run("v_mov_b32", k_mov32, d_out, 16, base);
    run("v_frexp_mant", k_frexp_mant, d_out, 8, base);
    run("v_frexp_exp", k_frexp_exp, d_out, 8, base);
    run("v_cvt_f64_i32", k_cvt, d_out, 8, base);
    run("v_and_b32", k_and, d_out, 8, base);
    run("v_cmp_ge_f64", k_cmp, d_out, 8, base);
    run("v_ldexp_f64", k_ldexp, d_out, 8, base);
    run("v_rcp_f64", k_rcp, d_out, 8, base);
    run("v_cndmask_b32", k_cndmask, d_out, 8, base);
    run("lds b128+add", k_ldsread, d_out, 8, base);
    return 0;
}
