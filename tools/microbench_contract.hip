// microbench_contract.hip -- what does the VALU work that accompanies every MFMA of K-factored's contraction cost?
// One loop iteration = 6 independent v_mfma_f64_16x16x4_f64 (the kernel's 6 accumulator slots), each with
//   variant 0: nothing else            1: + v_mul_f64 (weight advance)     2: + v_cmp + 2 v_cndmask (the cut-off mask)
//   variant 3: 1 + 2                   4: 3 + address add + ds_read_b64 of the A fragment (the real step)
// at 1 and 2 waves per SIMD.  Prints cycles per MFMA (wall clock x 2.4 GHz / MFMAs per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int N = 4000;

template <int V>
__global__ __launch_bounds__(512) void k(double *out, const int *cutp, double r)
{
    __shared__ double lds[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x)
        lds[i] = 1.0 + i * 1e-9;
    __syncthreads();
    d4 acc[6];
    double w[6], r4[6];
    int cut[6], aoff[6];
    for (int s = 0; s < 6; ++s) {
        acc[s] = (d4){0, 0, 0, 0};
        w[s] = 1.0 + threadIdx.x * 1e-9 + s;
        r4[s] = r + s * 1e-12;
        cut[s] = cutp[(threadIdx.x + s) & 63];
        aoff[s] = ((threadIdx.x & 63) * 66 + s * 4) & 4095;
    }
    double a[6];
    for (int s = 0; s < 6; ++s)
        a[s] = lds[aoff[s]];
    for (int i = 0; i < N; ++i) {
        if (V >= 4) {
#pragma unroll
            for (int s = 0; s < 6; ++s)
                a[s] = lds[aoff[s] + 4 * (i & 511)];
        }
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            double ww = w[s];
            if (V == 2 || V >= 3)
                ww = (i < cut[s]) ? w[s] : 0.0;
            if (V == 1 || V >= 3)
                w[s] *= r4[s];
            acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], ww, acc[s], 0, 0, 0);
        }
    }
    double g = 0;
    for (int s = 0; s < 6; ++s)
        g += acc[s][0] + acc[s][1] + acc[s][2] + acc[s][3] + w[s];
    out[blockIdx.x * blockDim.x + threadIdx.x] = g;
}

template <int V>
void run(double *out, int *cut, int threads)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<V>, dim3(256), dim3(threads), 0, 0, out, cut, 0.9999999);
    (void)hipEventRecord(e0, 0);
    for (int rep = 0; rep < 5; ++rep)
        hipLaunchKernelGGL(k<V>, dim3(256), dim3(threads), 0, 0, out, cut, 0.9999999);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = (double)N * 6 * (threads / 256); // MFMAs one SIMD executes per launch
    printf("variant %d, %d wave(s)/SIMD: %.1f cycles per MFMA (at 2.4 GHz)\n", V, threads / 256, ms / 5 * 1e-3 * 2.4e9 / per_simd);
}

int main()
{
    double *out;
    int *cut;
    (void)hipMalloc(&out, 8 * 256 * 512);
    (void)hipMalloc(&cut, 256);
    int h[64];
    for (int i = 0; i < 64; ++i)
        h[i] = N - 3 + (i & 3);
    (void)hipMemcpy(cut, h, sizeof(h), hipMemcpyHostToDevice);
    for (int threads = 256; threads <= 512; threads += 256) {
        run<0>(out, cut, threads);
        run<1>(out, cut, threads);
        run<2>(out, cut, threads);
        run<3>(out, cut, threads);
        run<4>(out, cut, threads);
    }
    return 0;
}
