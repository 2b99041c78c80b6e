// microbench_atomics.hip -- what bounds K-kmer: the rate of scattered 8-byte operations on a table in HBM.
//   A  random 8-byte loads                     B  random 64-bit atomic adds
//   C  load + atomic add in the same 16-byte slot (K-kmer's shape)
//   D  C with 4 neighbouring lanes in one 64-byte sector      E  C with 8 neighbouring lanes in one 128-byte line
// for tables of 16 GB, 1 GB, 128 MB (inside the 256 MB Infinity Cache) and 2 MB (inside one XCD's L2).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ uint64_t mix(uint64_t x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}

template <int MODE>
__global__ __launch_bounds__(256) void k(uint64_t *tab, uint64_t mask, uint64_t n, uint64_t *sink)
{
    uint64_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t slot;
        if (MODE == 3)
            slot = ((mix(i >> 2) << 2) | (i & 3)) & mask;  // 4 lanes per 64 B
        else if (MODE == 4)
            slot = ((mix(i >> 3) << 3) | (i & 7)) & mask;  // 8 lanes per 128 B
        else if (MODE == 5)
            slot = ((mix(i >> 3) << 3) | (7 - (i & 7))) & mask;  // ... in descending order
        else if (MODE == 6)
            slot = ((mix(i >> 3) << 3) | ((i * 5 + (mix(i >> 3) & 7)) & 7)) & mask;  // ... permuted and rotated
        else if (MODE == 7)
            slot = ((mix(i >> 3) << 3) | (mix(i) & 7)) & mask;  // ... random slots of the line (collisions)
        else
            slot = mix(i) & mask;
        uint64_t *p = tab + 2 * slot;
        if (MODE == 0) {
            acc += __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (MODE == 1) {
            atomicAdd((unsigned long long *)p + 1, 1ull);
        } else {
            acc += __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            atomicAdd((unsigned long long *)p + 1, 1ull);
        }
    }
    if (acc == 0x1234567)
        *sink = acc;
}

template <int MODE>
double run(uint64_t *tab, uint64_t slots, uint64_t n, uint64_t *sink)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * 16), dim3(256), 0, 0, tab, slots - 1, n / 8, sink);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * 16), dim3(256), 0, 0, tab, slots - 1, n, sink);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return n / (ms * 1e-3);
}

int main()
{
    const uint64_t n = 400000000ull;
    uint64_t *tab, *sink;
    if (hipMalloc(&tab, 16ull << 30) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMalloc(&sink, 8);
    (void)hipMemset(tab, 0, 16ull << 30);
    const char *names[8] = {"A load", "B atomic add", "C load + add (one slot)", "D C, 4 lanes per 64 B", "E C, 8 lanes per 128 B",
                            "F E descending", "G E permuted", "H E random slots"};
    for (uint64_t bytes : {16ull << 30, 1ull << 30, 128ull << 20, 2ull << 20}) {
        const uint64_t slots = bytes / 16;
        double r[8] = {run<0>(tab, slots, n, sink), run<1>(tab, slots, n, sink), run<2>(tab, slots, n, sink),
                       run<3>(tab, slots, n, sink), run<4>(tab, slots, n, sink), run<5>(tab, slots, n, sink),
                       run<6>(tab, slots, n, sink), run<7>(tab, slots, n, sink)};
        printf("table %6llu MB:", (unsigned long long)(bytes >> 20));
        for (int i = 0; i < 8; ++i) printf("  %s %.2e/s", names[i], r[i]);
        printf("\n");
        fflush(stdout);
    }
    return 0;
}
