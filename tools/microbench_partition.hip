// microbench_partition.hip -- can a TWO-LEVEL partition of 16-byte records beat one scattered {atomic, 16-byte store}
// per record (K-kmer's pass 1, kmer_bulk.hip: 1.8e10 records a second, the memory side's scattered-atomic rate)?
//   S  the one-level scatter as it is: record -> one of 2^22 buckets, a returning atomic on the bucket's cursor in HBM,
//      a 16-byte store behind it
//   A  level 1: a workgroup STAGES its records per coarse bin in LDS (C bins x CAP records) and flushes a full bin with
//      ONE returning atomic and CAP x 16 contiguous bytes
//   B  level 2: a workgroup takes one coarse bin (its records contiguous in HBM), the F fine buckets' cursors in LDS,
//      and scatters the records inside the bin's own region (16-byte stores, a write frontier of F sectors)
// Records: {64-bit pseudo-random code, 64-bit tag}; the bucket is a hash of the code.  Prints records a second.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/microbench_partition tools/microbench_partition.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef unsigned long long u64;

__device__ __forceinline__ u64 mix(u64 x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}

// S: one level
__global__ __launch_bounds__(256) void scatter_one(u64 n, unsigned log2_buckets, u64 room, u64 *cursors, ulonglong2 *out)
{
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        const u64 code = mix(i);
        const u64 b = code >> (64 - log2_buckets);
        const u64 pos = atomicAdd(&cursors[b], 1ull);
        if (pos < room)
            out[b * room + pos] = make_ulonglong2(code, i);
    }
}

// A: staged in LDS per coarse bin.  One record per thread and round; the thread that takes a bin's last place flushes it
// after the barrier.  (The records of a round that find their bin full wait for the next round.)
template <int C, int CAP>
__global__ __launch_bounds__(256) void stage_flush(u64 n, u64 room, u64 *cursors, ulonglong2 *out)
{
    extern __shared__ unsigned char lds_raw[];
    ulonglong2 *stage = reinterpret_cast<ulonglong2 *>(lds_raw);             // [C][CAP]
    unsigned *cnt = reinterpret_cast<unsigned *>(lds_raw + (size_t)C * CAP * 16); // [C]
    for (int i = threadIdx.x; i < C; i += blockDim.x)
        cnt[i] = 0u;
    __syncthreads();
    const u64 per = (n + gridDim.x - 1) / gridDim.x;
    const u64 first = (u64)blockIdx.x * per, last = first + per < n ? first + per : n;
    for (u64 base = first; base < last; base += blockDim.x) {
        const u64 i = base + threadIdx.x;
        bool pending = i < last;
        const u64 code = mix(i);
        const unsigned bin = (unsigned)(code >> 54) & (C - 1);
        while (__syncthreads_or(pending)) {
            unsigned p = ~0u;
            if (pending) {
                p = atomicAdd(&cnt[bin], 1u);
                if (p < (unsigned)CAP) {
                    stage[bin * CAP + p] = make_ulonglong2(code, i);
                    pending = false;
                }
            }
            __syncthreads();
            if (p == (unsigned)CAP - 1u) { // this thread took the bin's last place: flush it
                const u64 pos = atomicAdd(&cursors[bin], (u64)CAP);
                if (pos + CAP <= room)
                    for (int r = 0; r < CAP; ++r)
                        out[(u64)bin * room + pos + r] = stage[bin * CAP + r];
            }
            __syncthreads();
            if (p == (unsigned)CAP - 1u)
                cnt[bin] = 0u;
            __syncthreads();
        }
    }
    // what is left in the bins
    for (int b = threadIdx.x; b < C; b += blockDim.x) {
        const unsigned c = cnt[b] < (unsigned)CAP ? cnt[b] : (unsigned)CAP;
        if (c) {
            const u64 pos = atomicAdd(&cursors[b], (u64)c);
            for (unsigned r = 0; r < c; ++r)
                if (pos + r < room)
                    out[(u64)b * room + pos + r] = stage[b * CAP + r];
        }
    }
}

// A2: the same with a wave-cooperative flush -- the bins that filled up in a round are listed in LDS and every wave
// writes whole bins, CAP lanes a bin (contiguous 16-byte stores of neighbouring lanes)
template <int C, int CAP>
__global__ __launch_bounds__(256) void stage_flush_coop(u64 n, u64 room, u64 *cursors, ulonglong2 *out)
{
    extern __shared__ unsigned char lds_raw[];
    ulonglong2 *stage = reinterpret_cast<ulonglong2 *>(lds_raw);
    unsigned *cnt = reinterpret_cast<unsigned *>(lds_raw + (size_t)C * CAP * 16);
    __shared__ unsigned full_list[256];
    __shared__ unsigned n_full;
    for (int i = threadIdx.x; i < C; i += blockDim.x)
        cnt[i] = 0u;
    if (threadIdx.x == 0)
        n_full = 0u;
    __syncthreads();
    const u64 per = (n + gridDim.x - 1) / gridDim.x;
    const u64 first = (u64)blockIdx.x * per, last = first + per < n ? first + per : n;
    for (u64 base = first; base < last; base += blockDim.x) {
        const u64 i = base + threadIdx.x;
        bool pending = i < last;
        const u64 code = mix(i);
        const unsigned bin = (unsigned)(code >> 54) & (C - 1);
        while (__syncthreads_or(pending)) {
            if (pending) {
                const unsigned p = atomicAdd(&cnt[bin], 1u);
                if (p < (unsigned)CAP) {
                    stage[bin * CAP + p] = make_ulonglong2(code, i);
                    pending = false;
                    if (p == (unsigned)CAP - 1u)
                        full_list[atomicAdd(&n_full, 1u)] = bin;
                }
            }
            __syncthreads();
            const unsigned nf = n_full;
            for (unsigned f = threadIdx.x / CAP; f < nf; f += blockDim.x / CAP) {
                const unsigned b = full_list[f];
                const unsigned r = threadIdx.x % CAP;
                u64 pos = 0;
                if (r == 0)
                    pos = atomicAdd(&cursors[b], (u64)CAP);
                pos = __shfl(pos, (threadIdx.x & 63) - r, 64);
                if (pos + CAP <= room)
                    out[(u64)b * room + pos + r] = stage[b * CAP + r];
            }
            __syncthreads();
            for (unsigned f = threadIdx.x; f < nf; f += blockDim.x)
                cnt[full_list[f]] = 0u;
            if (threadIdx.x == 0)
                n_full = 0u;
            __syncthreads();
        }
    }
    for (int b = threadIdx.x; b < C; b += blockDim.x) {
        const unsigned c = cnt[b] < (unsigned)CAP ? cnt[b] : (unsigned)CAP;
        if (c) {
            const u64 pos = atomicAdd(&cursors[b], (u64)c);
            for (unsigned r = 0; r < c; ++r)
                if (pos + r < room)
                    out[(u64)b * room + pos + r] = stage[b * CAP + r];
        }
    }
}

// B: one workgroup per coarse bin: the bin's records (contiguous, `per_bin` of them) to F fine buckets inside the bin's
// region, cursors in LDS
template <int F>
__global__ __launch_bounds__(1024) void refine(const ulonglong2 *in, u64 per_bin, u64 fine_room, ulonglong2 *out, unsigned n_bins)
{
    __shared__ unsigned cur[F];
    for (unsigned bin = blockIdx.x; bin < n_bins; bin += gridDim.x) {
        for (int i = threadIdx.x; i < F; i += blockDim.x)
            cur[i] = 0u;
        __syncthreads();
        const ulonglong2 *src = in + (u64)bin * per_bin;
        ulonglong2 *dst = out + (u64)bin * F * fine_room;
        for (u64 i = threadIdx.x; i < per_bin; i += blockDim.x) {
            const ulonglong2 r = src[i];
            const unsigned f = (unsigned)(mix(r.x ^ 0x9E3779B97F4A7C15ull) >> 40) & (F - 1);
            const unsigned pos = atomicAdd(&cur[f], 1u);
            if (pos < fine_room)
                dst[(u64)f * fine_room + pos] = r;
        }
        __syncthreads();
    }
}

template <class Launch>
double timed(Launch go)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    go();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    go();
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        printf("  (error: %s)\n", hipGetErrorString(e));
    return ms;
}

int main()
{
    const u64 n = 1ull << 28; // 2.7e8 records = 4.3 GB (a 1.6 Gbp input's)
    ulonglong2 *a = nullptr, *b = nullptr;
    u64 *cursors = nullptr;
    const u64 slack = n + n / 2;
    if (hipMalloc(&a, slack * 16) != hipSuccess || hipMalloc(&b, slack * 16) != hipSuccess ||
        hipMalloc(&cursors, (1ull << 22) * 8) != hipSuccess) {
        printf("allocation failed\n");
        return 1;
    }
    int n_cu = 256;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) == hipSuccess)
        n_cu = prop.multiProcessorCount;
    printf("%d CUs, %llu records of 16 bytes\n", n_cu, n);
    for (unsigned lb : {22u, 19u}) {
        const u64 room = slack >> lb;
        const double ms = timed([&] {
            (void)hipMemsetAsync(cursors, 0, (1ull << lb) * 8, 0);
            hipLaunchKernelGGL(scatter_one, dim3(n_cu * 16), dim3(256), 0, 0, n, lb, room, cursors, a);
        });
        printf("S  one level, 2^%u buckets: %7.2f ms  %.3g records/s\n", lb, ms, n / (ms * 1e-3));
    }
#define STAGE(KERNEL, NAME, C, CAP, WGS)                                                                              \
    {                                                                                                                 \
        const size_t lds = (size_t)C * CAP * 16 + (size_t)C * 4;                                                      \
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&KERNEL<C, CAP>),                                    \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                              \
        const u64 room = slack / C;                                                                                   \
        const double ms = timed([&] {                                                                                 \
            (void)hipMemsetAsync(cursors, 0, (size_t)C * 8, 0);                                                       \
            hipLaunchKernelGGL((KERNEL<C, CAP>), dim3(n_cu * WGS), dim3(256), lds, 0, n, room, cursors, a);           \
        });                                                                                                           \
        printf("%s C=%d CAP=%d (%zu KB LDS, %d wg/CU): %7.2f ms  %.3g records/s\n", NAME, C, CAP, lds >> 10, WGS, ms, \
               n / (ms * 1e-3));                                                                                      \
    }
    STAGE(stage_flush, "A  staged, one lane flushes ", 1024, 4, 2)
    STAGE(stage_flush, "A  staged, one lane flushes ", 512, 4, 4)
    STAGE(stage_flush, "A  staged, one lane flushes ", 1024, 8, 1)
    STAGE(stage_flush_coop, "A2 staged, lanes share a flush", 1024, 4, 2)
    STAGE(stage_flush_coop, "A2 staged, lanes share a flush", 512, 4, 4)
    STAGE(stage_flush_coop, "A2 staged, lanes share a flush", 512, 8, 2)
    STAGE(stage_flush_coop, "A2 staged, lanes share a flush", 1024, 8, 1)
    STAGE(stage_flush_coop, "A2 staged, lanes share a flush", 256, 8, 4)
    // B: a staged array as level 1 leaves it (here: any contiguous records), C bins of n / C records
#define REFINE(F, C, WGS)                                                                                             \
    {                                                                                                                 \
        const u64 per_bin = n / C;                                                                                    \
        const u64 fine_room = (per_bin + per_bin / 2) / F;                                                            \
        const double ms = timed([&] {                                                                                 \
            hipLaunchKernelGGL(refine<F>, dim3(WGS), dim3(1024), 0, 0, a, per_bin, fine_room, b, (unsigned)C);        \
        });                                                                                                           \
        printf("B  refine C=%d bins x F=%d fine (%llu records a bin, frontier %d KB), %d workgroups: %7.2f ms  %.3g records/s\n", \
               C, F, per_bin, F * 64 / 1024, WGS, ms, n / (ms * 1e-3));                                               \
    }
    REFINE(4096, 1024, n_cu)
    REFINE(4096, 1024, n_cu / 2)
    REFINE(4096, 1024, n_cu / 4)
    REFINE(8192, 512, n_cu)
    REFINE(8192, 512, n_cu / 2)
    REFINE(2048, 1024, n_cu)
    REFINE(1024, 1024, n_cu)
    return 0;
}
