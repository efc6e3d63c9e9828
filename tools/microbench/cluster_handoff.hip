// Can the intermediate of a two-pass 2^16-point transform stay on chip on gfx950?
//
// The two-pass NTT moves every coefficient through HBM twice (in -> mid, mid -> out).  This
// microbenchmark reproduces exactly the memory shapes of ntt2_fwd_p1<8>/ntt2_fwd_p2<8> (4096-word
// tiles, 256 threads, 16 words per thread, `work` dependent FMAs per word per pass standing in for
// the butterflies) and measures
//   A. the two kernels back to back, intermediate = whole batch (today's engine);
//   B. the same two kernels over chunks of C transforms with ONE reused chunk-sized intermediate
//      (does the 256 MiB Infinity Cache absorb it?);
//   C. one persistent kernel in which the 16 workgroups of a "cluster" (same XCD by blockIdx % 8)
//      run pass 1 of a transform, meet at a cluster barrier and run pass 2, the intermediate being a
//      cluster-private slot that never grows beyond a few MiB per XCD (does it stay in L2?):
//        V0  plain stores, release fence (agent) / acquire fence (agent)   - always valid
//        V1  sc1 stores + sc1 loads, s_waitcnt vmcnt(0) before the arrive     - always valid
//        V2  plain stores + vmcnt(0), sc1 loads, no fence                     - valid on one XCD only
// Every variant is verified word for word.  Spin loops are capped: a lost barrier is reported, not hung.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int kT = 256;
constexpr size_t kN = 65536;

__device__ __forceinline__ void work_on(uint64_t (&x)[16], int work, double a, double b)
{
    // 16 independent FMA chains that return their inputs + 1 (a = 1, b = 0 at run time; opaque to the compiler)
    double d[16];
#pragma unroll
    for (int e = 0; e < 16; e++)
        d[e] = __builtin_bit_cast(double, (x[e] & 0x000fffffffffffffull) | 0x4330000000000000ull);
    for (int i = 0; i < work; i++)
    {
#pragma unroll
        for (int e = 0; e < 16; e++)
            d[e] = __builtin_fma(d[e], a, b);
    }
#pragma unroll
    for (int e = 0; e < 16; e++)
        x[e] = ((__builtin_bit_cast(uint64_t, d[e]) & 0x000fffffffffffffull) | (x[e] & 0xfff0000000000000ull)) + 1;
}

// ---- pass bodies (word indices inside one transform)
// pass 1, column tile cg: thread (c = tid & 15, hi = tid >> 4) reads (row e*16 + hi, col cg*16 + c)
// and writes tile order (hi*16 + cg)*256 + e*16 + c
template <int LD, int ST>
__device__ __forceinline__ void pass1(const uint64_t *in, uint64_t *mid, unsigned cg, unsigned tid, int work, double a, double b)
{
    const unsigned c = tid & 15, hi = tid >> 4;
    uint64_t x[16];
#pragma unroll
    for (int e = 0; e < 16; e++)
        x[e] = in[(size_t)(e * 16 + hi) * 256 + cg * 16 + c];
    work_on(x, work, a, b);
    uint64_t *o = mid + (size_t)(hi * 16 + cg) * 256 + c;
#pragma unroll
    for (int e = 0; e < 16; e++)
    {
        if (ST == 1)
            __hip_atomic_store(o + e * 16, x[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else
            o[e * 16] = x[e];
    }
}
// pass 2, row tile hg: contiguous 32 KiB in, contiguous 32 KiB out
template <int LD, int ST>
__device__ __forceinline__ void pass2(const uint64_t *mid, uint64_t *out, unsigned hg, unsigned tid, int work, double a, double b)
{
    uint64_t x[16];
    const uint64_t *m = mid + (size_t)hg * 4096 + tid;
#pragma unroll
    for (int e = 0; e < 16; e++)
    {
        if (LD == 1)
            x[e] = __hip_atomic_load(m + e * 256, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else
            x[e] = m[e * 256];
    }
    work_on(x, work, a, b);
    uint64_t *o = out + (size_t)hg * 4096 + tid;
#pragma unroll
    for (int e = 0; e < 16; e++)
        o[e * 256] = x[e];
}

// ---- A/B: separate kernels
__global__ void __launch_bounds__(kT) k_p1(const uint64_t *in, uint64_t *mid, int work, double a, double b)
{
    extern __shared__ uint64_t lds[];
    const size_t t = blockIdx.x >> 4;
    pass1<0, 0>(in + t * kN, mid + t * kN, blockIdx.x & 15, threadIdx.x, work, a, b);
    if (a == 2.0)
        lds[threadIdx.x] = 1;
}
__global__ void __launch_bounds__(kT) k_p2(const uint64_t *mid, uint64_t *out, int work, double a, double b)
{
    extern __shared__ uint64_t lds[];
    const size_t t = blockIdx.x >> 4;
    pass2<0, 0>(mid + t * kN, out + t * kN, blockIdx.x & 15, threadIdx.x, work, a, b);
    if (a == 2.0)
        lds[threadIdx.x] = 1;
}

// ---- C: persistent cluster kernel
struct ClusterArgs
{
    const uint64_t *in;
    uint64_t *out;
    uint64_t *mid;        // [ncluster][slots][kN]
    unsigned *counters;   // [ncluster] monotonic arrive counters (64-byte spaced)
    unsigned *status;     // [0] = lost barriers, [1] = clusters with members on different XCDs, [2..] xcc of each block
    unsigned transforms;
    unsigned slots;       // 1: two barriers per transform, 2: one
    int work;
    double a, b;
};

template <int V>
__device__ __forceinline__ void cluster_arrive_wait(unsigned *ctr, unsigned target, unsigned *status)
{
    // all threads: make this workgroup's stores visible, then one lane arrives and polls
    if (V == 1 || V == 2)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0)
    {
        if (V == 0)
        {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target)
        {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1u << 21) || __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            {
                __hip_atomic_fetch_add(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
        if (V == 0)
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

template <int V>
__global__ void __launch_bounds__(kT) k_cluster(ClusterArgs a)
{
    extern __shared__ uint64_t lds[];
    const unsigned bid = blockIdx.x, tid = threadIdx.x;
    const unsigned xcd = bid & 7, idx = bid >> 3;
    const unsigned per_xcd = gridDim.x >> 7; // clusters per XCD
    const unsigned cl = (idx >> 4) * 8 + xcd, rank = idx & 15;
    const unsigned ncl = per_xcd * 8;
    if (tid == 0)
    {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        a.status[2 + bid] = xcc & 15;
    }
    unsigned *ctr = a.counters + cl * 16;
    unsigned phase = 0;
    uint64_t *mid0 = a.mid + (size_t)cl * a.slots * kN;
    constexpr int ST = V == 1 ? 1 : 0, LD = V == 0 ? 0 : 1;
    unsigned it = 0;
    for (unsigned t = cl; t < a.transforms; t += ncl, it++)
    {
        uint64_t *mid = mid0 + (a.slots == 2 ? (it & 1) * kN : 0);
        pass1<LD, ST>(a.in + (size_t)t * kN, mid, rank, tid, a.work, a.a, a.b);
        cluster_arrive_wait<V>(ctr, 16 * ++phase, a.status);
        pass2<LD, ST>(mid, a.out + (size_t)t * kN, rank, tid, a.work, a.a, a.b);
        if (a.slots == 1)
            cluster_arrive_wait<2>(ctr, 16 * ++phase, a.status); // WAR only: reads are complete once consumed
    }
    if (a.a == 2.0)
        lds[tid] = 1;
}

__global__ void k_fill(uint64_t *in, size_t words)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x)
        in[i] = (i * 0x9E3779B97F4A7C15ull) >> 13;
}
__global__ void k_check(const uint64_t *in, const uint64_t *out, size_t transforms, unsigned long long *bad)
{
    // out[(hi*16 + cg)*256 + e*16 + c] == in[(e*16 + hi)*256 + cg*16 + c] + 2
    const size_t words = transforms * kN;
    unsigned long long nb = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x)
    {
        const size_t t = i >> 16, p = i & 65535;
        const unsigned hi = p >> 12, cg = (p >> 8) & 15, e = (p >> 4) & 15, c = p & 15;
        const uint64_t want = in[t * kN + (size_t)(e * 16 + hi) * 256 + cg * 16 + c] + 2;
        nb += out[i] != want;
    }
    if (nb)
        atomicAdd(bad, nb);
}

static unsigned long long verify(const uint64_t *in, uint64_t *out, size_t T, unsigned long long *d_bad)
{
    CK(hipMemset(d_bad, 0, 8));
    hipLaunchKernelGGL(k_check, dim3(4096), dim3(256), 0, 0, in, out, T, d_bad);
    unsigned long long bad;
    CK(hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost));
    CK(hipMemset(out, 0, T * kN * 8));
    return bad;
}

int main(int argc, char **argv)
{
    const size_t T = 480;
    const int reps = 5;
    const size_t lds_bytes = 36 * 1024; // as the real kernels: at most 4 workgroups per CU
    uint64_t *in, *out, *mid;
    unsigned long long *d_bad;
    CK(hipMalloc(&in, T * kN * 8));
    CK(hipMalloc(&out, T * kN * 8));
    CK(hipMalloc(&mid, T * kN * 8));
    CK(hipMalloc(&d_bad, 8));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, in, T * kN);
    CK(hipMemset(out, 0, T * kN * 8));
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const double alg = T * kN * 16.0;
    printf("480 transforms of 2^16 words; algorithmic bytes per run %.1f MB; GB/s = algorithmic\n", alg / 1e6);
    for (int work : { 0, 24, 48 })
    {
        printf("--- work = %d FMAs per word per pass\n", work);
        // A / B
        for (size_t chunk : { (size_t)480, (size_t)240, (size_t)120, (size_t)60, (size_t)30 })
        {
            float best = 1e9f;
            for (int r = 0; r < reps + 1; r++)
            {
                CK(hipEventRecord(e0));
                for (size_t t0 = 0; t0 < T; t0 += chunk)
                {
                    hipLaunchKernelGGL(k_p1, dim3(chunk * 16), dim3(kT), lds_bytes, 0, in + t0 * kN, mid, work, 1.0, 0.0);
                    hipLaunchKernelGGL(k_p2, dim3(chunk * 16), dim3(kT), lds_bytes, 0, mid, out + t0 * kN, work, 1.0, 0.0);
                }
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (r && ms < best)
                    best = ms;
            }
            unsigned long long bad = verify(in, out, T, d_bad);
            printf("two kernels, chunk %3zu transforms (mid %6.1f MB): %7.3f ms  %7.1f GB/s  bad=%llu\n", chunk, chunk * kN * 8 / 1e6, best,
                   alg / (best * 1e-3) / 1e9, bad);
        }
        // C
        for (unsigned wg_per_cu : { 2u, 4u })
            for (unsigned slots : { 1u, 2u })
                for (int V = 0; V < 3; V++)
                {
                    const unsigned grid = 256 * wg_per_cu;
                    const unsigned ncl = grid / 16;
                    unsigned *counters, *status;
                    CK(hipMalloc(&counters, ncl * 64));
                    CK(hipMalloc(&status, (2 + grid) * 4));
                    ClusterArgs a{ in, out, mid, counters, status, (unsigned)T, slots, work, 1.0, 0.0 };
                    float best = 1e9f;
                    unsigned lost = 0;
                    for (int r = 0; r < reps + 1; r++)
                    {
                        CK(hipMemset(counters, 0, ncl * 64));
                        CK(hipMemset(status, 0, (2 + grid) * 4));
                        CK(hipEventRecord(e0));
                        if (V == 0)
                            hipLaunchKernelGGL(k_cluster<0>, dim3(grid), dim3(kT), lds_bytes, 0, a);
                        else if (V == 1)
                            hipLaunchKernelGGL(k_cluster<1>, dim3(grid), dim3(kT), lds_bytes, 0, a);
                        else
                            hipLaunchKernelGGL(k_cluster<2>, dim3(grid), dim3(kT), lds_bytes, 0, a);
                        CK(hipEventRecord(e1));
                        CK(hipEventSynchronize(e1));
                        float ms;
                        CK(hipEventElapsedTime(&ms, e0, e1));
                        if (r && ms < best)
                            best = ms;
                        unsigned st;
                        CK(hipMemcpy(&st, status, 4, hipMemcpyDeviceToHost));
                        lost += st;
                        if (st)
                            break;
                    }
                    std::vector<unsigned> h(2 + grid);
                    CK(hipMemcpy(h.data(), status, (2 + grid) * 4, hipMemcpyDeviceToHost));
                    unsigned mixed = 0;
                    for (unsigned cl = 0; cl < ncl; cl++)
                    {
                        const unsigned xcd = cl & 7, j = cl >> 3;
                        bool mix = false;
                        for (unsigned r = 0; r < 16; r++)
                            mix |= h[2 + ((j * 16 + r) * 8 + xcd)] != h[2 + (j * 16 * 8 + xcd)];
                        mixed += mix;
                    }
                    unsigned long long bad = verify(in, out, T, d_bad);
                    printf("cluster V%d  %u WG/CU  %u slot(s) (mid %5.1f MB): %7.3f ms  %7.1f GB/s  bad=%llu lost_barriers=%u mixed_xcd_clusters=%u/%u\n", V,
                           wg_per_cu, slots, ncl * slots * kN * 8 / 1e6, best, alg / (best * 1e-3) / 1e9, bad, lost, mixed, ncl);
                    CK(hipFree(counters));
                    CK(hipFree(status));
                    if (lost)
                    {
                        printf("  (barrier lost: skipping remaining variants of this shape)\n");
                        break;
                    }
                }
    }
    return 0;
}
