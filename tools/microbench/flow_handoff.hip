// Two-pass 2^16-point transform with the intermediate handed over through the XCD's L2 by DATAFLOW instead of barriers.
//
// One persistent launch.  Transforms are dealt to the 8 XCDs (t % 8); every workgroup reads its XCC id and pulls work
// items from ITS XCD's ordered queue (one atomic counter per XCD).  Items of one XCD, in queue order, ring of R slots:
//     step k:  p1(k) [16 tiles], p2(k - d) [16 tiles]      (p2 runs d steps behind p1, 0 < d < R)
// p2(u) waits until the 16 p1 tiles of u have arrived (done1[u] == 16), p1(u+R) until the 16 p2 tiles of u have read the
// slot (done2[u] == 16).  Every dependency points to an EARLIER queue entry and an item is only held by a running
// workgroup, so the lowest unfinished item can always run: no deadlock whatever the dispatch order or residency.
// Visibility inside one XCD: plain stores + s_waitcnt vmcnt(0) + agent-scope atomic arrive; the consumer polls the counter
// with a relaxed agent load and reads the slot with sc1 loads (L1 bypass: the slot is reused, its L1 lines would be stale).
// Same memory shapes as ntt2_fwd_p1<8> / ntt2_fwd_p2<8> (4096-word tiles, 256 threads x 16 words), `work` FMAs per word per
// pass standing in for the butterflies; every word is verified.  Spin loops are capped: a lost dependency is reported.
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

template <bool NT>
__device__ __forceinline__ uint64_t ld_stream(const uint64_t *p)
{
    return NT ? __builtin_nontemporal_load(p) : *p;
}
template <bool NT>
__device__ __forceinline__ void st_stream(uint64_t *p, uint64_t v)
{
    if (NT)
        __builtin_nontemporal_store(v, p);
    else
        *p = v;
}

// pass 1, column tile cg: thread (c = tid & 15, hi = tid >> 4) reads (row e*16 + hi, col cg*16 + c), writes tile order
template <bool NT>
__device__ __forceinline__ void pass1(const uint64_t *in, uint64_t *mid, unsigned cg, unsigned tid, int work, double a, double b)
{
    const unsigned c = tid & 15, hi = tid >> 4;
    uint64_t x[16];
#pragma unroll
    for (int e = 0; e < 16; e++)
        x[e] = ld_stream<NT>(in + (size_t)(e * 16 + hi) * 256 + cg * 16 + c);
    work_on(x, work, a, b);
    uint64_t *o = mid + (size_t)(hi * 16 + cg) * 256 + c;
#pragma unroll
    for (int e = 0; e < 16; e++)
        o[e * 16] = x[e];
}
// pass 2, row tile hg: contiguous 32 KiB in (LD: 0 plain, 1 sc1), contiguous 32 KiB out
template <int LD, bool NT>
__device__ __forceinline__ void pass2(const uint64_t *mid, uint64_t *out, unsigned hg, unsigned tid, int work, double a, double b)
{
    uint64_t x[16];
    const uint64_t *m = mid + (size_t)hg * 4096 + tid;
#pragma unroll
    for (int e = 0; e < 16; e++)
        x[e] = LD ? __hip_atomic_load(m + e * 256, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : m[e * 256];
    work_on(x, work, a, b);
    uint64_t *o = out + (size_t)hg * 4096 + tid;
#pragma unroll
    for (int e = 0; e < 16; e++)
        st_stream<NT>(o + e * 256, x[e]);
}

__global__ void __launch_bounds__(kT) k_p1(const uint64_t *in, uint64_t *mid, int work, double a, double b)
{
    extern __shared__ uint64_t lds[];
    const size_t t = blockIdx.x >> 4;
    pass1<false>(in + t * kN, mid + t * kN, blockIdx.x & 15, threadIdx.x, work, a, b);
    if (a == 2.0)
        lds[threadIdx.x] = 1;
}
__global__ void __launch_bounds__(kT) k_p2(const uint64_t *mid, uint64_t *out, int work, double a, double b)
{
    extern __shared__ uint64_t lds[];
    const size_t t = blockIdx.x >> 4;
    pass2<0, false>(mid + t * kN, out + t * kN, blockIdx.x & 15, threadIdx.x, work, a, b);
    if (a == 2.0)
        lds[threadIdx.x] = 1;
}

struct FlowArgs
{
    const uint64_t *in;
    uint64_t *out;
    uint64_t *mid;      // [8][R][kN]
    unsigned *heads;    // [8] queue heads, 64-byte spaced
    unsigned *done1;    // [8][U]
    unsigned *done2;    // [8][U]
    unsigned *status;   // [0] lost dependencies, [1..8] workgroups seen per XCD, [9] total stall polls
    unsigned transforms; // multiple of 8
    unsigned R, d; // ring slots; p2 runs d steps behind p1 (0 < d < R)
    int work;
    double a, b;
};

__device__ __forceinline__ bool wait_for(unsigned *flag, unsigned target, unsigned *status)
{
    unsigned spins = 0;
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target)
    {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1u << 20) || __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        {
            __hip_atomic_fetch_add(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
    }
    if (spins)
        __hip_atomic_fetch_add(status + 9, spins, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
}

template <bool NT>
__global__ void __launch_bounds__(kT) k_flow(FlowArgs a)
{
    extern __shared__ uint64_t lds[];
    __shared__ unsigned s_item, s_ok;
    const unsigned tid = threadIdx.x;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7;
    const unsigned U = a.transforms / 8, R = a.R;
    const unsigned items = 32 * (U + a.d);
    unsigned *head = a.heads + xcc * 16;
    unsigned *d1 = a.done1 + (size_t)xcc * U, *d2 = a.done2 + (size_t)xcc * U;
    uint64_t *ring = a.mid + (size_t)xcc * R * kN;
    if (tid == 0)
    {
        atomicAdd(a.status + 1 + xcc, 1u);
        s_item = __hip_atomic_fetch_add(head, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    unsigned item = s_item;
    while (item < items)
    {
        // decode: step k = item / 32 holds the 16 tiles of p1(k) and then the 16 tiles of p2(k - d)
        unsigned kind, u, tile;
        {
            const unsigned k = item >> 5, w = item & 31;
            if (w < 16)
            {
                kind = 1;
                u = k; // >= U in the last d steps: nothing to do
                tile = w;
            }
            else
            {
                kind = 2;
                u = k - a.d; // wraps to a huge value in the first d steps: nothing to do
                tile = w - 16;
            }
        }
        __syncthreads(); // everyone has read s_item
        if (tid == 0)
        {
            // next item (pulled ahead: its latency hides behind this item's work) and this item's dependency
            s_item = __hip_atomic_fetch_add(head, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bool ok = true;
            if (u < U)
            {
                if (kind == 2)
                    ok = wait_for(d1 + u, 16, a.status);
                else if (u >= R)
                    ok = wait_for(d2 + (u - R), 16, a.status);
            }
            s_ok = ok;
        }
        __syncthreads();
        if (!s_ok)
            break;
        if (u < U)
        {
            const size_t t = (size_t)u * 8 + xcc;
            uint64_t *slot = ring + (size_t)(u % R) * kN;
            if (kind == 1)
            {
                pass1<NT>(a.in + t * kN, slot, tile, tid, a.work, a.a, a.b);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this thread's slot stores have reached the L2
                __syncthreads();
                if (tid == 0)
                    __hip_atomic_fetch_add(d1 + u, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            else
            {
                pass2<1, NT>(slot, a.out + t * kN, tile, tid, a.work, a.a, a.b);
                // the slot's words are in registers once consumed by work_on: arrive after the loads have returned
                __syncthreads();
                if (tid == 0)
                    __hip_atomic_fetch_add(d2 + u, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        item = s_item;
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
    const size_t T = argc > 1 ? atoi(argv[1]) : 3840; // 2 GB in + 2 GB out: far beyond the 256 MiB Infinity Cache
    const int reps = 4;
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
    printf("%zu transforms of 2^16 words; algorithmic bytes per run %.1f MB; GB/s = algorithmic\n", T, alg / 1e6);
    unsigned *heads, *done1, *done2, *status;
    const unsigned U = T / 8;
    CK(hipMalloc(&heads, 8 * 64));
    CK(hipMalloc(&done1, 8 * U * 4));
    CK(hipMalloc(&done2, 8 * U * 4));
    CK(hipMalloc(&status, 64));
    for (int work : { 0, 24 })
    {
        printf("--- work = %d FMAs per word per pass\n", work);
        for (size_t chunk : { T, (size_t)240, (size_t)120 })
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
            printf("two kernels, chunk %4zu transforms (mid %7.1f MB): %7.3f ms  %7.1f GB/s  bad=%llu\n", chunk, chunk * kN * 8 / 1e6, best,
                   alg / (best * 1e-3) / 1e9, bad);
        }
        for (int nt = 0; nt < 2; nt++)
            for (unsigned wg_per_cu : { 1u, 2u, 3u })
                for (unsigned R : { 4u, 6u, 8u, 12u })
                {
                    const unsigned grid = 256 * wg_per_cu;
                    FlowArgs a{ in, out, mid, heads, done1, done2, status, (unsigned)T, R, R / 2, work, 1.0, 0.0 };
                    float best = 1e9f;
                    unsigned lost = 0, polls = 0;
                    std::vector<unsigned> h(16);
                    for (int r = 0; r < reps + 1; r++)
                    {
                        CK(hipMemset(heads, 0, 8 * 64));
                        CK(hipMemset(done1, 0, 8 * U * 4));
                        CK(hipMemset(done2, 0, 8 * U * 4));
                        CK(hipMemset(status, 0, 64));
                        CK(hipEventRecord(e0));
                        if (nt)
                            hipLaunchKernelGGL(k_flow<true>, dim3(grid), dim3(kT), lds_bytes, 0, a);
                        else
                            hipLaunchKernelGGL(k_flow<false>, dim3(grid), dim3(kT), lds_bytes, 0, a);
                        CK(hipEventRecord(e1));
                        CK(hipEventSynchronize(e1));
                        float ms;
                        CK(hipEventElapsedTime(&ms, e0, e1));
                        if (r && ms < best)
                            best = ms;
                        CK(hipMemcpy(h.data(), status, 64, hipMemcpyDeviceToHost));
                        lost += h[0];
                        polls = h[9];
                        if (h[0])
                            break;
                    }
                    unsigned long long bad = verify(in, out, T, d_bad);
                    printf("flow nt=%d  %u WG/CU  ring %u (mid %5.1f MB): %7.3f ms  %7.1f GB/s  bad=%llu lost=%u stall_polls=%u wg/xcd=%u..%u\n", nt, wg_per_cu, R,
                           8.0 * R * kN * 8 / 1e6, best, alg / (best * 1e-3) / 1e9, bad, lost, polls, h[1], h[8]);
                    if (lost)
                        break;
                }
    }
    return 0;
}
