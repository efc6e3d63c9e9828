// Does taking the prefetch out of the register file (gfx950 LDS-DMA: global_load_lds_dwordx4) or wave priorities (s_setprio) buy the
// transform passes / the key switch anything?  (VERDICT r4, next #1.)
//
// The passes of the two-pass engine are tile copies with arithmetic in between: a workgroup of 256 threads reads a 32 KiB tile
// (16 words per thread), works on it, writes a 32 KiB tile, and keeps the NEXT tile's 16 loads in flight in 32 VGPRs while it
// works (ntt2_kernels.hip: fwd_p1_body / fwd_p2_body; ks2 does the same with a digit + its key words).  This microbenchmark keeps
// the memory shape, the persistent loop, the LDS footprint that limits the real kernels' occupancy and a dial for the arithmetic
// (`work` dependent FMAs per word, as tools/microbench/ks_flow.hip), and swaps the way the next tile travels:
//   reg        next tile in 32 VGPRs (today)
//   reg-spread the same loads issued in four groups between quarters of the arithmetic instead of back to back
//   dma1       next tile lands in a wave-private 8 KiB LDS slot by 8 global_load_lds_dwordx4 per wave; read out by 16 ds_read_b64
//              when its turn comes, then the slot is re-armed (one tile ahead, as `reg`, but no VGPR is held)
//   dma2       two slots per wave: two tiles ahead (what the freed registers cannot buy `reg`: its LDS does)
//   prio-c / prio-l   `reg` with s_setprio 1 around the arithmetic / around the load issue (the other workgroup of the CU is
//              in the other phase most of the time)
// every variant at the occupancy its resources allow next to the real kernels' exchange buffers (36.8 KiB per workgroup; a slot
// set is 32 KiB per workgroup), and once with the exchange buffer taken away (what LDS-DMA could do if the exchange were free).
// Output: read + write GB/s; every variant's output is compared with the input.
// build: hipcc -O3 --offload-arch=gfx950 ldsdma_pass.hip -o ldsdma_pass        usage: ldsdma_pass [GiB per direction (default 4)]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int kT = 256;
constexpr size_t kTileWords = 4096;

struct Args
{
    const uint64_t *in;
    uint64_t *out;
    size_t tiles;
    int work;
    double one, zero;
};

__device__ __forceinline__ void chain_part(double (&d)[16], int n, double a, double b)
{
    for (int i = 0; i < n; i++)
    {
#pragma unroll
        for (int e = 0; e < 16; e++)
            d[e] = __builtin_fma(d[e], a, b); // a = 1, b = 0 at run time: the value survives
    }
}

// word e of thread (wave w, lane) of a tile: w*1024 + e*64 + lane (a wave instruction = 512 contiguous bytes, as pass 2's loads)
__device__ __forceinline__ size_t word_of(unsigned tid, int e) { return (size_t)(tid >> 6) * 1024 + (size_t)e * 64 + (tid & 63); }

template <int PRIO> // 0 none, 1 around the arithmetic, 2 around the load issue
__global__ void __launch_bounds__(kT) k_reg(Args a)
{
    extern __shared__ uint64_t lds[];
    const unsigned tid = threadIdx.x;
    size_t t = blockIdx.x;
    if (t >= a.tiles)
        return;
    uint64_t nxt[16];
    auto fetch = [&](size_t tile) {
        const uint64_t *p = a.in + tile * kTileWords;
#pragma unroll
        for (int e = 0; e < 16; e++)
            nxt[e] = __builtin_nontemporal_load(p + word_of(tid, e));
    };
    fetch(t);
    for (; t < a.tiles; t += gridDim.x)
    {
        double x[16];
#pragma unroll
        for (int e = 0; e < 16; e++)
            x[e] = __builtin_bit_cast(double, nxt[e]);
        if (PRIO == 2)
            __builtin_amdgcn_s_setprio(1);
        if (t + gridDim.x < a.tiles)
            fetch(t + gridDim.x);
        if (PRIO == 2)
            __builtin_amdgcn_s_setprio(0);
        if (PRIO == 1)
            __builtin_amdgcn_s_setprio(1);
        chain_part(x, a.work, a.one, a.zero);
        if (PRIO == 1)
            __builtin_amdgcn_s_setprio(0);
        uint64_t *q = a.out + t * kTileWords;
#pragma unroll
        for (int e = 0; e < 16; e++)
            __builtin_nontemporal_store(__builtin_bit_cast(uint64_t, x[e]), q + word_of(tid, e));
    }
    if (a.work < 0)
        lds[tid] = 0; // (keeps the dynamic allocation referenced)
}

// the loads of the next tile in four groups, one before each quarter of the arithmetic
__global__ void __launch_bounds__(kT) k_reg_spread(Args a)
{
    extern __shared__ uint64_t lds[];
    const unsigned tid = threadIdx.x;
    size_t t = blockIdx.x;
    if (t >= a.tiles)
        return;
    uint64_t nxt[16];
    {
        const uint64_t *p = a.in + t * kTileWords;
#pragma unroll
        for (int e = 0; e < 16; e++)
            nxt[e] = __builtin_nontemporal_load(p + word_of(tid, e));
    }
    const int q4 = a.work / 4, rest = a.work - 3 * q4;
    for (; t < a.tiles; t += gridDim.x)
    {
        double x[16];
#pragma unroll
        for (int e = 0; e < 16; e++)
            x[e] = __builtin_bit_cast(double, nxt[e]);
        const bool more = t + gridDim.x < a.tiles;
        const uint64_t *p = a.in + (more ? t + gridDim.x : t) * kTileWords;
#pragma unroll
        for (int g = 0; g < 4; g++)
        {
#pragma unroll
            for (int e = 4 * g; e < 4 * g + 4; e++)
                nxt[e] = __builtin_nontemporal_load(p + word_of(tid, e));
            __builtin_amdgcn_sched_barrier(0);
            chain_part(x, g == 3 ? rest : q4, a.one, a.zero);
            __builtin_amdgcn_sched_barrier(0);
        }
        uint64_t *q = a.out + t * kTileWords;
#pragma unroll
        for (int e = 0; e < 16; e++)
            __builtin_nontemporal_store(__builtin_bit_cast(uint64_t, x[e]), q + word_of(tid, e));
    }
    if (a.work < 0)
        lds[tid] = 0;
}

// LDS-DMA: the wave's 8 KiB of a tile land at slot + i*1024 + lane*16 by 8 wave instructions of 1 KiB
typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef const __attribute__((address_space(1))) void *glb_ptr_t;
__device__ __forceinline__ void dma_tile(const uint64_t *tile, uint64_t *slot, unsigned tid)
{
    const unsigned w = (unsigned)__builtin_amdgcn_readfirstlane((int)(tid >> 6)), lane = tid & 63;
    const char *src = reinterpret_cast<const char *>(tile) + (size_t)w * 8192 + lane * 16;
#pragma unroll
    for (int i = 0; i < 8; i++)
        __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + i * 1024), (lds_ptr_t)(reinterpret_cast<char *>(slot) + i * 1024), 16, 0, 2 /* nt */);
}
// wait until at most `younger` of this wave's vector-memory instructions are outstanding (the immediate must be a constant)
__device__ __forceinline__ void wait_vm(unsigned younger)
{
    switch (younger)
    {
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
    case 24: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
    case 40: asm volatile("s_waitcnt vmcnt(40)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}
// SLOTS wave-private slots of 8 KiB per wave; slot s of wave w at lds + (s*4 + w)*1024 words.  A tile's DMA has landed when
// everything this wave issued BEFORE and INCLUDING it has completed (vmcnt counts loads, DMAs and stores in issue order): the
// wait allows exactly the instructions issued after it (`issued - mark[s]`: the later DMAs and the stores in between)
template <int SLOTS>
__global__ void __launch_bounds__(kT) k_dma(Args a)
{
    extern __shared__ uint64_t lds[];
    const unsigned tid = threadIdx.x, lane = tid & 63;
    const unsigned w = (unsigned)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    size_t t = blockIdx.x;
    if (t >= a.tiles)
        return;
    unsigned issued = 0, mark[SLOTS];
    // prologue: SLOTS tiles in flight
#pragma unroll
    for (int s = 0; s < SLOTS; s++)
    {
        if (t + (size_t)s * gridDim.x < a.tiles)
        {
            dma_tile(a.in + (t + (size_t)s * gridDim.x) * kTileWords, lds + (size_t)(s * 4 + w) * 1024, tid);
            issued += 8;
        }
        mark[s] = issued;
    }
    for (unsigned it = 0; t < a.tiles; t += gridDim.x, it++)
    {
#pragma unroll
        for (int s = 0; s < SLOTS; s++)
        {
            if ((int)(it % SLOTS) != s)
                continue;
            wait_vm(issued - mark[s]);
            uint64_t *slot = lds + (size_t)(s * 4 + w) * 1024;
            double x[16];
#pragma unroll
            for (int e = 0; e < 16; e++)
                x[e] = __builtin_bit_cast(double, slot[e * 64 + lane]); // word (w, e, lane) sits at byte e*512 + lane*8 of the wave's run
            // (the ds_reads above must have returned before the slot is re-armed: the DMA writes the same addresses)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (t + (size_t)SLOTS * gridDim.x < a.tiles)
            {
                dma_tile(a.in + (t + (size_t)SLOTS * gridDim.x) * kTileWords, slot, tid);
                issued += 8;
            }
            mark[s] = issued;
            chain_part(x, a.work, a.one, a.zero);
            uint64_t *q = a.out + t * kTileWords;
#pragma unroll
            for (int e = 0; e < 16; e++)
                __builtin_nontemporal_store(__builtin_bit_cast(uint64_t, x[e]), q + word_of(tid, e));
            issued += 16;
        }
    }
}

__global__ void k_fill(uint64_t *p, size_t words)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x)
        p[i] = __builtin_bit_cast(uint64_t, (double)(int64_t)((i * 0x9E3779B97F4A7C15ull) >> 40));
}
__global__ void k_diff(const uint64_t *x, const uint64_t *y, size_t words, unsigned long long *bad)
{
    unsigned long long nb = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x)
        nb += x[i] != y[i];
    if (nb)
        atomicAdd(bad, nb);
}

int main(int argc, char **argv)
{
    const size_t bytes = (size_t)(argc > 1 ? atoi(argv[1]) : 4) << 30;
    const size_t words = bytes / 8, tiles = words / kTileWords;
    uint64_t *in, *out;
    unsigned long long *d_bad;
    CK(hipMalloc(&in, bytes));
    CK(hipMalloc(&out, bytes));
    CK(hipMalloc(&d_bad, 8));
    hipLaunchKernelGGL(k_fill, dim3(8192), dim3(256), 0, 0, in, words);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    int cus = 256;
    {
        hipDeviceProp_t p;
        CK(hipGetDeviceProperties(&p, 0));
        cus = p.multiProcessorCount;
        printf("%s, %d CUs; %.1f GiB per direction, %zu tiles of 32 KiB\n", p.name, cus, bytes / 1073741824.0, tiles);
    }
    const size_t kExch = 36864 + 2048; // the exchange buffers + staged twiddles of the real passes
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_dma<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_dma<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_reg<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_reg<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_reg<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_reg_spread), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    auto run = [&](const char *name, auto kern, size_t lds_bytes, int wgs_per_cu, int work) {
        Args a{ in, out, tiles, work, 1.0, 0.0 };
        const unsigned grid = (unsigned)(cus * wgs_per_cu);
        // the grid has wgs_per_cu workgroups per CU; make the LDS request large enough that no CU can take one more
        const size_t excl = 160 * 1024 / (wgs_per_cu + 1) + 1024;
        if (lds_bytes < excl && excl * wgs_per_cu <= 160 * 1024)
            lds_bytes = excl;
        CK(hipMemset(out, 0xff, bytes));
        float best = 1e9f, sum = 0;
        const int reps = 4;
        for (int r = 0; r < reps + 1; r++)
        {
            float ms;
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(kern, dim3(grid), dim3(kT), lds_bytes, 0, a);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipGetLastError());
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (r)
            {
                best = ms < best ? ms : best;
                sum += ms;
            }
        }
        CK(hipMemset(d_bad, 0, 8));
        hipLaunchKernelGGL(k_diff, dim3(4096), dim3(256), 0, 0, in, out, words, d_bad);
        unsigned long long bad;
        CK(hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost));
        printf("  %-58s LDS %6.1f KiB  %d WG/CU  best %7.3f ms %7.1f GB/s  mean %7.1f GB/s  bad=%llu\n", name, lds_bytes / 1024.0, wgs_per_cu, best,
               2.0 * bytes / (best * 1e-3) / 1e9, 2.0 * bytes / (sum / reps * 1e-3) / 1e9, bad);
    };
    for (int work : { 0, 24, 48, 96 })
    {
        printf("--- work = %d dependent FMAs per word (the plain passes issue ~10-12 VALU per word and direction, ks2 ~55 per word)\n", work);
        // today's shape: register prefetch next to the exchange buffer; 4 workgroups per CU (128 VGPRs) and 2 (ks2: 256 VGPRs)
        run("reg, next tile in VGPRs", k_reg<0>, kExch, 4, work);
        run("reg, next tile in VGPRs", k_reg<0>, kExch, 2, work);
        run("reg, next tile in VGPRs, 3 per CU", k_reg<0>, kExch, 3, work);
        run("reg-spread, loads in four groups between the arithmetic", k_reg_spread, kExch, 4, work);
        run("reg-spread, loads in four groups between the arithmetic", k_reg_spread, kExch, 2, work);
        run("prio-c, s_setprio 1 around the arithmetic", k_reg<1>, kExch, 4, work);
        run("prio-c, s_setprio 1 around the arithmetic", k_reg<1>, kExch, 2, work);
        run("prio-l, s_setprio 1 around the load issue", k_reg<2>, kExch, 4, work);
        run("prio-l, s_setprio 1 around the load issue", k_reg<2>, kExch, 2, work);
        // LDS-DMA next to the exchange buffer: 32 KiB of slots + 38 KiB = 70 KiB -> 2 workgroups per CU whatever the VGPRs
        run("dma1, one tile ahead in LDS (+ exchange buffer)", k_dma<1>, kExch + 32768, 2, work);
        // two tiles ahead: 102 KiB -> 1 workgroup per CU
        run("dma2, two tiles ahead in LDS (+ exchange buffer)", k_dma<2>, kExch + 65536, 1, work);
        // if the exchange buffer cost nothing: slots only
        run("dma1, slots only (no exchange buffer)", k_dma<1>, 32768, 4, work);
        run("dma1, slots only (no exchange buffer)", k_dma<1>, 32768, 2, work);
        run("dma2, slots only (no exchange buffer)", k_dma<2>, 65536, 2, work);
    }
    return 0;
}
