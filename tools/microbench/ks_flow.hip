// Would the fused key switch gain from handing the pass-1 tiles to pass 2 ON CHIP?
//
// The real kernels (ntt2_kernels.hip: ks1_kernel / ks2_kernel) write K(K+1) half-transformed digits per ciphertext to HBM and read
// them back: 2 x 125.8 MB per C5 ciphertext, 64 GB per 256-ciphertext step.  With every load and store of the two kernels replaced
// by register values (-DSEALHIP_KS_NOMEM build, profiles/r04_ks_handover.txt) the key switch takes 14.0 ms instead of 19.2: that is
// what perfect on-chip hand-over could save at most.  This microbenchmark measures what a REAL hand-over structure keeps of it, with
// the memory shapes, grid orders, prefetch depth and instruction weight of the real kernels but no modular arithmetic:
//   A. two kernels (today): k1 = one workgroup per (column tile, target I, item b) looping over the 15 digits, next digit in flight,
//      writes tile order; k2 = one workgroup per (target I, row tile, item b) looping over the digits, next digit + this digit's key
//      words in flight, 2 x 16 running sums per thread, XCD-ordered so that a key tile is served by one XCD's L2 for the whole batch;
//   C. one persistent kernel: the 16 workgroups of a team (same XCD) own one (b, I): for every digit each runs pass 1 on its column
//      tile into a team-private slot, the team meets at a barrier in L2, each runs pass 2 on its row tile of the slot and adds into
//      its sums.  Only the sums leave the chip - but the key (252 MB) and the digits are now re-read per team instead of per XCD:
//      an XCD's four teams are 2 items x 2 targets, XCD x always serves targets 2x and 2x+1 (its 31 MB of the key).
// `work1` / `work2` dependent FMAs per word stand in for the butterflies (the real kernels issue 40.9 / 54.5 VALU instructions per
// coefficient and target-digit pair); every variant's sums are compared with variant A's.  Spin loops are capped.
// build: hipcc -O3 --offload-arch=gfx950 ks_flow.hip -o ks_flow       usage: ks_flow [items (default 64)]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int kT = 256;
constexpr size_t kN = 65536;
constexpr unsigned kDigits = 15, kTargets = 16;

__device__ __forceinline__ void chain(double (&d)[16], int work, double a, double b)
{
    for (int i = 0; i < work; i++)
    {
#pragma unroll
        for (int e = 0; e < 16; e++)
            d[e] = __builtin_fma(d[e], a, b); // a = 1, b = 0 at run time (opaque to the compiler): the value survives
    }
}

// pass 1 of digit J for target I: thread (c = tid & 15, hi = tid >> 4) owns (row e*16 + hi, col cg*16 + c), writes tile order
struct P1
{
    double x[16];
    __device__ __forceinline__ void load(const uint64_t *digit, unsigned cg, unsigned tid)
    {
        const unsigned c = tid & 15, hi = tid >> 4;
#pragma unroll
        for (int e = 0; e < 16; e++)
            x[e] = (double)(uint32_t)digit[(size_t)(e * 16 + hi) * 256 + cg * 16 + c];
    }
};
template <int ST>
__device__ __forceinline__ void p1_store(const double (&x)[16], uint64_t *mid, unsigned cg, unsigned tid, unsigned I)
{
    const unsigned c = tid & 15, hi = tid >> 4;
    uint64_t *o = mid + (size_t)(hi * 16 + cg) * 256 + c;
#pragma unroll
    for (int e = 0; e < 16; e++)
    {
        const uint64_t v = __builtin_bit_cast(uint64_t, x[e] + (double)I); // (the target enters the value: slots are not interchangeable)
        if (ST == 1)
            __hip_atomic_store(o + e * 16, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else
            o[e * 16] = v;
    }
}
template <int LD>
__device__ __forceinline__ void p2_load(uint64_t (&n)[16], const uint64_t *mid, unsigned hg, unsigned tid)
{
    const uint64_t *m = mid + (size_t)hg * 4096 + tid;
#pragma unroll
    for (int e = 0; e < 16; e++)
    {
        if (LD == 1)
            n[e] = __hip_atomic_load(m + e * 256, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else
            n[e] = m[e * 256];
    }
}
// the key words of (J, I), register order in pairs: coefficient (hg, e, tid) at ((hg*16 + e)*256 + tid) * 2
__device__ __forceinline__ void key_load(double (&k0)[16], double (&k1)[16], const uint64_t *key, unsigned J, unsigned I, unsigned hg, unsigned tid)
{
    const ulonglong2 *kp = reinterpret_cast<const ulonglong2 *>(key + ((size_t)(J * kTargets + I) * 2 << 16)) + (size_t)hg * 4096 + tid;
#pragma unroll
    for (int e = 0; e < 16; e++)
    {
        const ulonglong2 w = kp[e * 256];
        k0[e] = (double)(uint32_t)w.x;
        k1[e] = (double)(uint32_t)w.y;
    }
}
__device__ __forceinline__ void mac(double (&a0)[16], double (&a1)[16], const double (&x)[16], const double (&k0)[16], const double (&k1)[16], double one,
                                    double zero)
{
#pragma unroll
    for (int e = 0; e < 16; e++)
    {
        // an exact product costs seven instructions in the real kernel: two FMAs carry the value, five more stand for the rest
        double p0 = x[e] * k0[e], p1 = x[e] * k1[e];
        p0 = __builtin_fma(p0, one, zero), p1 = __builtin_fma(p1, one, zero);
        p0 = __builtin_fma(p0, one, zero), p1 = __builtin_fma(p1, one, zero);
        a0[e] = __builtin_fma(a0[e], one, p0);
        a1[e] = __builtin_fma(a1[e], one, p1);
    }
}
__device__ __forceinline__ void sums_store(const double (&a0)[16], const double (&a1)[16], uint64_t *out, unsigned b, unsigned I, unsigned hg, unsigned tid)
{
    uint64_t *o = out + ((size_t)(b * kTargets + I) * 2 << 16) + (size_t)hg * 4096 + tid;
#pragma unroll
    for (int e = 0; e < 16; e++)
    {
        o[e * 256] = __builtin_bit_cast(uint64_t, a0[e]);
        o[kN + e * 256] = __builtin_bit_cast(uint64_t, a1[e]);
    }
}

struct Args
{
    const uint64_t *digits; // [items][15][N]
    const uint64_t *key;    // [15][16][2N]
    uint64_t *mid;          // A: [items][16][15][N]   C: [teams][slots][N]
    uint64_t *out;          // [items][16][2][N]
    unsigned items;
    int work1, work2;
    double one, zero;
    unsigned *counters, *status;
    unsigned slots;
};

// where the half-transformed digit (b, I, J) lives and how far its sixteen 4096-word row tiles are apart.
// LAYOUT 0 (today): [b][I][J][N] - the 64 workgroups an XCD runs at a time (one (I, tile), consecutive items b) read tiles 126 MB apart;
// LAYOUT 1: [I][J][tile][b][4096] - the same 64 workgroups read ONE contiguous 2 MiB run per digit
template <int LAYOUT>
__device__ __forceinline__ size_t mid_tile(unsigned items, unsigned b, unsigned I, unsigned J, unsigned hg)
{
    if (LAYOUT == 0)
        return ((size_t)((b * kTargets + I) * kDigits + J) << 16) + ((size_t)hg << 12);
    return ((size_t)(((I * kDigits + J) * 16 + hg) * items + b)) << 12;
}

// ---- A: today's two kernels
template <int LAYOUT, bool DIG>
__global__ void __launch_bounds__(kT, 4) k1(Args a)
{
    const unsigned bid = blockIdx.x, tid = threadIdx.x;
    const unsigned low = bid & 7, rest = bid >> 3;
    const unsigned I = rest % kTargets, grp = (rest / kTargets) * 8 + low; // blocks sharing (b, cg) - the same digit tiles - sit on one XCD
    if (grp >= a.items * 16)
        return;
    const unsigned b = grp / 16, cg = grp % 16;
    P1 cur, nxt;
    nxt.load(a.digits + ((size_t)(b * kDigits) << 16), cg, tid);
    for (unsigned J = 0; J < kDigits; J++)
    {
        cur = nxt;
        if (DIG && J + 1 < kDigits)
            nxt.load(a.digits + ((size_t)(b * kDigits + J + 1) << 16), cg, tid);
        chain(cur.x, a.work1, a.one, a.zero);
        if (LAYOUT == 0)
            p1_store<0>(cur.x, a.mid + ((size_t)((b * kTargets + I) * kDigits + J) << 16), cg, tid, I);
        else
        {
            // the thread's sixteen rows e*16 + hi belong to row tile hg = e: sixteen 128-byte runs in sixteen different tile slabs
            const unsigned c = tid & 15, hi = tid >> 4;
#pragma unroll
            for (int e = 0; e < 16; e++)
            {
                // NOTE tile-order inside a slab: (hi within tile = hi... ) the slab of row tile `hi` holds (hi*16 + cg)*256 + e*16 + c in LAYOUT 0;
                // here slab = hi, offset inside = cg*256 + e*16 + c
                a.mid[mid_tile<1>(a.items, b, I, J, hi) + cg * 256 + e * 16 + c] = __builtin_bit_cast(uint64_t, cur.x[e] + (double)I);
            }
        }
    }
}
template <int LAYOUT, bool KEY>
__global__ void __launch_bounds__(kT, 2) k2(Args a)
{
    const unsigned bid = blockIdx.x, tid = threadIdx.x;
    const unsigned xcd = bid & 7, rest = bid >> 3;
    const unsigned b = rest % a.items, tile = (rest / a.items) * 8 + xcd; // every item of one (I, hg) on one XCD, back to back
    if (tile >= kTargets * 16)
        return;
    const unsigned I = tile / 16, hg = tile % 16;
    double a0[16], a1[16];
#pragma unroll
    for (int e = 0; e < 16; e++)
        a0[e] = a1[e] = 0.0;
    uint64_t nxt[16];
    p2_load<0>(nxt, a.mid + mid_tile<LAYOUT>(a.items, b, I, 0, hg), 0, tid);
    for (unsigned J = 0; J < kDigits; J++)
    {
        double x[16], k0[16], k1[16];
#pragma unroll
        for (int e = 0; e < 16; e++)
            x[e] = __builtin_bit_cast(double, nxt[e]);
        if (KEY)
            key_load(k0, k1, a.key, J, I, hg, tid);
        else
        {
#pragma unroll
            for (int e = 0; e < 16; e++)
                k0[e] = a.one + (double)J, k1[e] = a.one + (double)e;
        }
        if (J + 1 < kDigits)
            p2_load<0>(nxt, a.mid + mid_tile<LAYOUT>(a.items, b, I, J + 1, hg), 0, tid);
        chain(x, a.work2, a.one, a.zero);
        mac(a0, a1, x, k0, k1, a.one, a.zero);
    }
    sums_store(a0, a1, a.out, b, I, hg, tid);
}

// ---- C: persistent teams, hand-over through L2
template <int V>
__device__ __forceinline__ void team_arrive_wait(unsigned *ctr, unsigned target, unsigned *status)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0)
    {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target)
        {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 22) || __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            {
                __hip_atomic_fetch_add(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    __syncthreads();
}
// V = 1: sc1 stores + sc1 loads (valid wherever the team's workgroups run); V = 2: plain stores, sc1 loads (valid when the team shares an L2)
template <int V, bool PF>
__global__ void __launch_bounds__(kT, 2) k_team(Args a)
{
    const unsigned bid = blockIdx.x, tid = threadIdx.x;
    const unsigned xcd = bid & 7, idx = bid >> 3;
    const unsigned j = idx >> 4, rank = idx & 15; // j = team of this XCD (0..3), rank = tile of the team
    const unsigned team = j * 8 + xcd;
    const unsigned I = 2 * xcd + (j & 1);
    if (tid == 0)
    {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        a.status[2 + bid] = xcc & 15;
    }
    unsigned *ctr = a.counters + team * 16;
    unsigned phase = 0;
    uint64_t *slot0 = a.mid + (size_t)team * a.slots * kN;
    constexpr int ST = V == 1 ? 1 : 0;
    unsigned step = 0;
    for (unsigned b = (j >> 1); b < a.items; b += 2)
    {
        double a0[16], a1[16];
#pragma unroll
        for (int e = 0; e < 16; e++)
            a0[e] = a1[e] = 0.0;
        P1 cur, nxt;
        if (PF)
            nxt.load(a.digits + ((size_t)(b * kDigits) << 16), rank, tid);
        for (unsigned J = 0; J < kDigits; J++, step++)
        {
            uint64_t *slot = slot0 + (a.slots == 2 ? (step & 1) * kN : 0);
            if (PF)
            {
                cur = nxt;
                if (J + 1 < kDigits)
                    nxt.load(a.digits + ((size_t)(b * kDigits + J + 1) << 16), rank, tid);
            }
            else
                cur.load(a.digits + ((size_t)(b * kDigits + J) << 16), rank, tid);
            chain(cur.x, a.work1, a.one, a.zero);
            p1_store<ST>(cur.x, slot, rank, tid, I);
            double k0[16], k1[16];
            key_load(k0, k1, a.key, J, I, rank, tid); // travels while the team meets
            team_arrive_wait<V>(ctr, 16 * ++phase, a.status);
            uint64_t n[16];
            p2_load<1>(n, slot, rank, tid);
            double x[16];
#pragma unroll
            for (int e = 0; e < 16; e++)
                x[e] = __builtin_bit_cast(double, n[e]);
            chain(x, a.work2, a.one, a.zero);
            mac(a0, a1, x, k0, k1, a.one, a.zero);
            if (a.slots == 1)
                team_arrive_wait<V>(ctr, 16 * ++phase, a.status); // the slot may be overwritten only when every member has read it
        }
        sums_store(a0, a1, a.out, b, I, rank, tid);
    }
}

__global__ void k_fill(uint64_t *p, size_t words, uint64_t mask, uint64_t salt)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x)
        p[i] = (((i + salt) * 0x9E3779B97F4A7C15ull) >> 20) & mask;
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
    const unsigned items = argc > 1 ? (unsigned)atoi(argv[1]) : 64;
    const int reps = 3;
    const size_t lds_bytes = 36 * 1024; // as the real kernels
    uint64_t *digits, *key, *mid, *out, *ref;
    unsigned long long *d_bad;
    const size_t dw = (size_t)items * kDigits * kN, kw = (size_t)kDigits * kTargets * 2 * kN, mw = (size_t)items * kTargets * kDigits * kN,
                 ow = (size_t)items * kTargets * 2 * kN;
    CK(hipMalloc(&digits, dw * 8));
    CK(hipMalloc(&key, kw * 8));
    CK(hipMalloc(&mid, mw * 8));
    CK(hipMalloc(&out, ow * 8));
    CK(hipMalloc(&ref, ow * 8));
    CK(hipMalloc(&d_bad, 8));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, digits, dw, (uint64_t)0xFFFFF, (uint64_t)1);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, key, kw, (uint64_t)0x3FF, (uint64_t)77);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("%u items x %u targets x %u digits of 2^16 words: %.1f GB of half-transformed digits per direction, key %.0f MB, digits %.0f MB\n", items, kTargets, kDigits,
           mw * 8 / 1e9, kw * 8 / 1e6, dw * 8 / 1e6);
    const int works[][2] = { { 0, 0 }, { 40, 40 }, { 56, 64 } };
    for (auto &wk : works)
    {
        printf("--- work = %d / %d dependent FMAs per word in pass 1 / pass 2 (+ the key products)\n", wk[0], wk[1]);
        Args a{ digits, key, mid, ref, items, wk[0], wk[1], 1.0, 0.0, nullptr, nullptr, 1 };
        float best1 = 1e9f, best2 = 1e9f;
        const dim3 g(items * 16 * kTargets), blk(kT);
        auto timed = [&](auto launch) {
            float best = 1e9f;
            for (int r = 0; r < reps + 1; r++)
            {
                float m;
                CK(hipEventRecord(e0));
                launch();
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&m, e0, e1));
                if (r && m < best)
                    best = m;
            }
            return best;
        };
        // variants that say what the two kernels wait for: no digit loads / no key loads / the b-contiguous intermediate
        Args an = a;
        an.out = out;
        const float k1_nodig = timed([&] { hipLaunchKernelGGL((k1<0, false>), g, blk, lds_bytes, 0, an); });
        const float k1_l1 = timed([&] { hipLaunchKernelGGL((k1<1, true>), g, blk, lds_bytes, 0, an); });
        const float k2_l1 = timed([&] { hipLaunchKernelGGL((k2<1, true>), g, blk, lds_bytes, 0, an); });
        const float k2_l1_nokey = timed([&] { hipLaunchKernelGGL((k2<1, false>), g, blk, lds_bytes, 0, an); });
        best1 = timed([&] { hipLaunchKernelGGL((k1<0, true>), g, blk, lds_bytes, 0, a); });
        const float k2_nokey = timed([&] { hipLaunchKernelGGL((k2<0, false>), g, blk, lds_bytes, 0, an); });
        best2 = timed([&] { hipLaunchKernelGGL((k2<0, true>), g, blk, lds_bytes, 0, a); });
        CK(hipMemset(out, 0, ow * 8));
        printf("A two kernels                                : %7.3f + %7.3f = %7.3f ms   (%.1f us per item)\n", best1, best2, best1 + best2,
               (best1 + best2) * 1e3 / items);
        printf("  pass 1 without its digit loads %7.3f ms; pass 2 without its key loads %7.3f ms\n", k1_nodig, k2_nokey);
        printf("  intermediate laid out [I][J][tile][b] (an XCD's 64 concurrent workgroups read one 2 MiB run per digit): pass 1 %7.3f ms, pass 2 %7.3f ms (without key loads %7.3f)\n",
               k1_l1, k2_l1, k2_l1_nokey);
        for (unsigned slots : { 2u, 1u })
            for (int V : { 2, 1 })
                for (int pf : { 1, 0 })
                {
                    const unsigned grid = 512, nteam = 32;
                    unsigned *counters, *status;
                    CK(hipMalloc(&counters, nteam * 64));
                    CK(hipMalloc(&status, (2 + grid) * 4));
                    Args c{ digits, key, mid, out, items, wk[0], wk[1], 1.0, 0.0, counters, status, slots };
                    float best = 1e9f;
                    unsigned lost = 0;
                    for (int r = 0; r < reps + 1 && !lost; r++)
                    {
                        CK(hipMemset(counters, 0, nteam * 64));
                        CK(hipMemset(status, 0, (2 + grid) * 4));
                        CK(hipEventRecord(e0));
                        if (V == 1 && pf)
                            hipLaunchKernelGGL((k_team<1, true>), dim3(grid), dim3(kT), lds_bytes, 0, c);
                        else if (V == 1)
                            hipLaunchKernelGGL((k_team<1, false>), dim3(grid), dim3(kT), lds_bytes, 0, c);
                        else if (pf)
                            hipLaunchKernelGGL((k_team<2, true>), dim3(grid), dim3(kT), lds_bytes, 0, c);
                        else
                            hipLaunchKernelGGL((k_team<2, false>), dim3(grid), dim3(kT), lds_bytes, 0, c);
                        CK(hipEventRecord(e1));
                        CK(hipEventSynchronize(e1));
                        float ms;
                        CK(hipEventElapsedTime(&ms, e0, e1));
                        if (r && ms < best)
                            best = ms;
                        unsigned st;
                        CK(hipMemcpy(&st, status, 4, hipMemcpyDeviceToHost));
                        lost += st;
                    }
                    std::vector<unsigned> h(2 + grid);
                    CK(hipMemcpy(h.data(), status, (2 + grid) * 4, hipMemcpyDeviceToHost));
                    unsigned mixed = 0;
                    for (unsigned t = 0; t < nteam; t++)
                    {
                        const unsigned xcd = t & 7, j = t >> 3;
                        bool mix = false;
                        for (unsigned r = 0; r < 16; r++)
                            mix |= h[2 + ((j * 16 + r) * 8 + xcd)] != h[2 + (j * 16 * 8 + xcd)];
                        mixed += mix;
                    }
                    CK(hipMemset(d_bad, 0, 8));
                    hipLaunchKernelGGL(k_diff, dim3(4096), dim3(256), 0, 0, ref, out, ow, d_bad);
                    unsigned long long bad;
                    CK(hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost));
                    CK(hipMemset(out, 0, ow * 8));
                    printf("C teams, %u slot(s), %s stores, %s: %7.3f ms   (%.1f us per item)  x%.2f of A   bad=%llu lost_barriers=%u mixed_xcd_teams=%u/%u\n", slots,
                           V == 1 ? "sc1  " : "plain", pf ? "next digit in flight" : "no prefetch         ", best, best * 1e3 / items, (best1 + best2) / best, bad,
                           lost, mixed, nteam);
                    CK(hipFree(counters));
                    CK(hipFree(status));
                }
    }
    return 0;
}
