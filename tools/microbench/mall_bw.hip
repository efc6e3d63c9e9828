// What does the 256 MiB Infinity Cache give a re-used footprint on this part?  
//
// Part A — footprint sweep.  Read-only, write-only, in-place update (read + write of the SAME lines) and copy (F in -> F out) over a
// footprint F re-used REUSE times, F = 8 ... 1024 MiB, with every cache-policy flavour the ISA offers a 16-byte access
// (aux of the raw buffer instructions: 0 plain, 2 nt, 16 sc1, 17 sc0 sc1), in two shapes: "flat" (one 4 KiB chunk per workgroup
// iteration, grid-stride) and "tile" (a 32 KiB tile per workgroup iteration, 8 accesses of 16 B in flight per lane — the shape of a
// transform pass).  One table of TB/s.
//
// Part B — the ring.  What a single-launch two-pass transform whose intermediate lives in a re-used ring could get at best, with no
// synchronisation at all (the words are not checked: this is a bandwidth bound, not a hand-over): ONE persistent launch, every
// workgroup alternates   tile i: big input (4 GiB, HBM) -> ring slot i mod S     and     ring slot (i - lag) mod S -> big output,
// ring sizes 16 ... 512 MiB and "no ring" (the intermediate as large as the input = today's two-pass traffic, as one launch and as
// two launches), per flavour of the ring's stores and loads.  `work` FMAs per word stand in for the butterflies.
// GB/s in part B are ALGORITHMIC: (input bytes + output bytes) / time, the figure roofline.achieved uses.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int kT = 256;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t window(const void *p)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, 0x7fffffff, 0x00020000);
}
template <int AUX>
__device__ __forceinline__ u32x4 ld16(__amdgpu_buffer_rsrc_t r, unsigned lane_bytes, unsigned uni_bytes)
{
    return __builtin_amdgcn_raw_buffer_load_b128(r, (int)lane_bytes, (int)uni_bytes, AUX);
}
template <int AUX>
__device__ __forceinline__ void st16(__amdgpu_buffer_rsrc_t r, unsigned lane_bytes, unsigned uni_bytes, u32x4 v)
{
    __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)lane_bytes, (int)uni_bytes, AUX);
}

// ---------------------------------------------------------------- part A
// MODE 0 read, 1 write, 2 in-place update, 3 copy.  U = accesses in flight per lane (1: flat 4 KiB chunks; 8: 32 KiB tiles).
template <int MODE, int U, int LAUX, int SAUX>
__global__ void __launch_bounds__(kT) k_sweep(const uint8_t *src, uint8_t *dst, size_t bytes, u32x4 *sink)
{
    const size_t chunk = (size_t)kT * 16 * U, chunks = bytes / chunk;
    u32x4 acc = { 0, 0, 0, 0 };
    for (size_t t = blockIdx.x; t < chunks; t += gridDim.x)
    {
        const __amdgpu_buffer_rsrc_t rs = window(src + t * chunk), rd = window(dst + t * chunk);
        u32x4 v[U];
#pragma unroll
        for (int e = 0; e < U; e++)
        {
            if constexpr (MODE != 1)
                v[e] = ld16<LAUX>(rs, threadIdx.x * 16, e * kT * 16);
            else
                v[e] = u32x4{ (unsigned)t, (unsigned)e, threadIdx.x, 1 };
        }
#pragma unroll
        for (int e = 0; e < U; e++)
        {
            if constexpr (MODE == 0)
                acc ^= v[e];
            else
                st16<SAUX>(MODE == 2 ? rs : rd, threadIdx.x * 16, e * kT * 16, MODE == 2 ? v[e] + 1u : v[e]);
        }
    }
    if (MODE == 0 && acc.x == 0x12345678u && acc.y == 0x9abcdef0u)
        sink[0] = acc;
}

static float time_launches(int reuse, void (*launch)(void *), void *ctx)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++)
        launch(ctx);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reuse; i++)
        launch(ctx);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
    return ms / reuse;
}

struct SweepCtx
{
    const uint8_t *src;
    uint8_t *dst;
    size_t bytes;
    u32x4 *sink;
    unsigned grid;
};
template <int MODE, int U, int LAUX, int SAUX>
static void launch_sweep(void *p)
{
    SweepCtx *c = (SweepCtx *)p;
    hipLaunchKernelGGL((k_sweep<MODE, U, LAUX, SAUX>), dim3(c->grid), dim3(kT), 0, 0, c->src, c->dst, c->bytes, c->sink);
}

template <int MODE, int U>
static void sweep_row(const char *name, SweepCtx &c, int reuse)
{
    const size_t chunks = c.bytes / ((size_t)kT * 16 * U);
    // persistent grid (2048 workgroups) unless the footprint has fewer chunks
    c.grid = (unsigned)(chunks < 2048 ? chunks : 2048);
    const double moved = (MODE >= 2 ? 2.0 : 1.0) * (double)c.bytes;
    float ms[4];
    ms[0] = time_launches(reuse, launch_sweep<MODE, U, 0, 0>, &c);
    ms[1] = time_launches(reuse, launch_sweep<MODE, U, 2, 2>, &c);
    ms[2] = time_launches(reuse, launch_sweep<MODE, U, 16, 16>, &c);
    ms[3] = time_launches(reuse, launch_sweep<MODE, U, 17, 17>, &c);
    printf("  %-30s", name);
    for (int i = 0; i < 4; i++)
        printf("  %7.2f", moved / (ms[i] * 1e-3) / 1e12);
    printf("   TB/s  (plain, nt, sc1, sc0sc1)\n");
}

// ---------------------------------------------------------------- part B
__device__ __forceinline__ void work_on(u32x4 (&x)[8], int work, double a, double b)
{
    if (work == 0)
        return;
    double d[16];
#pragma unroll
    for (int e = 0; e < 8; e++)
    {
        d[2 * e] = __builtin_bit_cast(double, ((uint64_t)(x[e].y & 0xfffffu) << 32 | x[e].x) | 0x4330000000000000ull);
        d[2 * e + 1] = __builtin_bit_cast(double, ((uint64_t)(x[e].w & 0xfffffu) << 32 | x[e].z) | 0x4330000000000000ull);
    }
    for (int i = 0; i < work; i++)
    {
#pragma unroll
        for (int e = 0; e < 16; e++)
            d[e] = __builtin_fma(d[e], a, b);
    }
#pragma unroll
    for (int e = 0; e < 8; e++)
    {
        const uint64_t lo = __builtin_bit_cast(uint64_t, d[2 * e]), hi = __builtin_bit_cast(uint64_t, d[2 * e + 1]);
        x[e] = u32x4{ (unsigned)lo, (unsigned)(lo >> 32) & 0xfffffu, (unsigned)hi, (unsigned)(hi >> 32) & 0xfffffu };
    }
}

struct RingArgs
{
    const uint8_t *in;
    uint8_t *out;
    uint8_t *ring;
    size_t tiles;      // 32 KiB tiles of input (= of output)
    size_t ring_tiles; // S
    size_t lag;        // the read side runs `lag` tiles behind the write side
    int work;
    double a, b;
};
constexpr size_t kTile = (size_t)kT * 16 * 8; // 32 KiB

// PHASES: 3 = both sides in one launch (the ring form); 1 = input -> mid only; 2 = mid -> output only (the two-launch form).
template <int PHASES, int SAUX, int LAUX>
__global__ void __launch_bounds__(kT) k_ring(RingArgs a)
{
    for (size_t i = blockIdx.x; i < a.tiles + (PHASES == 3 ? a.lag : 0); i += gridDim.x)
    {
        if ((PHASES & 1) && i < a.tiles)
        {
            const __amdgpu_buffer_rsrc_t rs = window(a.in + i * kTile), rd = window(a.ring + (i % a.ring_tiles) * kTile);
            u32x4 v[8];
#pragma unroll
            for (int e = 0; e < 8; e++)
                v[e] = ld16<2>(rs, threadIdx.x * 16, e * kT * 16); // the big input streams non-temporal, as the real passes do
            work_on(v, a.work, a.a, a.b);
#pragma unroll
            for (int e = 0; e < 8; e++)
                st16<SAUX>(rd, threadIdx.x * 16, e * kT * 16, v[e]);
        }
        if ((PHASES & 2) && (PHASES != 3 || i >= a.lag))
        {
            const size_t j = PHASES == 3 ? i - a.lag : i;
            // a transposing choice of slot so that the reader is not the workgroup that wrote it (as pass 2 reads what 16 pass-1 tiles wrote)
            const size_t slot = (j ^ 1) < a.tiles ? (j ^ 1) : j;
            const __amdgpu_buffer_rsrc_t rs = window(a.ring + (slot % a.ring_tiles) * kTile), rd = window(a.out + j * kTile);
            u32x4 v[8];
#pragma unroll
            for (int e = 0; e < 8; e++)
                v[e] = ld16<LAUX>(rs, threadIdx.x * 16, e * kT * 16);
            work_on(v, a.work, a.a, a.b);
#pragma unroll
            for (int e = 0; e < 8; e++)
                st16<2>(rd, threadIdx.x * 16, e * kT * 16, v[e]);
        }
    }
}

struct RingCtx
{
    RingArgs a;
    unsigned grid;
};
template <int PHASES, int SAUX, int LAUX>
static void launch_ring(void *p)
{
    RingCtx *c = (RingCtx *)p;
    if (PHASES == 3)
        hipLaunchKernelGGL((k_ring<3, SAUX, LAUX>), dim3(c->grid), dim3(kT), 0, 0, c->a);
    else
    {
        hipLaunchKernelGGL((k_ring<1, SAUX, LAUX>), dim3(c->grid), dim3(kT), 0, 0, c->a);
        hipLaunchKernelGGL((k_ring<2, SAUX, LAUX>), dim3(c->grid), dim3(kT), 0, 0, c->a);
    }
}
template <int SAUX, int LAUX>
static void ring_cell(RingCtx &c, bool two_launches)
{
    const float ms = time_launches(6, two_launches ? launch_ring<0, SAUX, LAUX> : launch_ring<3, SAUX, LAUX>, &c);
    printf("  %7.0f", 2.0 * (double)c.a.tiles * kTile / (ms * 1e-3) / 1e9);
}
static void ring_row(const char *name, RingCtx &c, bool two_launches)
{
    printf("  %-40s", name);
    ring_cell<0, 0>(c, two_launches);   // plain / plain
    ring_cell<2, 2>(c, two_launches);   // nt / nt  (today's intermediate)
    ring_cell<16, 16>(c, two_launches); // sc1 / sc1 (the valid write-through hand-over)
    ring_cell<17, 17>(c, two_launches); // sc0 sc1 both sides
    ring_cell<16, 0>(c, two_launches);  // sc1 stores, plain loads (valid behind an agent acquire)
    ring_cell<0, 16>(c, two_launches);  // plain stores (valid behind an agent release), sc1 loads
    printf("   GB/s algorithmic\n");
}

int main(int argc, char **argv)
{
    const char *part = argc > 1 ? argv[1] : "ab";
    const size_t big = size_t(4) << 30;
    uint8_t *a, *b, *ring;
    u32x4 *sink;
    CK(hipMalloc(&a, big));
    CK(hipMalloc(&b, big));
    CK(hipMalloc(&ring, big));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(a, 1, big));
    CK(hipMemset(b, 2, big));
    CK(hipMemset(ring, 3, big));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, %d CUs, L2 %d MiB reported\n", prop.name, prop.multiProcessorCount, prop.l2CacheSize >> 20);

    if (strchr(part, 'a'))
    {
        printf("== part A: footprint F re-used; 2048-workgroup persistent grid; 16 B per lane; TB/s of bytes moved (update / copy count read + write)\n");
        for (size_t mib : { 8, 16, 32, 64, 96, 128, 192, 256, 512, 1024 })
        {
            SweepCtx c{ a, b, mib << 20, sink, 0 };
            const int reuse = mib <= 64 ? 200 : mib <= 256 ? 100 : 50;
            printf("F = %4zu MiB, re-used %d times\n", mib, reuse);
            sweep_row<0, 1>("read   flat (4 KiB / WG iter)", c, reuse);
            sweep_row<0, 8>("read   tile (32 KiB / WG iter)", c, reuse);
            sweep_row<1, 1>("write  flat", c, reuse);
            sweep_row<1, 8>("write  tile", c, reuse);
            sweep_row<2, 1>("update flat (same lines r+w)", c, reuse);
            sweep_row<2, 8>("update tile (same lines r+w)", c, reuse);
            sweep_row<3, 1>("copy   flat (F in -> F out)", c, reuse);
            sweep_row<3, 8>("copy   tile (F in -> F out)", c, reuse);
        }
    }
    if (strchr(part, 'b'))
    {
        for (int work : { 0, 24 })
        {
            for (unsigned grid : { 1024u, 2048u })
            {
                printf("== part B: 4 GiB in -> intermediate -> 4 GiB out, 32 KiB tiles, %u persistent workgroups, work = %d FMAs per word per side\n", grid, work);
                printf("  %-40s  %7s  %7s  %7s  %7s  %7s  %7s\n", "intermediate (store / load flavour:)", "pl/pl", "nt/nt", "sc1/sc1", "s01/s01", "sc1/pl", "pl/sc1");
                RingCtx c;
                c.grid = grid;
                c.a = RingArgs{ a, b, ring, big / kTile, big / kTile, 0, work, 1.0000001, 0.5 };
                ring_row("two launches, 4 GiB intermediate", c, true);
                c.a.lag = 4 * grid;
                ring_row("one launch,   4 GiB intermediate", c, false);
                for (size_t mib : { 512, 256, 192, 128, 96, 64, 32, 16 })
                {
                    c.a.ring_tiles = (mib << 20) / kTile;
                    for (size_t lagx : { 2, 4 })
                    {
                        c.a.lag = lagx * grid;
                        if (c.a.lag >= c.a.ring_tiles)
                            continue;
                        char name[96];
                        snprintf(name, sizeof name, "one launch, ring %4zu MiB, lag %zu x grid", mib, lagx);
                        ring_row(name, c, false);
                    }
                }
            }
        }
    }
    return 0;
}
