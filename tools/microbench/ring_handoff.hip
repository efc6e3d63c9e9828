// The skeleton of a ONE-launch two-pass 2^16-point transform whose intermediate is a re-used ring.
//
// tools/microbench/mall_bw.hip part B measured the bound with no synchronisation: 3.5 TB/s algorithmic against 2.6 for two launches,
// when the ring is written and read with temporal (plain or sc1) accesses.  This file adds what a real kernel needs - the hand-over
// protocol, the passes' real access shapes, every word checked - and measures what is left of that bound.
//
// Structure: S "teams".  Team s transforms u = s, s + S, s + 2S, ... (iteration k <-> u = s + k S) and owns R ring slots of 2^16 words
// (iteration k uses slot k mod R).  A team is 16 pass-1 workgroups (column tile cg) + 16 pass-2 workgroups (row tile hg), all resident
// at once (the grid never exceeds what the chip holds, so nothing depends on dispatch order).  Progress words, zeroed before the launch:
//   prog1[team][cg]        = iterations whose pass-1 tile cg is in memory   (pass 2 of iteration k waits for all sixteen >= k + 1)
//   prog2[team][hg][wave]  = iterations whose pass-2 loads have landed      (pass 1 of iteration k waits for all 64 >= k - R + 1)
// Hand-over (MI355X guide, "valid forms"): the ring is written with write-through sc1 stores, drained (s_waitcnt vmcnt(0)) by every
// wave, workgroup barrier, ONE relaxed agent-scope store of the progress word; the reader polls with relaxed agent-scope loads, then
// reads the ring with sc1 loads (L1 bypass; a slot is re-used, L1 lines of its previous occupant would be stale).  The publish of
// iteration k is DEFERRED to iteration k + 1 (the drain is free by then), the last one is published after the loop.
// Variants: W = 8 (the real kernels' 8-byte accesses, tile order) or 16 (16-byte accesses: rows 2j, 2j+1 of a column adjacent in the
// intermediate, pass 2 hands the odd row over with v_permlane16_swap); FL = 0 sc1/sc1, 1 plain stores + agent release fence by one
// lane / agent acquire + plain loads.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int kT = 256;
constexpr size_t kN = 65536;
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t window(const void *p)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, 0x7fffffff, 0x00020000);
}
template <int AUX>
__device__ __forceinline__ uint64_t ld8(__amdgpu_buffer_rsrc_t r, unsigned lane_bytes, unsigned uni_bytes)
{
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)lane_bytes, (int)uni_bytes, AUX);
    return (uint64_t)v.x | ((uint64_t)v.y << 32);
}
template <int AUX>
__device__ __forceinline__ void st8(__amdgpu_buffer_rsrc_t r, unsigned lane_bytes, unsigned uni_bytes, uint64_t v)
{
    __builtin_amdgcn_raw_buffer_store_b64(u32x2{ (uint32_t)v, (uint32_t)(v >> 32) }, r, (int)lane_bytes, (int)uni_bytes, AUX);
}
template <int AUX>
__device__ __forceinline__ void ld16(__amdgpu_buffer_rsrc_t r, unsigned lane_bytes, unsigned uni_bytes, uint64_t &a, uint64_t &b)
{
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)lane_bytes, (int)uni_bytes, AUX);
    a = (uint64_t)v.x | ((uint64_t)v.y << 32);
    b = (uint64_t)v.z | ((uint64_t)v.w << 32);
}
template <int AUX>
__device__ __forceinline__ void st16(__amdgpu_buffer_rsrc_t r, unsigned lane_bytes, unsigned uni_bytes, uint64_t a, uint64_t b)
{
    __builtin_amdgcn_raw_buffer_store_b128(u32x4{ (uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32) }, r, (int)lane_bytes, (int)uni_bytes, AUX);
}

__device__ __forceinline__ void work_on(uint64_t (&x)[16], int work, double a, double b)
{
    if (work)
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
            x[e] = ((__builtin_bit_cast(uint64_t, d[e]) & 0x000fffffffffffffull) | (x[e] & 0xfff0000000000000ull));
    }
#pragma unroll
    for (int e = 0; e < 16; e++)
        x[e] += 1;
}

struct Args
{
    const uint64_t *in;
    uint64_t *out;
    uint64_t *ring;        // [S][R][kN]
    unsigned *prog1;       // [S][16]
    unsigned *prog2;       // [S][64]
    unsigned *status;      // [0] lost dependencies, [1] stall polls pass 1, [2] stall polls pass 2
    const uint32_t *roles; // per block: role << 31 | team << 4 | tile
    unsigned transforms, S, R;
    int sleep; // 0: s_sleep 2 between polls, 1: s_sleep 32, 2: s_sleep 127
    int work;
    double a, b;
};

// every wave polls for itself (lane i word i) until all `n` progress words are >= target: no workgroup barrier on the waiting side
__device__ __forceinline__ bool wave_wait(const unsigned *words, unsigned n, unsigned target, unsigned *status, unsigned which, int sleep = 0)
{
    const unsigned lane = threadIdx.x & 63;
    unsigned spins = 0;
    bool ok = true;
    for (;;)
    {
        const unsigned v = lane < n ? __hip_atomic_load(words + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : target;
        if (__all(v >= target))
            break;
        if (sleep == 0)
            __builtin_amdgcn_s_sleep(2);
        else if ((sleep & 3) == 1)
            __builtin_amdgcn_s_sleep(32);
        else
            __builtin_amdgcn_s_sleep(127);
        if (++spins > (1u << 18) || __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        {
            ok = false;
            break;
        }
    }
    if (lane == 0)
    {
        if (!ok)
            __hip_atomic_fetch_add(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (spins)
            __hip_atomic_fetch_add(status + which, spins, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return ok;
}

template <int W, int FL, bool SYNC = true>
__global__ void __launch_bounds__(kT) k_ring(Args a)
{
    __shared__ uint64_t lds[16 * 17 * 16];
    const unsigned tid = threadIdx.x;
    const uint32_t rl = a.roles[blockIdx.x];
    const unsigned role = rl >> 31, team = (rl >> 4) & 0x7ffffff, tile = rl & 15;
    const unsigned iters = (a.transforms - team + a.S - 1) / a.S;
    uint64_t *ring = a.ring + (size_t)team * a.R * kN;
    constexpr int SAUX = FL == 0 ? 16 : FL == 3 ? 2 : 0, LAUX = FL == 0 ? 16 : FL == 3 ? 2 : 0; // FL 2: plain, no fences; 3: nt, no fences (timing only)
    if (role == 0)
    {
        // ---------------- pass 1: column tile cg of every transform of the team
        const unsigned cg = tile, c = tid & 15, hi = tid >> 4;
        unsigned *mine = a.prog1 + team * 16 + cg;
        const unsigned *theirs = a.prog2 + team * 64;
        uint64_t nxt[16];
        auto fetch = [&](unsigned k) {
            const uint64_t *in = a.in + (size_t)(team + k * a.S) * kN;
#pragma unroll
            for (int e = 0; e < 16; e++)
                nxt[e] = __builtin_nontemporal_load(in + (size_t)(e * 16 + hi) * 256 + cg * 16 + c);
        };
        fetch(0);
        for (unsigned k = 0; k < iters; k++)
        {
            uint64_t x[16];
#pragma unroll
            for (int e = 0; e < 16; e++)
                x[e] = nxt[e];
            if (k + 1 < iters)
                fetch(k + 1);
            work_on(x, a.work, a.a, a.b);
            // the workgroup-wide exchange of the real pass (two barriers per tile); the deferred publish of iteration k - 1 rides on it
            // this wave's ring stores of iteration k - 1 are in memory: they were issued before the 16 prefetch loads, returns are in order
            if (k + 1 < iters)
                asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int e = 0; e < 16; e++)
                lds[(e * 16 + hi) * 17 + c] = x[e];
            __syncthreads();
            if (SYNC && k > 0 && tid == 0)
            {
                if (FL == 1)
                {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __hip_atomic_store(mine, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int e = 0; e < 16; e++)
                x[e] = lds[(e * 16 + hi) * 17 + c];
            // the slot's previous occupant (iteration k - R) must have been read by all 64 waves of pass 2
            __syncthreads();
            if (SYNC && k >= a.R && !wave_wait(theirs, 64, k - a.R + 1, a.status, 1, a.sleep))
                return;
            uint64_t *slot = ring + (size_t)(k % a.R) * kN;
            const __amdgpu_buffer_rsrc_t rs = window(slot + (size_t)(hi * 16 + cg) * 256);
            if constexpr (W == 8)
            {
#pragma unroll
                for (int e = 0; e < 16; e++)
                    st8<SAUX>(rs, c * 8, e * 128, x[e]); // tile order: block (hg = hi, cg), word e * 16 + c
            }
            else
            {
#pragma unroll
                for (int j = 0; j < 8; j++)
                    st16<SAUX>(rs, c * 16, j * 256, x[2 * j], x[2 * j + 1]); // word (e >> 1) * 32 + c * 2 + (e & 1)
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (SYNC && tid == 0)
        {
            if (FL == 1)
            {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __hip_atomic_store(mine, iters, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    else
    {
        // ---------------- pass 2: row tile hg; thread (u = tid >> 4, v = tid & 15) ends up with row u, column v of the sixteen blocks e
        const unsigned hg = tile, u = tid >> 4, v = tid & 15;
        unsigned *mine = a.prog2 + team * 64 + hg * 4 + (tid >> 6);
        const unsigned *theirs = a.prog1 + team * 16;
        uint64_t nxt[16];
        auto fetch = [&](unsigned k) {
            const uint64_t *slot = ring + (size_t)(k % a.R) * kN + (size_t)hg * 4096;
            const __amdgpu_buffer_rsrc_t rs = window(slot);
            if constexpr (W == 8)
            {
#pragma unroll
                for (int e = 0; e < 16; e++)
                    nxt[e] = ld8<LAUX>(rs, tid * 8, e * 2048);
            }
            else
            {
                // rows (u & ~1, u | 1) of column v of blocks e = (u & 1) * 8 + j
#pragma unroll
                for (int j = 0; j < 8; j++)
                    ld16<LAUX>(rs, (u >> 1) * 256 + v * 16, ((u & 1) * 8 + j) * 2048, nxt[2 * j], nxt[2 * j + 1]);
            }
        };
        auto acquire = [&](unsigned k) -> bool {
            if (!SYNC)
                return true;
            if (!wave_wait(theirs, 16, k + 1, a.status, 2, a.sleep))
                return false;
            if (FL == 1)
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); // per wave (variant for comparison only)
            return true;
        };
        if (!acquire(0))
            return;
        fetch(0);
        for (unsigned k = 0; k < iters; k++)
        {
            uint64_t x[16];
            if constexpr (W == 8)
            {
#pragma unroll
                for (int e = 0; e < 16; e++)
                    x[e] = nxt[e];
            }
            else
            {
                // even rows keep their .x (row u of blocks 0..7) and take the odd partner's .x (row u of blocks 8..15); odd rows
                // take the even partner's .y (row u of blocks 0..7) and keep their .y: swap Y[even rows] <-> X[odd rows]
#pragma unroll
                for (int j = 0; j < 8; j++)
                {
                    uint32_t xl = (uint32_t)nxt[2 * j], xh = (uint32_t)(nxt[2 * j] >> 32), yl = (uint32_t)nxt[2 * j + 1], yh = (uint32_t)(nxt[2 * j + 1] >> 32);
                    const auto rl2 = __builtin_amdgcn_permlane16_swap(xl, yl, false, false); // odd rows of the first <-> even rows of the second
                    const auto rh2 = __builtin_amdgcn_permlane16_swap(xh, yh, false, false);
                    const uint64_t X = (uint64_t)rl2[0] | ((uint64_t)rh2[0] << 32), Y = (uint64_t)rl2[1] | ((uint64_t)rh2[1] << 32);
                    // even rows: X = own x (block j), Y = partner's x (block 8 + j).  odd rows: X = partner's y (block j), Y = own y (block 8 + j)
                    x[j] = X;
                    x[8 + j] = Y;
                }
            }
            // this wave's loads of iteration k have landed (the last one issued has: returns are in order; the empty asm is a use the
            // compiler must wait for): the slot may be rewritten as far as this wave is concerned
            asm volatile("" ::"v"(x[15]), "v"(x[7]) : "memory");
            if (SYNC && (tid & 63) == 0)
                __hip_atomic_store(mine, k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (k + 1 < iters)
            {
                if (!acquire(k + 1))
                    return;
                fetch(k + 1);
            }
            work_on(x, a.work, a.a, a.b);
            uint64_t *o = a.out + (size_t)(team + k * a.S) * kN + (size_t)hg * 4096 + tid;
#pragma unroll
            for (int e = 0; e < 16; e++)
                __builtin_nontemporal_store(x[e], o + e * 256);
        }
    }
}

// ------------------------------------------------------------------ the fused form: every workgroup alternates pass 1 and pass 2
// Team = the 16 workgroups (tile t) that share the transforms u = team + k S.  Iteration k of a workgroup: pass-1 tile t of transform k
// -> ring slot k mod R, then pass-2 tile t of transform k - L out of slot (k - L) mod R.  Every workgroup of a team does the same work
// at the same pace, so the words a wait is for were published about L - 1 iterations earlier: the polls are loaded one iteration ahead
// (latency hidden) and succeed the first time; nobody spins in the steady state.  Progress words PER WAVE (no barrier on the publishing
// side): prog1[team][tile][wave] = pass-1 iterations whose ring stores are in memory, prog2[...] = pass-2 iterations whose loads landed.
// A wave publishes both where its ring loads of this iteration have landed: returns are in order, so everything it issued before them
// (its ring stores of the previous iteration) is complete.
template <int W, bool SYNC, int WAVES>
__global__ void __launch_bounds__(kT, WAVES) k_fused(Args a, unsigned L)
{
    __shared__ uint64_t lds[16 * 17 * 16];
    const unsigned tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned team = blockIdx.x >> 4, tile = blockIdx.x & 15;
    const unsigned iters = (a.transforms - team + a.S - 1) / a.S;
    uint64_t *ring = a.ring + (size_t)team * a.R * kN;
    const unsigned *prog1 = a.prog1 + team * 64, *prog2 = a.prog2 + team * 64;
    unsigned *my1 = a.prog1 + team * 64 + tile * 4 + wave, *my2 = a.prog2 + team * 64 + tile * 4 + wave;
    const unsigned c = tid & 15, hi = tid >> 4; // pass 1: column c of column tile cg = tile, rows e * 16 + hi
    const unsigned u = tid >> 4, v = tid & 15;  // pass 2: row u, column v of the sixteen blocks of row tile hg = tile
    uint64_t nxa[16], nxb[16];
    auto fetch_a = [&](unsigned k) {
        const uint64_t *in = a.in + (size_t)(team + k * a.S) * kN;
#pragma unroll
        for (int e = 0; e < 16; e++)
            nxa[e] = __builtin_nontemporal_load(in + (size_t)(e * 16 + hi) * 256 + tile * 16 + c);
    };
    auto fetch_b = [&](unsigned j) {
        const __amdgpu_buffer_rsrc_t rs = window(ring + (size_t)(j % a.R) * kN + (size_t)tile * 4096);
        if constexpr (W == 8)
        {
#pragma unroll
            for (int e = 0; e < 16; e++)
                nxb[e] = ld8<16>(rs, tid * 8, e * 2048);
        }
        else
        {
#pragma unroll
            for (int j2 = 0; j2 < 8; j2++)
                ld16<16>(rs, (u >> 1) * 256 + v * 16, ((u & 1) * 8 + j2) * 2048, nxb[2 * j2], nxb[2 * j2 + 1]);
        }
    };
    auto ensure = [&](const unsigned *words, unsigned have, unsigned target, unsigned which) -> bool {
        if (!SYNC || __all(have >= target))
            return true;
        return wave_wait(words, 64, target, a.status, which, a.sleep);
    };
    unsigned have1 = 0, have2 = 0;
    fetch_a(0);
    for (unsigned k = 0; k < iters + L; k++)
    {
        const bool has1 = k < iters, has2 = k >= L;
        const unsigned j = k - L;
        // one prefetch buffer is in flight at a time: the ring tile of pass 2 is requested once pass 1 has taken its input out of
        // its buffer, the next input of pass 1 once pass 2 has taken the ring tile
        uint64_t x[16];
        if (has1)
        {
#pragma unroll
            for (int e = 0; e < 16; e++)
                x[e] = nxa[e];
        }
        if (has2)
        {
            if (!ensure(prog1, have1, j + 1, 2))
                return;
            fetch_b(j);
        }
        if (has1)
        {
            work_on(x, a.work, a.a, a.b);
#pragma unroll
            for (int e = 0; e < 16; e++)
                lds[(e * 16 + hi) * 17 + c] = x[e];
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 16; e++)
                x[e] = lds[(e * 16 + hi) * 17 + c];
            __syncthreads();
            if (k >= a.R && !ensure(prog2, have2, k - a.R + 1, 1))
                return;
            const __amdgpu_buffer_rsrc_t rs = window(ring + (size_t)(k % a.R) * kN + (size_t)(hi * 16 + tile) * 256);
            if constexpr (W == 8)
            {
#pragma unroll
                for (int e = 0; e < 16; e++)
                    st8<16>(rs, c * 8, e * 128, x[e]);
            }
            else
            {
#pragma unroll
                for (int j2 = 0; j2 < 8; j2++)
                    st16<16>(rs, c * 16, j2 * 256, x[2 * j2], x[2 * j2 + 1]);
            }
        }
        if (has2)
        {
            if constexpr (W == 8)
            {
#pragma unroll
                for (int e = 0; e < 16; e++)
                    x[e] = nxb[e];
            }
            else
            {
#pragma unroll
                for (int j2 = 0; j2 < 8; j2++)
                {
                    uint32_t xl = (uint32_t)nxb[2 * j2], xh = (uint32_t)(nxb[2 * j2] >> 32), yl = (uint32_t)nxb[2 * j2 + 1], yh = (uint32_t)(nxb[2 * j2 + 1] >> 32);
                    const auto rl2 = __builtin_amdgcn_permlane16_swap(xl, yl, false, false);
                    const auto rh2 = __builtin_amdgcn_permlane16_swap(xh, yh, false, false);
                    x[j2] = (uint64_t)rl2[0] | ((uint64_t)rh2[0] << 32);
                    x[8 + j2] = (uint64_t)rl2[1] | ((uint64_t)rh2[1] << 32);
                }
            }
            asm volatile("" ::"v"(x[15]), "v"(x[7]) : "memory"); // the ring loads have landed, and with them everything issued before them
        }
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (SYNC && lane == 0 && !(a.sleep & 16))
        {
            // ring stores of pass-1 iterations 0 .. k-1 are complete (those of iteration k were issued after the ring loads)
            __hip_atomic_store(my1, k < iters ? k : iters, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (has2)
                __hip_atomic_store(my2, j + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (k + 1 < iters)
            fetch_a(k + 1);
        if (SYNC && (a.sleep & 32))
            have1 = have2 = 0x7fffffff;
        else if (SYNC)
        {
            // next iteration's polls, one coalesced 256-byte load each, consumed an iteration later
            have1 = __hip_atomic_load(prog1 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            have2 = __hip_atomic_load(prog2 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (has2)
        {
            work_on(x, a.work, a.a, a.b);
            uint64_t *o = a.out + (size_t)(team + j * a.S) * kN + (size_t)tile * 4096 + tid;
#pragma unroll
            for (int e = 0; e < 16; e++)
                __builtin_nontemporal_store(x[e], o + e * 256);
        }
    }
}

// ------------------------------------------------------------------ fused, one publisher and one poller per workgroup
// As k_fused, with the hand-over hung on the two barriers of pass 1's exchange (B1 before the LDS writes, B2 after them):
//   * ONE progress pair per workgroup: prog[team][tile] = { pass-1 iterations complete, pass-2 iterations landed }, published by lane 0
//     after B1 of the NEXT iteration (every wave has by then waited for its next input, which it requested after its ring stores, and
//     has consumed its ring loads: returns are in order).
//   * ONE poller per workgroup: wave 0 loads the team's 16 pairs (128 bytes) after B2 - the load lands during the rest of the iteration -
//     and leaves the verdict for the next iteration in LDS before B1; every wave reads it after B1 and then requests its ring tile.
//   * a team's progress block sits on its own 4 KiB + 256 bytes, so that the polls of different teams spread over the memory channels.
// Per team and iteration: 16 polls of 128 bytes and 16 publishes, instead of 128 polls of 256 bytes and 128 publishes.
constexpr unsigned kProgStride = (4096 + 256) / 4; // words between the progress blocks of consecutive teams
template <int W, bool SYNC, int WAVES>
__global__ void __launch_bounds__(kT, WAVES) k_fused2(Args a, unsigned L)
{
    __shared__ uint64_t lds[16 * 17 * 16];
    __shared__ unsigned s_have[2]; // min over the team of { pass-1 complete, pass-2 landed } as polled by wave 0
    const unsigned tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned team = blockIdx.x >> 4, tile = blockIdx.x & 15;
    const unsigned iters = (a.transforms - team + a.S - 1) / a.S;
    uint64_t *ring = a.ring + (size_t)team * a.R * kN;
    unsigned *prog = a.prog1 + (size_t)team * kProgStride; // [16 tiles][2]
    const unsigned c = tid & 15, hi = tid >> 4;
    const unsigned u = tid >> 4, v = tid & 15;
    uint64_t nxa[16], nxb[16];
    auto fetch_a = [&](unsigned k) {
        const uint64_t *in = a.in + (size_t)(team + k * a.S) * kN;
#pragma unroll
        for (int e = 0; e < 16; e++)
            nxa[e] = __builtin_nontemporal_load(in + (size_t)(e * 16 + hi) * 256 + tile * 16 + c);
    };
    auto fetch_b = [&](unsigned j) {
        const __amdgpu_buffer_rsrc_t rs = window(ring + (size_t)(j % a.R) * kN + (size_t)tile * 4096);
        if constexpr (W == 8)
        {
#pragma unroll
            for (int e = 0; e < 16; e++)
                nxb[e] = ld8<16>(rs, tid * 8, e * 2048);
        }
        else
        {
#pragma unroll
            for (int j2 = 0; j2 < 8; j2++)
                ld16<16>(rs, (u >> 1) * 256 + v * 16, ((u & 1) * 8 + j2) * 2048, nxb[2 * j2], nxb[2 * j2 + 1]);
        }
    };
    // slow path (rare): this wave polls the team's pairs itself until word `which` of all sixteen is >= target
    auto spin = [&](unsigned which, unsigned target) -> bool {
        unsigned spins = 0;
        for (;;)
        {
            const unsigned v2 = lane < 16 ? __hip_atomic_load(prog + lane * 2 + which, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : target;
            if (__all(v2 >= target))
                break;
            __builtin_amdgcn_s_sleep(32);
            if (++spins > (1u << 18) || __hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            {
                if (lane == 0)
                    __hip_atomic_fetch_add(a.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
        if (lane == 0 && spins)
            __hip_atomic_fetch_add(a.status + 1 + which, spins, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return true;
    };
    unsigned polled = 0; // wave 0, lanes 0..31: word `lane` of the team's progress block as of the last poll
    if (tid < 2)
        s_have[tid] = 0;
    fetch_a(0);
    for (unsigned k = 0; k < iters + L; k++)
    {
        const bool has1 = k < iters, has2 = k >= L;
        const unsigned j = k - L;
        uint64_t x[16];
        if (has1)
        {
#pragma unroll
            for (int e = 0; e < 16; e++)
                x[e] = nxa[e];
            work_on(x, a.work, a.a, a.b); // (phase A)
        }
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (SYNC && wave == 0 && k > 0)
        {
            // verdict of the poll requested after B2 of the previous iteration: min over the sixteen workgroups, per word
            unsigned m = lane < 32 ? polled : 0xffffffffu;
#pragma unroll
            for (int sft = 2; sft < 32; sft <<= 1)
                m = min(m, (unsigned)__shfl_xor((int)m, sft));
            if (lane < 2)
                s_have[lane] = m;
        }
        __syncthreads(); // B1
        if (SYNC && tid == 0 && k > 0)
        {
            // pass-1 iterations 0 .. k-1 are complete, pass-2 iterations 0 .. j-1 have landed, for every wave of this workgroup
            __hip_atomic_store(prog + tile * 2, k < iters ? k : iters, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (k > L)
                __hip_atomic_store(prog + tile * 2 + 1, j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (has2)
        {
            if (SYNC && s_have[0] < j + 1 && !spin(0, j + 1))
                return;
            fetch_b(j);
        }
        if (has1)
        {
#pragma unroll
            for (int e = 0; e < 16; e++)
                lds[(e * 16 + hi) * 17 + c] = x[e];
        }
        const unsigned have2 = s_have[1];
        __syncthreads(); // B2
        if (SYNC && wave == 0 && lane < 32)
            polled = __hip_atomic_load(prog + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (has1)
        {
#pragma unroll
            for (int e = 0; e < 16; e++)
                x[e] = lds[(e * 16 + hi) * 17 + c];
            if (SYNC && k >= a.R && have2 < k - a.R + 1 && !spin(1, k - a.R + 1))
                return;
            const __amdgpu_buffer_rsrc_t rs = window(ring + (size_t)(k % a.R) * kN + (size_t)(hi * 16 + tile) * 256);
            if constexpr (W == 8)
            {
#pragma unroll
                for (int e = 0; e < 16; e++)
                    st8<16>(rs, c * 8, e * 128, x[e]);
            }
            else
            {
#pragma unroll
                for (int j2 = 0; j2 < 8; j2++)
                    st16<16>(rs, c * 16, j2 * 256, x[2 * j2], x[2 * j2 + 1]);
            }
        }
        if (has2)
        {
            if constexpr (W == 8)
            {
#pragma unroll
                for (int e = 0; e < 16; e++)
                    x[e] = nxb[e];
            }
            else
            {
#pragma unroll
                for (int j2 = 0; j2 < 8; j2++)
                {
                    uint32_t xl = (uint32_t)nxb[2 * j2], xh = (uint32_t)(nxb[2 * j2] >> 32), yl = (uint32_t)nxb[2 * j2 + 1], yh = (uint32_t)(nxb[2 * j2 + 1] >> 32);
                    const auto rl2 = __builtin_amdgcn_permlane16_swap(xl, yl, false, false);
                    const auto rh2 = __builtin_amdgcn_permlane16_swap(xh, yh, false, false);
                    x[j2] = (uint64_t)rl2[0] | ((uint64_t)rh2[0] << 32);
                    x[8 + j2] = (uint64_t)rl2[1] | ((uint64_t)rh2[1] << 32);
                }
            }
            asm volatile("" ::"v"(x[15]), "v"(x[7]) : "memory"); // the ring loads have landed before the next input is requested
        }
        if (k + 1 < iters)
            fetch_a(k + 1);
        if (has2)
        {
            work_on(x, a.work, a.a, a.b);
            uint64_t *o = a.out + (size_t)(team + j * a.S) * kN + (size_t)tile * 4096 + tid;
#pragma unroll
            for (int e = 0; e < 16; e++)
                __builtin_nontemporal_store(x[e], o + e * 256);
        }
    }
}

// ------------------------------------------------------------------ the two-launch reference with the same shapes
template <int ROLE>
__global__ void __launch_bounds__(kT) k_two(const uint64_t *in, uint64_t *mid, uint64_t *out, unsigned transforms, int work, double a, double b)
{
    __shared__ uint64_t lds[16 * 17 * 16];
    const unsigned tid = threadIdx.x, tile = blockIdx.x & 15;
    for (unsigned t = blockIdx.x >> 4; t < transforms; t += gridDim.x >> 4)
    {
        uint64_t x[16];
        if (ROLE == 0)
        {
            const unsigned c = tid & 15, hi = tid >> 4;
#pragma unroll
            for (int e = 0; e < 16; e++)
                x[e] = __builtin_nontemporal_load(in + (size_t)t * kN + (size_t)(e * 16 + hi) * 256 + tile * 16 + c);
            work_on(x, work, a, b);
#pragma unroll
            for (int e = 0; e < 16; e++)
                lds[(e * 16 + hi) * 17 + c] = x[e];
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 16; e++)
                x[e] = lds[(e * 16 + hi) * 17 + c];
            __syncthreads();
            uint64_t *o = mid + (size_t)t * kN + (size_t)(hi * 16 + tile) * 256 + c;
#pragma unroll
            for (int e = 0; e < 16; e++)
                __builtin_nontemporal_store(x[e], o + e * 16);
        }
        else
        {
            const uint64_t *m = mid + (size_t)t * kN + (size_t)tile * 4096 + tid;
#pragma unroll
            for (int e = 0; e < 16; e++)
                x[e] = __builtin_nontemporal_load(m + e * 256);
            work_on(x, work, a, b);
            uint64_t *o = out + (size_t)t * kN + (size_t)tile * 4096 + tid;
#pragma unroll
            for (int e = 0; e < 16; e++)
                __builtin_nontemporal_store(x[e], o + e * 256);
        }
    }
}

__global__ void k_fill(uint64_t *in, size_t words)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x)
        in[i] = (i * 0x9E3779B97F4A7C15ull) >> 13;
}
// out word (hg, e, u, v) of transform t = in word (row e' * 16 + hi', col cg * 16 + c) + 2 with hi' = hg, cg = e, e' = u, c = v
__global__ void k_check(const uint64_t *in, const uint64_t *out, size_t transforms, unsigned long long *bad)
{
    const size_t words = transforms * kN;
    unsigned long long nb = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x)
    {
        const size_t t = i >> 16, p = i & 65535;
        const unsigned hg = p >> 12, e = (p >> 8) & 15, u = (p >> 4) & 15, v = p & 15;
        const uint64_t want = in[t * kN + (size_t)(u * 16 + hg) * 256 + e * 16 + v] + 2;
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

// block -> (role, team, tile).  mix = 0: teams contiguous (16 pass-1 blocks, then the team's 16 pass-2 blocks); 1: runs of 256 blocks
// alternate roles, so that (with round-robin dispatch over 8 XCDs x 32 CUs) every CU holds both roles
static std::vector<uint32_t> make_roles(unsigned S, int mix)
{
    std::vector<uint32_t> r(S * 32);
    if (mix == 0)
    {
        for (unsigned b = 0; b < S * 32; b++)
            r[b] = ((b >> 4) & 1) << 31 | (b >> 5) << 4 | (b & 15);
    }
    else
    {
        unsigned n[2] = { 0, 0 };
        for (unsigned b = 0; b < S * 32; b++)
        {
            unsigned role = (b / 256) & 1;
            if (n[role] >= S * 16)
                role ^= 1;
            const unsigned idx = n[role]++;
            r[b] = role << 31 | (idx >> 4) << 4 | (idx & 15);
        }
    }
    return r;
}

int main(int argc, char **argv)
{
    const size_t T = argc > 1 ? atoi(argv[1]) : 3840;
    const int reps = 4;
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
    printf("%zu transforms of 2^16 words; algorithmic bytes per run %.1f MB; GB/s = algorithmic (16 N per transform)\n", T, alg / 1e6);
    unsigned *prog1, *prog2, *status;
    uint32_t *d_roles;
    CK(hipMalloc(&prog1, 64 * kProgStride * 4));
    CK(hipMalloc(&prog2, 64 * 64 * 4));
    CK(hipMalloc(&status, 64));
    CK(hipMalloc(&d_roles, 64 * 32 * 4));
    for (int work : { 0, 24 })
    {
        printf("--- work = %d FMAs per word per pass\n", work);
        for (unsigned grid : { 4096u, 8192u })
        {
            float best = 1e9f;
            for (int r = 0; r < reps + 1; r++)
            {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(k_two<0>, dim3(grid), dim3(kT), 0, 0, in, mid, out, (unsigned)T, work, 1.0, 0.0);
                hipLaunchKernelGGL(k_two<1>, dim3(grid), dim3(kT), 0, 0, in, mid, out, (unsigned)T, work, 1.0, 0.0);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (r && ms < best)
                    best = ms;
            }
            unsigned long long bad = verify(in, out, T, d_bad);
            printf("two launches, %u workgroups each, nt intermediate %7.1f MB: %7.3f ms  %7.1f GB/s  bad=%llu\n", grid, T * kN * 8 / 1e6, best, alg / (best * 1e-3) / 1e9, bad);
        }
        for (int fv = 0; fv < 8; fv++) // 0: W8 sync, 1: W16 sync, 2: W8 nosync, 3: W16 nosync; 4..7: the same for k_fused2
            for (unsigned S : { 48u, 64u })
                for (unsigned LR : { 0x24u, 0x35u, 0x235u, 0x48u })
                {
                    const unsigned L = (LR >> 4) & 15, R = LR & 15;
                    const int mask = (LR >> 8) * 16; // 16: no publishes, 32: no polls, 48: neither (timing only)
                    if (mask && (fv & 3) >= 2)
                        continue;
                    if (mask == 16)
                        continue;
                    Args a{ in, out, mid, prog1, prog2, status, d_roles, (unsigned)T, S, R, 2 + mask, work, 1.0, 0.0 };
                    float best = 1e9f;
                    unsigned lost = 0, polls1 = 0, polls2 = 0;
                    for (int r = 0; r < reps + 1; r++)
                    {
                        CK(hipMemsetAsync(prog1, 0, 64 * kProgStride * 4));
                        CK(hipMemsetAsync(prog2, 0, 64 * 64 * 4));
                        CK(hipMemsetAsync(status, 0, 64));
                        CK(hipEventRecord(e0));
                        const dim3 g(S * 16), b(kT);
                        switch (fv)
                        {
                        case 0: if (S == 48) hipLaunchKernelGGL((k_fused<8, true, 3>), g, b, 0, 0, a, L); else hipLaunchKernelGGL((k_fused<8, true, 4>), g, b, 0, 0, a, L); break;
                        case 1: if (S == 48) hipLaunchKernelGGL((k_fused<16, true, 3>), g, b, 0, 0, a, L); else hipLaunchKernelGGL((k_fused<16, true, 4>), g, b, 0, 0, a, L); break;
                        case 2: if (S == 48) hipLaunchKernelGGL((k_fused<8, false, 3>), g, b, 0, 0, a, L); else hipLaunchKernelGGL((k_fused<8, false, 4>), g, b, 0, 0, a, L); break;
                        case 3: if (S == 48) hipLaunchKernelGGL((k_fused<16, false, 3>), g, b, 0, 0, a, L); else hipLaunchKernelGGL((k_fused<16, false, 4>), g, b, 0, 0, a, L); break;
                        case 4: if (S == 48) hipLaunchKernelGGL((k_fused2<8, true, 3>), g, b, 0, 0, a, L); else hipLaunchKernelGGL((k_fused2<8, true, 4>), g, b, 0, 0, a, L); break;
                        case 5: if (S == 48) hipLaunchKernelGGL((k_fused2<16, true, 3>), g, b, 0, 0, a, L); else hipLaunchKernelGGL((k_fused2<16, true, 4>), g, b, 0, 0, a, L); break;
                        case 6: if (S == 48) hipLaunchKernelGGL((k_fused2<8, false, 3>), g, b, 0, 0, a, L); else hipLaunchKernelGGL((k_fused2<8, false, 4>), g, b, 0, 0, a, L); break;
                        case 7: if (S == 48) hipLaunchKernelGGL((k_fused2<16, false, 3>), g, b, 0, 0, a, L); else hipLaunchKernelGGL((k_fused2<16, false, 4>), g, b, 0, 0, a, L); break;
                        }
                        CK(hipEventRecord(e1));
                        CK(hipEventSynchronize(e1));
                        float ms;
                        CK(hipEventElapsedTime(&ms, e0, e1));
                        if (r && ms < best)
                            best = ms;
                        unsigned h[16];
                        CK(hipMemcpy(h, status, 64, hipMemcpyDeviceToHost));
                        lost += h[0];
                        polls1 = h[1];
                        polls2 = h[2];
                        if (h[0])
                            break;
                    }
                    unsigned long long bad = verify(in, out, T, d_bad);
                    printf("fused%s %-10s mask %2d %2u teams (%4u WGs) lag %u, %u slots (ring %6.1f MB): %7.3f ms  %7.1f GB/s  bad=%llu lost=%u stall_polls p1=%u p2=%u\n",
                           fv >= 4 ? "2" : " ", (fv & 3) == 0 ? "W8 sync" : (fv & 3) == 1 ? "W16 sync" : (fv & 3) == 2 ? "W8 NOSYNC" : "W16 NOSYNC", mask, S, S * 16, L, R, (double)S * R * kN * 8 / 1e6, best, alg / (best * 1e-3) / 1e9,
                           bad, lost, polls1, polls2);
                }
        if (argc > 2)
            continue;
        struct Var { const char *name; int id; int sleep; };
        const Var vars[] = { { "W8 sc1 s2", 0, 0 }, { "W16 sc1 s2", 1, 0 }, { "W16 sc1 s32", 1, 1 }, { "W16 sc1 s127", 1, 2 }, { "W8 sc1 s127", 0, 2 }, { "W16 fences s127", 2, 2 },
                             { "W8 sc1 NOSYNC", 3, 0 }, { "W16 sc1 NOSYNC", 4, 0 }, { "W8 plain NOSYNC", 5, 0 }, { "W16 plain NOSYNC", 6, 0 }, { "W8 nt NOSYNC", 7, 0 }, { "W16 nt NOSYNC", 8, 0 } };
        for (const Var &var : vars)
            for (int mix = 1; mix < 2; mix++)
                for (unsigned S : { 24u, 32u })
                    for (unsigned R : { 4u, 8u })
                    {
                        const std::vector<uint32_t> roles = make_roles(S, mix);
                        CK(hipMemcpy(d_roles, roles.data(), roles.size() * 4, hipMemcpyHostToDevice));
                        Args a{ in, out, mid, prog1, prog2, status, d_roles, (unsigned)T, S, R, var.sleep, work, 1.0, 0.0 };
                        float best = 1e9f;
                        unsigned lost = 0, polls1 = 0, polls2 = 0;
                        for (int r = 0; r < reps + 1; r++)
                        {
                            CK(hipMemsetAsync(prog1, 0, 64 * 16 * 4));
                            CK(hipMemsetAsync(prog2, 0, 64 * 64 * 4));
                            CK(hipMemsetAsync(status, 0, 64));
                            CK(hipEventRecord(e0));
                            const dim3 g(S * 32), b(kT);
                            switch (var.id)
                            {
                            case 0: hipLaunchKernelGGL((k_ring<8, 0>), g, b, 0, 0, a); break;
                            case 1: hipLaunchKernelGGL((k_ring<16, 0>), g, b, 0, 0, a); break;
                            case 2: hipLaunchKernelGGL((k_ring<16, 1>), g, b, 0, 0, a); break;
                            case 3: hipLaunchKernelGGL((k_ring<8, 0, false>), g, b, 0, 0, a); break;
                            case 4: hipLaunchKernelGGL((k_ring<16, 0, false>), g, b, 0, 0, a); break;
                            case 5: hipLaunchKernelGGL((k_ring<8, 2, false>), g, b, 0, 0, a); break;
                            case 6: hipLaunchKernelGGL((k_ring<16, 2, false>), g, b, 0, 0, a); break;
                            case 7: hipLaunchKernelGGL((k_ring<8, 3, false>), g, b, 0, 0, a); break;
                            case 8: hipLaunchKernelGGL((k_ring<16, 3, false>), g, b, 0, 0, a); break;
                            }
                            CK(hipEventRecord(e1));
                            CK(hipEventSynchronize(e1));
                            float ms;
                            CK(hipEventElapsedTime(&ms, e0, e1));
                            if (r && ms < best)
                                best = ms;
                            unsigned h[16];
                            CK(hipMemcpy(h, status, 64, hipMemcpyDeviceToHost));
                            lost += h[0];
                            polls1 = h[1];
                            polls2 = h[2];
                            if (h[0])
                                break;
                        }
                        unsigned long long bad = verify(in, out, T, d_bad);
                        printf("ring %-17s mix=%d  %2u teams (%4u WGs) x %u slots (ring %6.1f MB): %7.3f ms  %7.1f GB/s  bad=%llu lost=%u stall_polls p1=%u p2=%u\n",
                               var.name, mix, S, S * 32, R, (double)S * R * kN * 8 / 1e6, best, alg / (best * 1e-3) / 1e9, bad, lost, polls1, polls2);
                        if (lost)
                            break;
                    }
    }
    return 0;
}
