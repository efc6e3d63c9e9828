// One-launch forward negacyclic NTT at N = 2^16 with the intermediate in a re-used, Infinity-Cache-resident ring (round 6).
// Replaces, for large plain batches of double-precision components, the two launches of ntt2_kernels.hip (ntt2_fwd_p1 / ntt2_fwd_p2):
//   ntt_negacyclic_harvey[_lazy]   native/src/seal/util/ntt.cpp:394-437, DWTHandler::transform_to_rev native/src/seal/util/dwthandler.h:94-191
// Same twiddles, same stage order, same exact double-precision residue arithmetic (field.h), canonical results: bit-identical words.
// The skeleton of this kernel without arithmetic, with every word checked, is tools/microbench/ring_handoff.hip.
#include "ntt2_device.h"
#include <algorithm>
#include <cstdio>
#include <mutex>

namespace sealhip
{
#if defined(__HIPCC__) // (the fiber emulator of the CPU tests runs one workgroup at a time: a kernel whose workgroups wait for each other is not for it)
    namespace
    {
        // ---------------------------------------------------------------------------------------
        // ONE-launch forward transform at N = 2^16 whose intermediate lives in a re-used ring (round 6)
        // ---------------------------------------------------------------------------------------
        // Why: a 2^16-point transform does not fit a CU, so its words cross the memory system twice in each direction; what CAN be
        // taken away is the HBM half of the inner crossing.  The 256 MiB Infinity Cache keeps lines that are written and re-read
        // soon afterwards - if they are written and read with TEMPORAL accesses and the addresses are re-used
        // (tools/microbench/mall_bw.hip, profiles/r06_mall_bw.txt: a copy through a re-used ring moves 3.5 TB/s algorithmic against
        // 2.6 through a one-shot intermediate; with non-temporal accesses, the default of the two-launch engine, nothing).
        // How (tools/microbench/ring_handoff.hip is this kernel without the arithmetic, profiles/r06_ring_handoff.txt):
        //   * the 16 workgroups (tile t = 0..15) of a TEAM share the transforms (component c, outer items z = slice + k Z), k = 0, 1, ...
        //     Iteration k of a workgroup: pass-1 tile t (column tile) of transform k -> ring slot k mod R, then pass-2 tile t (row
        //     tile) of transform k - L out of slot (k - L) mod R.  Both roles in every workgroup: the team advances at one pace,
        //     twiddles of both passes stay resident (they depend on (prime, tile) only), the waits are satisfied iterations ahead.
        //   * hand-over (MI355X guide, inter-workgroup visibility, "valid forms"): ring stores are write-through (sc1), ring loads
        //     bypass the L1 (sc1) - a slot is re-used, so nothing a CU cached of its previous occupant may be served.  A workgroup
        //     publishes { pass-1 iterations whose stores are in memory, pass-2 iterations whose loads have landed } by ONE lane after
        //     the first barrier of the NEXT iteration's exchange: every wave has by then waited for its next input, which it requested
        //     after its ring stores (vector memory operations of a wave complete in order), and has consumed its ring loads.
        //   * ONE poller per workgroup: wave 0 loads the team's sixteen pairs (128 bytes) after the second barrier, the verdict
        //     reaches the other waves through LDS at the next first barrier.  (Per-wave polls of per-wave words cost 25 % of the
        //     kernel in the skeleton: the progress lines of a team sit in one memory channel.)  A team's pairs have 4 KiB + 256 B
        //     to themselves so that the teams' polls spread over the channels.
        //   * the grid never exceeds what the chip holds at once (occupancy query), and launches of this kernel are serialised
        //     across streams by an event chain (launch_ring): a team whose members are not all resident can only wait for
        //     workgroups that finish without it.  A dependency that does not arrive within ~seconds traps (loud, bounded).
        //   * intermediate in "pair order": rows 2j, 2j+1 of a column adjacent, so that both sides move 16 bytes per lane
        //     (8-byte sc1 accesses run at about half the rate); pass 2 gets the odd row from its partner lane with
        //     v_permlane16_swap.  Word (row-in-tile r, column-in-block c) of block (hg, cg) at (hg*16 + cg)*256 + (r >> 1)*32 + c*2 + (r & 1).
        //   * LDS: pass 1's exchange buffer and pass 2's wave-local buffers share 4 x 1152 words - wave w reads only rows
        //     64 w .. 64 w + 63 of the exchange, which are laid over ITS OWN wave-local buffer, so two barriers per iteration do.
        //     Behind them the 240 row-shared twiddles of pass 2's first phase and the 240 of pass 1's second phase (39.9 KiB in all).
        struct RingArgs
        {
            uint64_t *ring;   // [teams][R][N] words, pair order
            unsigned *prog;   // [teams] blocks of kRingProgStride words: [tile][2], zeroed before the launch; then 16 status words
            unsigned teams_per_comp; // Z
            unsigned R, L;
            unsigned flags; // measurement builds only (SEALHIP_RING_FLAGS): 1 = never wait (wrong words, timing only)
        };
        constexpr unsigned kRingProgStride = (4096 + 256) / 4;
        constexpr unsigned kRingQuarter = 4 * kRowWords; // words of LDS per wave

        template <int WAVES>
        __global__ void __launch_bounds__(kThreads, WAVES) ntt2_fwd_ring(FwdArgs a, RingArgs r)
        {
#if defined(__HIP_DEVICE_COMPILE__)
            typedef Field<true> F;
            constexpr int D1 = 8;
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            HIP_DYNAMIC_SHARED(uint64_t, lds)
            __shared__ unsigned s_have[2]; // min over the team of { pass-1 complete, pass-2 landed } as polled by wave 0
            const unsigned tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
            const unsigned team = blockIdx.x >> 4, tile = blockIdx.x & 15;
            const unsigned Z = r.teams_per_comp, R = r.R, L = r.L;
            const unsigned comp = team / Z + a.comp0, slice = team % Z;
            const unsigned prime = SHL_UNIFORM(a.comp_prime ? a.comp_prime[comp] : a.prime_first + comp);
            const F::Mod m = F::make_mod(ld_uniform_mod(&a.t.mods[prime]), ld_uniform_fpd(&a.t.fpd[prime]));
            const double *tab = tw_table<true>(a.t, false, prime);
            const unsigned iters = slice < a.nouter ? (a.nouter - slice + Z - 1) / Z : 0;
            uint64_t *ring = r.ring + (((size_t)team * R) << 16);
            unsigned *prog = r.prog + (size_t)team * kRingProgStride;
            unsigned *status = r.prog + (size_t)gridDim.x / 16 * kRingProgStride;
            const unsigned c = tid & 15, hi = tid >> 4; // pass 1: column c of column tile `tile`; rows e*16 + hi, then hi*16 + rb
            const unsigned u = tid >> 4, v = tid & 15;  // pass 2: row u, column-in-block v of the sixteen blocks of row tile `tile`
            uint64_t *lds_wave = lds + wave * kRingQuarter;
            // resident twiddles: pass 1 phase B in LDS (they depend on (prime, hi) only: slot (1 << t) + g of row group hi at
            // tw1[slot * 16 + hi], sixteen lanes read one word), pass 2 phase A in LDS (row-shared), pass 2 phase B in registers
            TwRegs<true> pre_b;
            double *twa = reinterpret_cast<double *>(lds + 4 * kRingQuarter);
            double *tw1 = twa + 240;
            stage_twa<D1>(twa, tab, tile, tid);
            if (tid >= 16)
            {
                const unsigned slot = tid >> 4, h = tid & 15, t = 31 - __builtin_clz(slot), g = slot - (1u << t);
                tw1[tid] = tab[(1u << (4 + t)) + (h << t) + g];
            }
            load_tw<true, 4>(pre_b, tab, [&](int t) { return (1u << (D1 + 4 + t)) + (((tile * 16 + u) * 16 + v) << t); });
            if (tid < 2)
                s_have[tid] = 0;
            uint64_t nxt[16];
            const uint64_t *in0 = a.data + ((size_t)comp << 16) + tile * 16 + c;
            auto fetch_a = [&](unsigned k) {
                const uint64_t *in = in0 + (size_t)(slice + k * Z) * a.outer_stride;
#pragma unroll
                for (int e = 0; e < 16; e++)
                    nxt[e] = mid_ld<16>(in + (size_t)(e * 16 + hi) * 256);
            };
            auto window = [](const uint64_t *p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<uint64_t *>(p), 0, 0x7fffffff, 0x00020000); };
            auto fetch_b = [&](unsigned j) {
                // rows (u & ~1, u | 1) of column v of the blocks (u & 1) * 8 + j2
                const __amdgpu_buffer_rsrc_t rs = window(ring + ((size_t)(j % R) << 16) + (size_t)tile * 4096);
#pragma unroll
                for (int j2 = 0; j2 < 8; j2++)
                {
                    const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)((u >> 1) * 256 + v * 16), (int)(((u & 1) * 8 + j2) * 2048), 16);
                    nxt[2 * j2] = (uint64_t)w.x | ((uint64_t)w.y << 32);
                    nxt[2 * j2 + 1] = (uint64_t)w.z | ((uint64_t)w.w << 32);
                }
            };
            // slow path (rare): this wave polls the team's pairs itself until word `which` of all sixteen is >= target
            auto spin = [&](unsigned which, unsigned target) {
                unsigned spins = 0;
                for (;;)
                {
                    const unsigned have = lane < 16 ? __hip_atomic_load(prog + lane * 2 + which, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : target;
                    if (__all(have >= target))
                        break;
                    __builtin_amdgcn_s_sleep(32);
                    if (++spins > (1u << 21))
                    {
                        if (lane == 0)
                            __hip_atomic_fetch_add(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __builtin_trap(); // a lost dependency: fail the launch instead of hanging the queue
                    }
                }
                if (lane == 0)
                {
                    __hip_atomic_fetch_add(status + 1 + which, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (spins)
                        __hip_atomic_fetch_add(status + 3 + which, spins, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            };
            unsigned polled = 0; // wave 0, lanes 0..31: word `lane` of the team's progress block as of the last poll
            __syncthreads();     // twa, s_have
            if (iters)
                fetch_a(0);
            for (unsigned k = 0; k < iters + L; k++)
            {
                const bool has1 = k < iters, has2 = k >= L && iters != 0;
                const unsigned j = k - L;
                double x[16];
                if (has1)
                {
#pragma unroll
                    for (int e = 0; e < 16; e++)
                        x[e] = F::from_canon(nxt[e], m);
                    // pass 1, phase A: stages 0..3, wave-uniform twiddles
                    phase_fwd_end<true, 4, true>(x, m, [&](int t, int g) { return ld_uniform(tab, (1u << t) + g); });
                }
                else
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (wave == 0 && k > 0)
                {
                    // verdict of the poll requested after the second barrier of the previous iteration: min over the team, per word
                    unsigned mn = lane < 32 ? polled : 0xffffffffu;
#pragma unroll
                    for (int sft = 2; sft < 32; sft <<= 1)
                        mn = min(mn, (unsigned)__shfl_xor((int)mn, sft));
                    if (lane < 2)
                        s_have[lane] = mn;
                }
                __syncthreads(); // B1: the exchange buffer is free, the verdict is in LDS
                if (tid == 0 && k > 0)
                {
                    // pass-1 iterations 0 .. k-1 are in memory and pass-2 iterations 0 .. j-1 have landed, for every wave of this workgroup
                    __hip_atomic_store(prog + tile * 2, k < iters ? k : iters, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (k > L)
                        __hip_atomic_store(prog + tile * 2 + 1, j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (has2 && j < iters)
                {
                    if (s_have[0] < j + 1 && !(r.flags & 1))
                        spin(0, j + 1);
                    fetch_b(j);
                }
                if (has1)
                {
                    // row a*16 + hi -> quarter a >> 2, row-in-quarter (a & 3)*16 + hi
#pragma unroll
                    for (int e = 0; e < 16; e++)
                        lds[(e >> 2) * kRingQuarter + ((e & 3) * 16 + hi) * 17 + c] = F::raw(x[e]);
                }
                const unsigned have2 = s_have[1];
                __syncthreads(); // B2
                if (wave == 0 && lane < 32)
                    polled = __hip_atomic_load(prog + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (has1)
                {
                    // rows hi*16 + rb: this wave's own quarter
#pragma unroll
                    for (int rb = 0; rb < 16; rb++)
                        x[rb] = F::unraw(lds_wave[((hi & 3) * 16 + rb) * 17 + c]);
                    // phase B: stages 4..7, resident twiddles; the intermediate leaves with |x| <= q/2
                    phase_fwd_end<true, 4, true>(x, m, [&](int t, int g) { return tw1[(((1 << t) + g) << 4) + hi]; });
                    // the slot's previous occupant (iteration k - R) must have been read by the whole team
                    if (k >= R && have2 < k - R + 1 && !(r.flags & 1))
                        spin(1, k - R + 1);
                    const __amdgpu_buffer_rsrc_t rs = window(ring + ((size_t)(k % R) << 16) + (size_t)(hi * 16 + tile) * 256);
#pragma unroll
                    for (int j2 = 0; j2 < 8; j2++)
                    {
                        const uint64_t w0 = F::raw(x[2 * j2]), w1 = F::raw(x[2 * j2 + 1]);
                        __builtin_amdgcn_raw_buffer_store_b128(u32x4{ (uint32_t)w0, (uint32_t)(w0 >> 32), (uint32_t)w1, (uint32_t)(w1 >> 32) }, rs, (int)(c * 16), j2 * 256, 16);
                    }
                }
                const bool run2 = has2 && j < iters;
                if (run2)
                {
                    // even rows keep their first words (row u of blocks 0..7) and take the odd partner's first words (row u of blocks
                    // 8..15); odd rows take the even partner's second words and keep their own: odd rows of X <-> even rows of Y
#pragma unroll
                    for (int j2 = 0; j2 < 8; j2++)
                    {
                        const auto lo = __builtin_amdgcn_permlane16_swap((uint32_t)nxt[2 * j2], (uint32_t)nxt[2 * j2 + 1], false, false);
                        const auto hi32 = __builtin_amdgcn_permlane16_swap((uint32_t)(nxt[2 * j2] >> 32), (uint32_t)(nxt[2 * j2 + 1] >> 32), false, false);
                        x[j2] = F::unraw((uint64_t)lo[0] | ((uint64_t)hi32[0] << 32));
                        x[8 + j2] = F::unraw((uint64_t)lo[1] | ((uint64_t)hi32[1] << 32));
                    }
                    // the ring loads have landed - and with them everything this wave issued before them - before the next input is requested
                    asm volatile("" ::"v"(x[15]), "v"(x[7]) : "memory");
                }
                if (k + 1 < iters)
                    fetch_a(k + 1);
                if (run2)
                {
                    p2_tile<true, D1, false, false, true, true>(x, m, tab, twa, nullptr, lds_wave, tile, tid, nullptr, &pre_b);
                    constexpr int BOUT = kP2Out<0, D1>;
                    uint64_t val[16];
#pragma unroll
                    for (int e = 0; e < 16; e++)
                        val[e] = a.lazy ? fwd_out_lazy<true, 0, BOUT>(x[e], m) : fwd_out_canon<true, 0, BOUT>(x[e], m);
                    const size_t row0 = ((size_t)comp << 16) + ((size_t)(tile * 16 + wave * 4) << 8);
                    store_rows(val, lds_wave, a.data + (size_t)(slice + j * Z) * a.outer_stride + row0, tid);
                }
            }
#endif
        }

        // ---------------------------------------------------------------------------------------
        // The SPECIALISED form (SEALHIP_NTT_RING=2): pass-1 workgroups and pass-2 workgroups in one launch
        // ---------------------------------------------------------------------------------------
        // The fused kernel above pays for doing both passes in one workgroup with its registers (202 VGPRs: two waves per SIMD).  Here a
        // workgroup has ONE role - a team is 16 pass-1 workgroups (column tile) + 16 pass-2 workgroups (row tile) of one (component,
        // slice) - so either role keeps the register budget of the separate kernels (the kernel is compiled for four waves per SIMD) and
        // a CU holds both roles side by side: whichever role is behind gets the CU, the other one sleeps on its progress word.
        // Pass 1 runs ahead of pass 2 by at most R slots.  Same ring, same pair order, same write-through / L1-bypass hand-over, same
        // one-publisher / one-poller-per-workgroup rule as above; what differs is that ONE side waits by construction, so the slow
        // path is built for it: wave 0 polls with long sleeps, the other waves wait at a workgroup barrier.
        //   prog[team][tile][0] = iterations whose pass-1 tile `tile` is in memory, [1] = iterations whose pass-2 tile has landed
        // Block -> (role, team, tile): layers of 256 blocks alternate the roles (with dispatch going round the 8 XCDs x 32 CUs every
        // CU gets both roles), the remainder is split in halves.
        template <int WAVES>
        __global__ void __launch_bounds__(kThreads, WAVES) ntt2_fwd_ring2(FwdArgs a, RingArgs r)
        {
#if defined(__HIP_DEVICE_COMPILE__)
            typedef Field<true> F;
            constexpr int D1 = 8;
            typedef Geo<D1> G;
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            HIP_DYNAMIC_SHARED(uint64_t, lds)
            __shared__ unsigned s_have[2]; // pass 1: [k & 1] unused, [*] = min landed; pass 2: [k & 1] = min complete
            const unsigned tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
            const unsigned Z = r.teams_per_comp, R = r.R;
            const unsigned per_role = gridDim.x >> 1; // 16 * teams
            unsigned role, idx;
            {
                const unsigned b = blockIdx.x, pairs = gridDim.x / 512, rest = gridDim.x - pairs * 512;
                if (b < pairs * 512)
                {
                    const unsigned layer = b >> 8;
                    role = layer & 1;
                    idx = (layer >> 1) * 256 + (b & 255);
                }
                else
                {
                    const unsigned rb = b - pairs * 512;
                    role = rb >= rest / 2 ? 1u : 0u;
                    idx = pairs * 256 + rb - role * (rest / 2);
                }
            }
            const unsigned team = idx >> 4, tile = idx & 15;
            const unsigned comp = team / Z + a.comp0, slice = team % Z;
            const unsigned prime = SHL_UNIFORM(a.comp_prime ? a.comp_prime[comp] : a.prime_first + comp);
            const F::Mod m = F::make_mod(ld_uniform_mod(&a.t.mods[prime]), ld_uniform_fpd(&a.t.fpd[prime]));
            const double *tab = tw_table<true>(a.t, false, prime);
            const unsigned iters = slice < a.nouter ? (a.nouter - slice + Z - 1) / Z : 0;
            uint64_t *ring = r.ring + (((size_t)team * R) << 16);
            unsigned *prog = r.prog + (size_t)team * kRingProgStride;
            unsigned *status = r.prog + (size_t)(per_role >> 4) * kRingProgStride;
            auto window = [](const uint64_t *p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<uint64_t *>(p), 0, 0x7fffffff, 0x00020000); };
            // the whole workgroup waits until word `which` of all sixteen pairs of the team is >= target: wave 0 polls, the others sit at the barrier
            auto wait_team = [&](unsigned which, unsigned target) {
                if (wave == 0 && !(r.flags & 1))
                {
                    unsigned spins = 0;
                    for (;;)
                    {
                        const unsigned have = lane < 16 ? __hip_atomic_load(prog + lane * 2 + which, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : target;
                        if (__all(have >= target))
                            break;
                        __builtin_amdgcn_s_sleep(64);
                        if (++spins > (1u << 20))
                        {
                            if (lane == 0)
                                __hip_atomic_fetch_add(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __builtin_trap(); // a lost dependency: fail the launch instead of hanging the queue
                        }
                    }
                    if (lane == 0)
                    {
                        __hip_atomic_fetch_add(status + 1 + which, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (spins)
                            __hip_atomic_fetch_add(status + 3 + which, spins, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                __syncthreads();
            };
            // wave 0: min over the sixteen workgroups of word `which` of what was polled (lanes 0..31 hold the block's 32 words)
            auto team_min = [&](unsigned polled, unsigned which) -> unsigned {
                unsigned mn = (lane < 32 && (lane & 1) == which) ? polled : 0xffffffffu;
#pragma unroll
                for (int sft = 1; sft < 32; sft <<= 1)
                    mn = min(mn, (unsigned)__shfl_xor((int)mn, sft));
                return mn;
            };
            if (tid < 2)
                s_have[tid] = 0;
            uint64_t nxt[16];
            unsigned polled = 0;
            if (role == 0)
            {
                // ---------------- pass 1: column tile `tile` of every transform of the team
                const unsigned c = tid & 15, hi = tid >> 4;
                TwRegs<true> tw1;
                p1_load_tw<true, D1>(tw1, tab, tid);
                const uint64_t *in0 = a.data + ((size_t)comp << 16) + tile * 16 + c;
                auto fetch = [&](unsigned k) {
                    const uint64_t *in = in0 + (size_t)(slice + k * Z) * a.outer_stride;
#pragma unroll
                    for (int e = 0; e < 16; e++)
                        nxt[e] = mid_ld<16>(in + (size_t)(e * 16 + hi) * 256);
                };
                __syncthreads(); // s_have
                if (iters)
                    fetch(0);
                for (unsigned k = 0; k < iters; k++)
                {
                    double x[16];
#pragma unroll
                    for (int e = 0; e < 16; e++)
                        x[e] = F::from_canon(nxt[e], m);
                    if (k + 1 < iters)
                        fetch(k + 1);
                    phase_fwd_end<true, 4, true>(x, m, [&](int t, int g) { return ld_uniform(tab, (1u << t) + g); });
                    if (wave == 0 && k > 0)
                    {
                        const unsigned mn = team_min(polled, 1);
                        if (lane == 0)
                            s_have[0] = mn;
                    }
                    // this wave's ring stores of iteration k - 1 are in memory: they were issued before the 16 prefetch loads above, and a
                    // wave's vector memory operations complete in order
                    if (k + 1 < iters)
                        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                    else
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads(); // B1
                    if (tid == 0 && k > 0)
                        __hip_atomic_store(prog + tile * 2, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                    for (int e = 0; e < 16; e++)
                        lds[(e * 16 + hi) * G::CP + c] = F::raw(x[e]);
                    const unsigned landed = s_have[0];
                    __syncthreads(); // B2
                    if (wave == 0 && lane < 32)
                        polled = __hip_atomic_load(prog + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                    for (int rb = 0; rb < 16; rb++)
                        x[rb] = F::unraw(lds[(hi * 16 + rb) * G::CP + c]);
                    phase_fwd_end<true, 4, true>(x, m, [&](int t, int g) { return tw1.get((1 << t) + g); });
                    // the slot's previous occupant (iteration k - R) must have been read by all sixteen pass-2 workgroups
                    if (k >= R && landed < k - R + 1)
                        wait_team(1, k - R + 1);
                    const __amdgpu_buffer_rsrc_t rs = window(ring + ((size_t)(k % R) << 16) + (size_t)(hi * 16 + tile) * 256);
#pragma unroll
                    for (int j2 = 0; j2 < 8; j2++)
                    {
                        const uint64_t w0 = F::raw(x[2 * j2]), w1 = F::raw(x[2 * j2 + 1]);
                        __builtin_amdgcn_raw_buffer_store_b128(u32x4{ (uint32_t)w0, (uint32_t)(w0 >> 32), (uint32_t)w1, (uint32_t)(w1 >> 32) }, rs, (int)(c * 16), j2 * 256, 16);
                    }
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0 && iters)
                    __hip_atomic_store(prog + tile * 2, iters, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            else
            {
                // ---------------- pass 2: row tile `tile`; thread (u, v) ends up with row u, column-in-block v of the sixteen blocks
                const unsigned u = tid >> 4, v = tid & 15;
                uint64_t *lds_wave = lds + wave * (4 * kRowWords);
                TwRegs<true> pre_b;
                double *twa = reinterpret_cast<double *>(lds + kLds2Words);
                stage_twa<D1>(twa, tab, tile, tid);
                load_tw<true, 4>(pre_b, tab, [&](int t) { return (1u << (D1 + 4 + t)) + (((tile * 16 + u) * 16 + v) << t); });
                auto fetch = [&](unsigned j) {
                    const __amdgpu_buffer_rsrc_t rs = window(ring + ((size_t)(j % R) << 16) + (size_t)tile * 4096);
#pragma unroll
                    for (int j2 = 0; j2 < 8; j2++)
                    {
                        const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)((u >> 1) * 256 + v * 16), (int)(((u & 1) * 8 + j2) * 2048), 16);
                        nxt[2 * j2] = (uint64_t)w.x | ((uint64_t)w.y << 32);
                        nxt[2 * j2 + 1] = (uint64_t)w.z | ((uint64_t)w.w << 32);
                    }
                };
                __syncthreads(); // twa, s_have
                if (iters)
                {
                    wait_team(0, 1);
                    fetch(0);
                }
                for (unsigned k = 0; k < iters; k++)
                {
                    double x[16];
#pragma unroll
                    for (int j2 = 0; j2 < 8; j2++)
                    {
                        const auto lo = __builtin_amdgcn_permlane16_swap((uint32_t)nxt[2 * j2], (uint32_t)nxt[2 * j2 + 1], false, false);
                        const auto hi32 = __builtin_amdgcn_permlane16_swap((uint32_t)(nxt[2 * j2] >> 32), (uint32_t)(nxt[2 * j2 + 1] >> 32), false, false);
                        x[j2] = F::unraw((uint64_t)lo[0] | ((uint64_t)hi32[0] << 32));
                        x[8 + j2] = F::unraw((uint64_t)lo[1] | ((uint64_t)hi32[1] << 32));
                    }
                    asm volatile("" ::"v"(x[15]), "v"(x[7]) : "memory"); // this wave's ring loads of iteration k have landed
                    if (wave == 0 && k > 0)
                    {
                        const unsigned mn = team_min(polled, 0);
                        if (lane == 0)
                            s_have[k & 1] = mn;
                    }
                    __syncthreads(); // every wave has its tile: the slot may be rewritten
                    if (tid == 0)
                        __hip_atomic_store(prog + tile * 2 + 1, k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (k + 1 < iters)
                    {
                        if (k == 0 || s_have[k & 1] < k + 2)
                            wait_team(0, k + 2);
                        fetch(k + 1);
                        if (wave == 0 && lane < 32)
                            polled = __hip_atomic_load(prog + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    p2_tile<true, D1, false, false, true, true>(x, m, tab, twa, nullptr, lds_wave, tile, tid, nullptr, &pre_b);
                    constexpr int BOUT = kP2Out<0, D1>;
                    uint64_t val[16];
#pragma unroll
                    for (int e = 0; e < 16; e++)
                        val[e] = a.lazy ? fwd_out_lazy<true, 0, BOUT>(x[e], m) : fwd_out_canon<true, 0, BOUT>(x[e], m);
                    const size_t row0 = ((size_t)comp << 16) + ((size_t)(tile * 16 + wave * 4) << 8);
                    store_rows(val, lds_wave, a.data + (size_t)(slice + k * Z) * a.outer_stride + row0, tid);
                }
            }
#endif
        }
        // ---- host side of ntt2_fwd_ring
#ifndef SEALHIP_RING_WAVES
#define SEALHIP_RING_WAVES 3 // waves per SIMD = workgroups per CU the kernel is compiled for
#endif
#ifndef SEALHIP_RING2_WAVES
#define SEALHIP_RING2_WAVES 4
#endif
        // L, R: pass 2 runs three iterations behind pass 1, six slots per team.  What a workgroup polls after the second barrier of iteration k
        // is what its team published at the first barrier of iteration k (k pass-1 iterations complete, k - L pass-2 iterations landed) - or
        // of iteration k - 1 by members up to one iteration behind.  Iteration k + 1 needs k + 2 - L complete and k + 2 - R landed: L >= 3 and
        // R >= L + 3 keep the polled verdict sufficient for a team whose members drift by up to an iteration (L = 2, R = 4: 85 % of the waits
        // took the slow path - a fresh poll behind every outstanding store - and the kernel ran three times slower).
        constexpr unsigned kRingLag = 3, kRingSlots = 6;
        constexpr size_t kRingLdsBytes = (4 * kRingQuarter + 240 + 256) * 8;
        constexpr size_t kRing2LdsBytes = (kLds2Words + 240) * 8; // pass 2's wave-local buffers + its row-shared twiddles (pass 1's exchange is smaller)
        struct RingPlan
        {
            unsigned capacity = 0; // workgroups of the selected kernel the device holds at once (0: kernel not usable)
            int mode = 0;          // SEALHIP_NTT_RING: 1 = fused workgroups (ntt2_fwd_ring), 2 = specialised workgroups (ntt2_fwd_ring2)
        };
        inline const RingPlan &ring_plan()
        {
            static const RingPlan plan = [] {
                RingPlan p;
                // OPT-IN (SEALHIP_NTT_RING=1): measured on MI355X the kernel is bit-exact but SLOWER than the two launches (3.4 - 5 ms against
                // 2.7 ms per 7168 transforms, profiles/r06_ring_kernel.txt): one workgroup doing both passes needs 202 VGPRs - two waves per
                // SIMD, or three with spills whose reloads queue behind the prefetch - against four for the separate passes, and the
                // teams' lock step turns every slow workgroup into a convoy.  Kept as the tested record of the experiment.
                const char *env = std::getenv("SEALHIP_NTT_RING");
                if (!env || std::atoi(env) == 0)
                    return p;
                const int mode = std::atoi(env) == 2 ? 2 : 1;
                int dev = 0, cus = 0, per_cu = 0;
                if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
                    return p;
                const int waves = mode == 2 ? SEALHIP_RING2_WAVES : SEALHIP_RING_WAVES;
                const hipError_t e = mode == 2
                    ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(&ntt2_fwd_ring2<SEALHIP_RING2_WAVES>), kThreads, kRing2LdsBytes)
                    : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(&ntt2_fwd_ring<SEALHIP_RING_WAVES>), kThreads, kRingLdsBytes);
                if (e != hipSuccess)
                    return p;
                if (per_cu > waves)
                    per_cu = waves;
                if (per_cu > 0 && cus > 0)
                {
                    p.capacity = (unsigned)per_cu * (unsigned)cus;
                    p.mode = mode;
                }
                return p;
            }();
            return plan;
        }
        // teams per component for a run of `nc` double-precision components over `nouter` outer items; 0 = the run stays on the two launches
        inline unsigned teams_per_comp(unsigned nc, unsigned nouter)
        {
            const unsigned cap = ring_plan().capacity;
            if (!cap || !nc)
                return 0;
            unsigned z = cap / ((ring_plan().mode == 2 ? 32 : 16) * nc); // a team is 16 workgroups, or 16 + 16 with one role each
            // every team needs a few iterations for the pipeline (lag + slots) to pay; small batches keep the two-launch kernels
            const unsigned min_iters = 2 * (kRingLag + kRingSlots);
            if (z > nouter / min_iters)
                z = nouter / min_iters;
            return z;
        }
        inline size_t words_for(unsigned nc, unsigned z)
        {
            const size_t teams = (size_t)nc * z;
            return ((teams * kRingSlots) << 16) + (teams * kRingProgStride + 16 + 1) / 2;
        }
        // Launches of the ring kernel are serialised across the streams of the process (one event chain per device): two of them side by
        // side could each hold part of the other's teams off the chip.
        struct RingChain
        {
            std::mutex mu;
            hipEvent_t last = nullptr;
            bool recorded = false;
        };
        inline RingChain &ring_chain()
        {
            static RingChain c;
            return c;
        }
    } // namespace

    unsigned ntt2_ring_teams(unsigned nc, unsigned nouter)
    {
        return teams_per_comp(nc, nouter);
    }
    size_t ntt2_ring_run_words(unsigned nc, unsigned teams_per_component)
    {
        return words_for(nc, teams_per_component);
    }
    bool ntt2_ring_stream_ok(hipStream_t st)
    {
        // not while the stream is being captured: the launch waits for an event of another stream's launch (the serialising chain)
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        return hipStreamIsCapturing(st, &cap) == hipSuccess && cap == hipStreamCaptureStatusNone;
    }
    hipError_t ntt2_ring_launch(const NttTables &t, const NttRingRun &run, hipStream_t st)
    {
        const unsigned z = run.teams_per_comp, teams = run.nc * z;
        if (!z || words_for(run.nc, z) > run.ring_words)
            return hipErrorInvalidValue;
        FwdArgs f{};
        f.data = run.data;
        f.outer_stride = run.outer_stride;
        f.comp_prime = run.comp_prime;
        f.prime_first = run.prime_first;
        f.ncomp = run.ncomp;
        f.comp0 = run.c0;
        f.nouter = run.nouter;
        f.lazy = run.lazy;
        f.t = t;
        RingArgs ra;
        ra.ring = run.ring;
        ra.prog = reinterpret_cast<unsigned *>(run.ring + (((size_t)teams * kRingSlots) << 16));
        ra.teams_per_comp = z;
        ra.R = kRingSlots;
        ra.L = kRingLag;
        static const unsigned dev_flags = std::getenv("SEALHIP_RING_FLAGS") ? (unsigned)std::atoi(std::getenv("SEALHIP_RING_FLAGS")) : 0u;
        ra.flags = dev_flags;
        RingChain &ch = ring_chain();
        std::lock_guard<std::mutex> lock(ch.mu);
        hipError_t e;
        if (!ch.last && (e = hipEventCreateWithFlags(&ch.last, hipEventDisableTiming)) != hipSuccess)
            return e;
        if (ch.recorded && (e = hipStreamWaitEvent(st, ch.last, 0)) != hipSuccess)
            return e;
        if ((e = hipMemsetAsync(ra.prog, 0, ((size_t)teams * kRingProgStride + 16) * 4, st)) != hipSuccess)
            return e;
        if (ring_plan().mode == 2)
            hipLaunchKernelGGL((ntt2_fwd_ring2<SEALHIP_RING2_WAVES>), dim3(teams * 32), dim3(kThreads), kRing2LdsBytes, st, f, ra);
        else
            hipLaunchKernelGGL((ntt2_fwd_ring<SEALHIP_RING_WAVES>), dim3(teams * 16), dim3(kThreads), kRingLdsBytes, st, f, ra);
        if ((e = hipGetLastError()) != hipSuccess)
            return e;
        if ((e = hipEventRecord(ch.last, st)) != hipSuccess)
            return e;
        ch.recorded = true;
        static const bool debug = std::getenv("SEALHIP_RING_DEBUG") != nullptr; // diagnostics: waits that took the slow path, per launch
        if (debug)
        {
            unsigned h[5] = { 0, 0, 0, 0, 0 };
            if (hipStreamSynchronize(st) == hipSuccess && hipMemcpy(h, ra.prog + (size_t)teams * kRingProgStride, sizeof h, hipMemcpyDeviceToHost) == hipSuccess)
                std::fprintf(stderr, "ntt2_fwd_ring: mode %d, %u teams, %u outer items, lost %u, slow waits (wave level) for pass 1 %u (%u sleeps) / for a free slot %u (%u sleeps)\n", ring_plan().mode, teams, run.nouter, h[0], h[1], h[3], h[2], h[4]);
        }
        return hipSuccess;
    }
#else
    unsigned ntt2_ring_teams(unsigned, unsigned)
    {
        return 0;
    }
    size_t ntt2_ring_run_words(unsigned, unsigned)
    {
        return 0;
    }
    bool ntt2_ring_stream_ok(hipStream_t)
    {
        return false;
    }
    hipError_t ntt2_ring_launch(const NttTables &, const NttRingRun &, hipStream_t)
    {
        return hipErrorInvalidValue;
    }
#endif
} // namespace sealhip
