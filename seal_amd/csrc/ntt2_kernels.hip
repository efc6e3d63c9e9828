// Two-pass negacyclic NTT engine for 2^12 <= N <= 2^16 and the fused key-switching kernels.
//
// What it replaces in the reference (results are canonical residues, hence bit-identical):
//   ntt_negacyclic_harvey[_lazy]              native/src/seal/util/ntt.cpp:394-437 (dwthandler.h:94-191)
//   the I/J loop of switch_key_inplace        native/src/seal/evaluator.cpp:2663-2755
//
// Decomposition (D1 = n - 8, forward, Cooley-Tukey, natural -> bit-reversed order as the reference):
//   pass 1: stages 0..D1-1 on strided "columns" of the N/256 x 256 matrix; a workgroup owns a
//           tile of 2^D1 rows x C columns (C = 4096 / 2^D1, always 4096 coefficients, 256 threads,
//           16 coefficients per thread); D1-4 stages in registers, one LDS exchange, 4 stages;
//           the tile is written to a private intermediate buffer in *tile order*
//              mid[(hg*16 + col_hi)*256 + h_lo*16 + col_lo],  h = 16*hg + h_lo the row, col = 16*col_hi + col_lo
//   pass 2: stages D1..D1+7 on 16 contiguous rows (hg) of 256 coefficients; thanks to the tile
//           order every load is a fully coalesced 2 KiB wave access; 4 stages, a wave-local LDS
//           exchange (the 16 lanes that own one row sit in one wavefront: no s_barrier), 4 stages.
// Twiddles are the reference's merged psi powers in bit-reversed order (ntt.cpp:273-278), so the
// two passes compose without a twist multiply.  Phase-A twiddles of pass 1 are wave-uniform and
// live in SGPRs; the others are staged through LDS or loaded per thread.
//
// Arithmetic is a template parameter (field.h): the general 64-bit Shoup/Harvey back end, or the
// exact double-precision back end for primes below 2^50 (3x fewer VALU instructions per butterfly).
//
// Fused key switching (ks1/ks2): pass 1 loops over all target moduli I for one decomposition
// digit J (the digit is read once), pass 2 loops over the digits J of one target modulus I, keeps
// the 2x16 running sums of (digit x key) per thread in registers and writes only the reduced sums:
// the K(K+1) transformed digits never reach HBM in NTT form.
#include "ntt2_device.h"
#include <algorithm>
#include <atomic>
#include <mutex>
#include <cstdlib>
#include <type_traits>
#include <vector>



namespace sealhip
{
    namespace
    {

        template <bool FP, int D1, int ICLS = 0, bool PACK = false>
        __device__ __forceinline__ void fwd_p1_body(const FwdArgs &a, unsigned prime, unsigned comp, unsigned outer, uint64_t *lds, unsigned tile)
        {
            typedef Field<FP> F;
            typedef Geo<D1> G;
            const unsigned tid = threadIdx.x, cg = tile;
            const typename F::Mod m = F::make_mod(ld_uniform_mod(&a.t.mods[prime]), ld_uniform_fpd(&a.t.fpd[prime]));
            const typename F::tw_t *tab = tw_table<FP>(a.t, false, prime);
            SrcMap sm{ 0, 0, 0, 0 };
            const uint64_t *in0;
            size_t in_stride;
            if (a.src)
            {
                in0 = a.src + ((size_t)(comp % a.src_ncomp) << G::n);
                in_stride = a.src_outer_stride;
                sm.mode = a.src_mode;
                sm.half = a.src_half;
                sm.src_q = a.src_q;
                sm.fix = a.src_mode >= 2 ? a.src_fix[comp] : 0;
            }
            else
            {
                in0 = a.data + ((size_t)comp << G::n);
                in_stride = a.outer_stride;
            }
            const unsigned c = tid & (G::C - 1), rbl = tid >> G::LC;
            in0 += cg * G::C + c;
            TwRegs<FP> tw;
            p1_load_tw<FP, D1>(tw, tab, tid);
            uint64_t nxt[16];
            auto fetch = [&](unsigned z) {
                const uint64_t *in = in0 + (size_t)z * in_stride;
#pragma unroll
                for (int e = 0; e < 16; e++)
                {
                    const unsigned ra = e >> (4 - G::rA), rbh = e & ((1 << (4 - G::rA)) - 1);
                    const unsigned R = ra * 16 + (rbh << G::rA) + rbl;
#ifdef SEALHIP_P1_NOLOAD
                    nxt[e] = (uint64_t)(tid * 16 + e + z * 4099u + R) & 0xFFFFFFFFFFFFull; // measurement build: no loads (any word below 2^48 does)
#else
                    nxt[e] = a.src ? in[(size_t)R * 256] : mid_ld<16>(in + (size_t)R * 256); // (a mapped source is shared by the components)
#endif
                }
            };
            const unsigned ostride = gridDim.z;
            fetch(outer);
            for (; outer < a.nouter; outer += ostride)
            {
                typename F::elem x[16];
#pragma unroll
                for (int e = 0; e < 16; e++)
                    x[e] = map_src<FP>(nxt[e], sm, m);
                prio_phase<2>();
                if (outer + ostride < a.nouter)
                    fetch(outer + ostride);
                prio_phase<1>();
                uint64_t *mid_tr = a.mid + (((size_t)outer * a.ncomp + comp) << G::n);
                // SEALHIP_P1_PLAIN_LEAN=1 (measured, not kept: 2925 vs 2941 GB/s on the leg, profiles/r05_p1_bound.txt): double precision,
                // eight stages - every source mapping hands over |x| <= q (+ 2^32) <= kLeanEntry q, so ONE fix() after stage 6 would do
                // (1.0 -> 1.69 -> 2.50 -> 3.47 -> 4.63 -> 5.99 -> 7.61 < 8, fix, -> 1.09 -> 1.80; the intermediate then leaves at 1.80 q,
                // 52 bits when packed, which pass 2's first phase takes): 48 of the ~700 vector instructions per wave and tile less, and
                // no time - what the pass waits for without memory traffic is its LDS exchange and barriers, not its issue slots
#ifndef SEALHIP_P1_PLAIN_LEAN
#define SEALHIP_P1_PLAIN_LEAN 0
#endif
                p1_tile<FP, D1, 256, SEALHIP_P1_PLAIN_LEAN && FP && G::rA == 4, ICLS, PACK>(x, m, tab, tw, lds, mid_tr, cg, tid);
            }
        }

        // CLS: 0 = every component of the launch uses the integer back end, 1 = the double-precision one,
        // 2 = decided per workgroup (costs the registers of both bodies)
        template <int D1, int CLS>
#ifndef SEALHIP_PACK_WAVES_P1
#define SEALHIP_PACK_WAVES_P1 3 // the packing pass needs ~150 VGPRs; at 128 it spills 64 bytes per lane and runs 6 % slower than the plain pass
#endif
        __global__ void __launch_bounds__(kThreads, CLS == 5 ? SEALHIP_PACK_WAVES_P1 : CLS == 1 ? SEALHIP_FP_WAVES_P1 : 2) ntt2_fwd_p1(FwdArgs a)
        {
            HIP_DYNAMIC_SHARED(uint64_t, lds)
            const Blk blk = spread_blocks();
            const unsigned comp = blk.y + a.comp0, outer = blk.z;
            const unsigned prime = SHL_UNIFORM(a.comp_prime ? a.comp_prime[comp] : a.prime_first + comp);
            if constexpr (CLS == 5) // as 1, storing the packed intermediate (kPackWords; N = 2^16 only)
                fwd_p1_body<true, D1, 0, D1 == 8>(a, prime, comp, outer, lds, blk.tile);
            else if constexpr (CLS == 1)
                fwd_p1_body<true, D1>(a, prime, comp, outer, lds, blk.tile);
            else if constexpr (CLS == 0)
                with_int_class(a.t, prime, [&](auto ic) { fwd_p1_body<false, D1, decltype(ic)::value>(a, prime, comp, outer, lds, blk.tile); });
            else if (a.t.fpd[prime].qi)
                fwd_p1_body<true, D1>(a, prime, comp, outer, lds, blk.tile);
            else
                fwd_p1_body<false, D1, 2>(a, prime, comp, outer, lds, blk.tile); // mixed launches keep the guarded butterflies
        }

        // HOIST (plain transforms of the double-precision back end, no epilogue): the tile's 30 twiddles stay in
        // registers for every outer item of the workgroup's loop - a tile's twiddles are as many bytes as its
        // coefficients.  Costs ~60 VGPRs: measured +4 % on the batched NTT, -3 % on the epilogue variants, hence
        // only here.
        // HOIST_LDS (CLS 4): the row-shared phase-A twiddles are staged once in LDS and only the 15 per-thread phase-B twiddles
        // stay in registers: the same "no twiddle is re-read per transform" at 128 VGPRs (four waves per SIMD) instead of 214 (two)
        template <bool FP, int D1, bool HOIST = false, bool HOIST_LDS = false, int ICLS = 0, bool PACK = false>
        __device__ __forceinline__ void fwd_p2_body(const FwdArgs &a, unsigned prime, unsigned comp, unsigned outer, uint64_t *lds, unsigned tile)
        {
            typedef Field<FP> F;
            typedef Geo<D1> G;
            const unsigned tid = threadIdx.x, hg = tile;
            const typename F::Mod m = F::make_mod(ld_uniform_mod(&a.t.mods[prime]), ld_uniform_fpd(&a.t.fpd[prime]));
            const typename F::tw_t *tab = tw_table<FP>(a.t, false, prime);
            const uint64_t *mid0 = a.mid + ((size_t)comp << G::n) + ((size_t)hg << 12) + mid_lane(tid);
            uint64_t *lds_wave = lds + (tid >> 6) * (4 * kRowWords);
            uint64_t nxt[16];
            auto fetch = [&](unsigned z) {
                const uint64_t *mp = mid0 + (((size_t)z * a.ncomp) << G::n);
#pragma unroll
                for (int e = 0; e < 16; e++)
                    nxt[e] = mid_ld<4>(mp + e * kMidRow);
            };
            // packed intermediate (kPackWords above): thread (e = tid >> 4, v = tid & 15) loads pack (cg = e, v) of this row tile
            constexpr bool packed = PACK;
            static_assert(!PACK || (FP && D1 == 8), "the packed intermediate is defined for the double-precision pass at N = 2^16");
            [[maybe_unused]] const uint64_t *pk0 = a.mid + ((size_t)comp << G::n) + (size_t)(hg * 16 + (tid >> 4)) * kPackBlock + (tid & 15);
            [[maybe_unused]] auto fetch_packed = [&](unsigned z) {
                const uint64_t *mp = pk0 + (((size_t)z * a.ncomp) << G::n);
#if SEALHIP_PACK_VEC16
                const uint64_t *m2 = mp + (tid & 15); // pair j of this thread's column at j*32 + v*2 (mp already holds + v)
#pragma unroll
                for (int j = 0; j < 6; j++)
                    mid_ld2<4>(m2 + j * 32, nxt[2 * j], nxt[2 * j + 1]);
                nxt[12] = mid_ld<4>(mp + 192);
#else
#pragma unroll
                for (int k = 0; k < 13; k++)
                    nxt[k] = mid_ld<4>(mp + k * 16);
#endif
            };
            const unsigned ostride = gridDim.z;
            TwRegs<FP> pre_a, pre_b;
            const typename F::tw_t *twa = nullptr;
            if constexpr (HOIST_LDS)
            {
                static_assert(FP, "double-precision back end only");
                double *la = reinterpret_cast<double *>(lds + kLds2Words);
                stage_twa<D1>(la, tab, hg, tid);
                twa = la;
                const unsigned h = hg * 16 + (tid >> 4), v = tid & 15;
                load_tw<FP, 4>(pre_b, tab, [&](int t) { return (1u << (D1 + 4 + t)) + ((h * 16 + v) << t); });
                __syncthreads();
            }
            else if constexpr (HOIST)
                p2_load_tw<FP, D1>(pre_a, pre_b, tab, hg, tid);
            if constexpr (packed)
                fetch_packed(outer);
            else
                fetch(outer);
            for (; outer < a.nouter; outer += ostride)
            {
            typename F::elem x[16];
            if constexpr (packed)
            {
                // decode the pack's sixteen rows and hand row r to thread (u = r, v) through the exchange buffer:
                // word (u, e, v) at u * kPackLdsRow + e * 16 + v (writes: a wave's 64 consecutive words; reads: conflict-free)
                uint64_t w[13];
#pragma unroll
                for (int k = 0; k < 13; k++)
                    w[k] = nxt[k];
                if (outer + ostride < a.nouter)
                    fetch_packed(outer + ostride);
                double y[16];
                unpack52(w, y);
                __syncthreads(); // the previous tile's transposes are done with the buffer
#pragma unroll
                for (int r = 0; r < 16; r++)
                    lds[r * kPackLdsRow + tid] = fp_to_bits(y[r]);
                __syncthreads();
#pragma unroll
                for (int e = 0; e < 16; e++)
                    x[e] = fp_from_bits(lds[(tid >> 4) * kPackLdsRow + e * 16 + (tid & 15)]);
                __syncthreads(); // before any wave's exchange writes land in it
            }
            else
            {
#pragma unroll
            for (int e = 0; e < 16; e++)
                x[e] = F::unraw(nxt[e]);
            prio_phase<2>();
            if (outer + ostride < a.nouter)
                fetch(outer + ostride);
            prio_phase<1>();
            }
            if constexpr (HOIST_LDS)
                p2_tile<FP, D1, false, false, true, true>(x, m, tab, twa, nullptr, lds_wave, hg, tid, &pre_a, &pre_b);
            else if constexpr (HOIST)
                p2_tile<FP, D1, false, false, true>(x, m, tab, nullptr, nullptr, lds_wave, hg, tid, &pre_a, &pre_b);
            else
                p2_tile<FP, D1, false, false, false, false, false, ICLS, kP1Out<ICLS, D1>>(x, m, tab, nullptr, nullptr, lds_wave, hg, tid);
            constexpr int BOUT = kP2Out<ICLS, D1>; // integer back end: bound of the results
            uint64_t val[16];
            const size_t row0 = ((size_t)comp << G::n) + ((size_t)(hg * 16 + (tid >> 6) * 4) << 8);
            if ((HOIST || HOIST_LDS) || a.epi == 0) // the hoisted variants are launched for plain transforms only
            {
#pragma unroll
                for (int e = 0; e < 16; e++)
                    val[e] = a.lazy ? fwd_out_lazy<FP, ICLS, BOUT>(x[e], m) : fwd_out_canon<FP, ICLS, BOUT>(x[e], m);
                store_rows(val, lds_wave, a.data + (size_t)outer * a.outer_stride + row0, tid);
            }
            else
            {
                // fused tail: the transform is consumed here and never stored
#pragma unroll
                for (int e = 0; e < 16; e++)
                    val[e] = fwd_out_lazy<FP, ICLS, BOUT>(x[e], m); // < 4q
                const uint64_t q = a.t.mods[prime].q;
                const ShoupOp mul = a.epi_mul[comp];
                const uint64_t *A = a.epi_a + (size_t)outer * a.epi_a_stride + row0;
                if (a.epi == 1)
                {
                    uint64_t *O = a.epi_out0 + (size_t)outer * a.epi_out_stride + row0;
                    emit_rows(val, lds_wave, tid, [&](unsigned off, uint64_t tv) { O[off] = mul_shoup(A[off] + 4 * q - tv, mul.w, mul.wq, q); });
                }
                else if (a.epi == 3)
                {
                    // A = c + S P^-1 already (KsFusedArgs::fold_c0): the ciphertext words are written, not updated
                    uint64_t *O = ((outer & 1) ? a.epi_out1 : a.epi_out0) + (size_t)(outer >> 1) * a.epi_out_stride + row0;
                    emit_rows(val, lds_wave, tid, [&](unsigned off, uint64_t tv) {
                        O[off] = sub_mod(mid_ld<16>(A + off), mul_shoup(tv, mul.w, mul.wq, q), q);
                    });
                }
                else
                {
                    uint64_t *O = ((outer & 1) ? a.epi_out1 : a.epi_out0) + (size_t)(outer >> 1) * a.epi_out_stride + row0;
                    emit_rows(val, lds_wave, tid, [&](unsigned off, uint64_t tv) {
                        O[off] = add_mod(O[off], mul_shoup(A[off] + 4 * q - tv, mul.w, mul.wq, q), q);
                    });
                }
            }
            }
        }

        template <int D1, int CLS>
        __global__ void __launch_bounds__(kThreads, (CLS == 1 || CLS == 4 || CLS == 5) ? SEALHIP_FP_WAVES_P2 : 2) ntt2_fwd_p2(FwdArgs a)
        {
            HIP_DYNAMIC_SHARED(uint64_t, lds)
            const Blk blk = spread_blocks();
            const unsigned comp = blk.y + a.comp0, outer = blk.z;
            const unsigned prime = SHL_UNIFORM(a.comp_prime ? a.comp_prime[comp] : a.prime_first + comp);
            if constexpr (CLS == 5) // as 4, reading the packed intermediate (N = 2^16 only)
                fwd_p2_body<true, D1, false, true, 0, D1 == 8>(a, prime, comp, outer, lds, blk.tile);
            else if constexpr (CLS == 4) // as 3 with the row-shared half of the twiddles in LDS: four waves per SIMD
                fwd_p2_body<true, D1, false, true>(a, prime, comp, outer, lds, blk.tile);
            else if constexpr (CLS == 3) // double-precision back end, plain transform, twiddles hoisted
                fwd_p2_body<true, D1, true>(a, prime, comp, outer, lds, blk.tile);
            else if constexpr (CLS == 1)
                fwd_p2_body<true, D1>(a, prime, comp, outer, lds, blk.tile);
            else if constexpr (CLS == 0)
                with_int_class(a.t, prime, [&](auto ic) { fwd_p2_body<false, D1, false, false, decltype(ic)::value>(a, prime, comp, outer, lds, blk.tile); });
            else if (a.t.fpd[prime].qi)
                fwd_p2_body<true, D1>(a, prime, comp, outer, lds, blk.tile);
            else
                fwd_p2_body<false, D1, false, false, 2>(a, prime, comp, outer, lds, blk.tile);
        }

        // ---------------------------------------------------------------------------------------
        // NttTail2 (ntt_kernels.h): two rounding divisions in one transform.  Separate kernels so that the plain passes above
        // keep their register budgets.  Pass 1 maps two sources into the target modulus, x = v P^-1 + u; pass 2's tail computes
        // out = (c + S P^-1 - NTT(x)) q_last^-1.
        // ---------------------------------------------------------------------------------------
        struct Tail2Args
        {
            FwdArgs f;
            NttTail2 x;
        };

        template <bool FP, int D1, int ICLS>
        __device__ __forceinline__ void tail2_p1_body(const Tail2Args &t2, unsigned prime, unsigned comp, unsigned outer, uint64_t *lds, unsigned tile)
        {
            typedef Field<FP> F;
            typedef Geo<D1> G;
            const FwdArgs &a = t2.f;
            const unsigned tid = threadIdx.x, cg = tile;
            const typename F::Mod m = F::make_mod(ld_uniform_mod(&a.t.mods[prime]), ld_uniform_fpd(&a.t.fpd[prime]));
            const typename F::tw_t *tab = tw_table<FP>(a.t, false, prime);
            const int mode = t2.x.halves_added ? 3 : 2;
            const SrcMap s1{ mode, a.src_half, a.src_q, a.src_fix[comp] }, s2{ mode, t2.x.src2_half, t2.x.src2_q, t2.x.src2_fix[comp] };
            const ShoupOp pm = t2.x.pmul[comp];
            const uint64_t q = a.t.mods[prime].q;
            const unsigned c = tid & (G::C - 1), rbl = tid >> G::LC;
            const size_t col = (size_t)cg * G::C + c;
            TwRegs<FP> tw;
            p1_load_tw<FP, D1>(tw, tab, tid);
            const unsigned ostride = gridDim.z;
            for (; outer < a.nouter; outer += ostride)
            {
                const uint64_t *in1 = a.src + (size_t)outer * a.src_outer_stride + col;
                const uint64_t *in2 = ((outer & 1) ? t2.x.src2_1 : t2.x.src2_0) + (size_t)(outer >> 1) * t2.x.src2_stride + col;
                uint64_t n1[16], n2[16];
#pragma unroll
                for (int e = 0; e < 16; e++)
                {
                    const unsigned ra = e >> (4 - G::rA), rbh = e & ((1 << (4 - G::rA)) - 1);
                    const unsigned R = ra * 16 + (rbh << G::rA) + rbl;
                    n1[e] = in1[(size_t)R * 256];
                    n2[e] = in2[(size_t)R * 256];
                }
                typename F::elem x[16];
                if constexpr (FP)
                {
                    // Round 6: one fix() per coefficient instead of three.  v = (word + rounding fix) is below 2 q + 2^32 unfixed, its
                    // product with the balanced P^-1 at most q (0.5 + 0.1875 * 2.1) < 0.9 q (field.h); u = (word + fix) is exact as a
                    // double whatever the ratio of the moduli - below 2^52 + q when the source modulus is (the usual case: one
                    // conversion instead of the split one), below q + 2^32 + q otherwise; the sum stays under 2^53 and is fixed once
                    const double pinv = pm.w > q / 2 ? -(double)(q - pm.w) : (double)pm.w; // balanced, exact below 2^50
                    const double f1 = fp_from_u52(s1.fix), f2 = fp_from_u52(s2.fix);
                    const bool u_small = !(s2.src_q >> 52);
#pragma unroll
                    for (int e = 0; e < 16; e++)
                    {
                        const uint64_t r1 = mode == 3 ? n1[e] : csub(n1[e] + s1.half, s1.src_q), r2 = mode == 3 ? n2[e] : csub(n2[e] + s2.half, s2.src_q);
                        const double v = F::from_any(r1, m) + f1, u = (u_small ? fp_from_u52(r2) : F::from_any(r2, m)) + f2;
                        x[e] = fp_mulmod(v, pinv, m.q, m.qinv) + u;
                        F::fix(x[e], m);
                    }
                }
                else
                {
#pragma unroll
                    for (int e = 0; e < 16; e++)
                    {
                        const typename F::elem v = map_src<FP>(n1[e], s1, m), u = map_src<FP>(n2[e], s2, m);
                        x[e] = mul_shoup(v, pm.w, pm.wq, q) + u; // below q + 2q: inside the forward input range [0, 4q)
                    }
                }
                uint64_t *mid_tr = a.mid + (((size_t)outer * a.ncomp + comp) << G::n);
                // double precision, eight stages: x is fixed (|x| <= q/2), so one fix() after stage 6 does (p1_tile, LEAN); the
                // intermediate leaves at 1.80 q, which pass 2's first phase takes (-> 6.21 q)
                p1_tile<FP, D1, 256, FP && G::rA == 4, ICLS>(x, m, tab, tw, lds, mid_tr, cg, tid);
            }
        }

        template <bool FP, int D1, int ICLS>
        __device__ __forceinline__ void tail2_p2_body(const Tail2Args &t2, unsigned prime, unsigned comp, unsigned outer, uint64_t *lds, unsigned tile)
        {
            typedef Field<FP> F;
            typedef Geo<D1> G;
            const FwdArgs &a = t2.f;
            const unsigned tid = threadIdx.x, hg = tile;
            const typename F::Mod m = F::make_mod(ld_uniform_mod(&a.t.mods[prime]), ld_uniform_fpd(&a.t.fpd[prime]));
            const typename F::tw_t *tab = tw_table<FP>(a.t, false, prime);
            const uint64_t *mid0 = a.mid + ((size_t)comp << G::n) + ((size_t)hg << 12) + mid_lane(tid);
            uint64_t *lds_wave = lds + (tid >> 6) * (4 * kRowWords);
            const uint64_t q = a.t.mods[prime].q;
            const ShoupOp mul = a.epi_mul[comp], pm = t2.x.pmul[comp];
            // the two constants as balanced doubles (exact: the primes of this back end are below 2^50)
            [[maybe_unused]] const double pm_fp = pm.w > q / 2 ? -(double)(q - pm.w) : (double)pm.w;
            [[maybe_unused]] const double mul_fp = mul.w > q / 2 ? -(double)(q - mul.w) : (double)mul.w;
            const size_t row0 = ((size_t)comp << G::n) + ((size_t)(hg * 16 + (tid >> 6) * 4) << 8);
            uint64_t nxt[16];
            auto fetch = [&](unsigned z) {
                const uint64_t *mp = mid0 + (((size_t)z * a.ncomp) << G::n);
#pragma unroll
                for (int e = 0; e < 16; e++)
                    nxt[e] = mid_ld<4>(mp + e * kMidRow);
            };
            const unsigned ostride = gridDim.z;
            fetch(outer);
            for (; outer < a.nouter; outer += ostride)
            {
                typename F::elem x[16];
#pragma unroll
                for (int e = 0; e < 16; e++)
                    x[e] = F::unraw(nxt[e]);
                if (outer + ostride < a.nouter)
                    fetch(outer + ostride);
                // the tail's two operands are requested before the transform, not at the stores that need them
                const uint64_t *A = a.epi_a + (size_t)outer * a.epi_a_stride + row0;
                const uint64_t *C = ((outer & 1) ? t2.x.c1 : t2.x.c0) + (size_t)(outer >> 1) * t2.x.c_stride + row0;
                const bool folded = t2.x.a_has_c; // A = c + S P^-1 already: one operand
                uint64_t av[16], cv[16];
#pragma unroll
                for (int k = 0; k < 16; k++)
                {
                    const unsigned off = (k >> 2) * 256 + (k & 3) * 64 + (tid & 63);
                    av[k] = mid_ld<16>(A + off);
                    cv[k] = folded ? 0 : mid_ld<16>(C + off);
                }
                p2_tile<FP, D1, false, false, false, false, false, ICLS, kP1Out<ICLS, D1>>(x, m, tab, nullptr, nullptr, lds_wave, hg, tid);
                uint64_t val[16];
                uint64_t *O = ((outer & 1) ? a.epi_out1 : a.epi_out0) + (size_t)(outer >> 1) * a.epi_out_stride + row0;
                if constexpr (FP)
                {
                    // double-precision primes: the tail stays in the field the transform worked in (round 3).  S and c are canonical
                    // (exact doubles), the constants balanced: |S P^-1 mod q| <= 0.69 q, + c < 1.69 q, - NTT(x) (|.| <= q/2) < 2.2 q,
                    // times q_last^-1 -> 0.91 q, fixed and made canonical: 27 vector instructions a coefficient where the
                    // 64-bit Shoup products of the integer form took 69 (llvm-objdump, round 3)
#pragma unroll
                    for (int e = 0; e < 16; e++)
                        val[e] = F::raw(x[e]);
                    emit_rows_k(val, lds_wave, tid, [&](int k, unsigned off, uint64_t tv) {
                        const double s = folded ? fp_from_u52(av[k]) : fp_mulmod(fp_from_u52(av[k]), pm_fp, m.q, m.qinv) + fp_from_u52(cv[k]);
                        const double r = fp_mulmod(s - fp_from_bits(tv), mul_fp, m.q, m.qinv);
                        O[off] = fp_to_canon(fp_fix(r, m.q, m.qinv), m);
                    });
                }
                else
                {
#pragma unroll
                    for (int e = 0; e < 16; e++)
                        val[e] = fwd_out_lazy<FP, ICLS, kP2Out<ICLS, D1>>(x[e], m); // < 4q
                    emit_rows_k(val, lds_wave, tid, [&](int k, unsigned off, uint64_t tv) {
                        const uint64_t s = folded ? av[k] : add_mod(mul_shoup(av[k], pm.w, pm.wq, q), cv[k], q); // c + S P^-1, canonical
                        O[off] = mul_shoup(s + 4 * q - tv, mul.w, mul.wq, q);
                    });
                }
            }
        }

        // CLS as ntt2_fwd_p1 / ntt2_fwd_p2
        template <int D1, int CLS>
        __global__ void __launch_bounds__(kThreads, 2) ntt2_tail2_p1(Tail2Args a)
        {
            HIP_DYNAMIC_SHARED(uint64_t, lds)
            const Blk blk = spread_blocks();
            const unsigned comp = blk.y + a.f.comp0, outer = blk.z;
            const unsigned prime = SHL_UNIFORM(a.f.prime_first + comp);
            if constexpr (CLS == 1)
                tail2_p1_body<true, D1, 0>(a, prime, comp, outer, lds, blk.tile);
            else if constexpr (CLS == 0)
                with_int_class(a.f.t, prime, [&](auto ic) { tail2_p1_body<false, D1, decltype(ic)::value>(a, prime, comp, outer, lds, blk.tile); });
            else if (a.f.t.fpd[prime].qi)
                tail2_p1_body<true, D1, 0>(a, prime, comp, outer, lds, blk.tile);
            else
                tail2_p1_body<false, D1, 2>(a, prime, comp, outer, lds, blk.tile);
        }
        template <int D1, int CLS>
        __global__ void __launch_bounds__(kThreads, 2) ntt2_tail2_p2(Tail2Args a)
        {
            HIP_DYNAMIC_SHARED(uint64_t, lds)
            const Blk blk = spread_blocks();
            const unsigned comp = blk.y + a.f.comp0, outer = blk.z;
            const unsigned prime = SHL_UNIFORM(a.f.prime_first + comp);
            if constexpr (CLS == 1)
                tail2_p2_body<true, D1, 0>(a, prime, comp, outer, lds, blk.tile);
            else if constexpr (CLS == 0)
                with_int_class(a.f.t, prime, [&](auto ic) { tail2_p2_body<false, D1, decltype(ic)::value>(a, prime, comp, outer, lds, blk.tile); });
            else if (a.f.t.fpd[prime].qi)
                tail2_p2_body<true, D1, 0>(a, prime, comp, outer, lds, blk.tile);
            else
                tail2_p2_body<false, D1, 2>(a, prime, comp, outer, lds, blk.tile);
        }

        // ---------------------------------------------------------------------------------------
        // generic inverse transform kernels (bit-reversed order -> natural order, scaled by N^-1):
        // pass A = the mirror of forward pass 2 (rows), pass B = the mirror of forward pass 1 (columns)
        // ---------------------------------------------------------------------------------------
        struct InvArgs
        {
            uint64_t *data;
            size_t outer_stride;
            const uint64_t *src; // natural-order input (= data when in place)
            size_t src_outer_stride;
            uint64_t *mid;
            const uint32_t *comp_prime;
            int cls_hint; // NttBatch::cls_hint (host side only: picks the kernels)
            unsigned prime_first;
            unsigned ncomp;
            unsigned comp0; // as FwdArgs
            unsigned nouter; // used by the single-launch kernels, whose workgroups loop over outer items
            int lazy;
            uint64_t out_add; // NttBatch::out_add
            // NttBatch::prod_x: the input is the 2 x 2 tensor product of two size-2 operands, formed while it is loaded (two-pass kernels)
            const uint64_t *prod_x, *prod_y;
            unsigned prod_batch, prod_outer0; // items per polynomial; outer index of this launch's first item
            uint64_t *prod_out;               // NttBatch::prod_out (null: the product is not stored)
            size_t prod_out_stride;
            uint32_t src_gal; // NttBatch::src_galois_elt (0: plain source)
            NttTables t;
        };

        // exponents (IntBounds) of the integer back end between the phases of an inverse transform with canonical input:
        // after pass A's two phases, after pass B's first phase
        template <int ICLS>
        constexpr int kInvE1 = inv_phase_exp<ICLS>(0, 4, 0);
        template <int ICLS>
        constexpr int kInvE2 = inv_phase_exp<ICLS>(kInvE1<ICLS>, 4, 0);
        template <int ICLS>
        constexpr int kInvE3 = inv_phase_exp<ICLS>(kInvE2<ICLS>, 4, 0);

        template <bool FP, int D1, int ICLS = 2>
        __device__ __forceinline__ void inv_pa_body(const InvArgs &a, unsigned prime, unsigned comp, unsigned outer, uint64_t *lds, unsigned tile)
        {
            typedef Field<FP> F;
            typedef Geo<D1> G;
            const unsigned tid = threadIdx.x, hg = tile;
            const unsigned v = tid & 15, u = tid >> 4, ul = u & 3;
            const unsigned h = hg * 16 + u;
            const typename F::Mod m = F::make_mod(ld_uniform_mod(&a.t.mods[prime]), ld_uniform_fpd(&a.t.fpd[prime]));
            const typename F::tw_t *tab = tw_table<FP>(a.t, true, prime);
            uint64_t *lds_wave = lds + (tid >> 6) * (4 * kRowWords);
            uint64_t raw[16];
            typename F::elem x[16];
            bool have_x = false; // the product branch of the double-precision back end leaves field elements, not words
            const size_t rows = ((size_t)comp << G::n) + ((size_t)(hg * 16 + (tid >> 6) * 4) << 8);
            if (a.prod_x)
            {
                // the dyadic ciphertext product (evaluator.cpp:497-541, sizes 2 x 2) formed here instead of being stored by one kernel
                // and read back by this one: polynomial p of the result is x0 y0, x0 y1 + x1 y0 or x1 y1 of item b
                const unsigned go = outer + a.prod_outer0, pp = go / a.prod_batch, b = go - pp * a.prod_batch;
                const size_t plane = (size_t)a.prod_batch * a.src_outer_stride, off = (size_t)b * a.src_outer_stride + rows;
                [[maybe_unused]] const ModDesc md = ld_uniform_mod(&a.t.mods[prime]);
                uint64_t rb[16];
                load_rows(raw, lds_wave, a.prod_x + (pp == 2 ? plane : 0) + off, tid);
                load_rows(rb, lds_wave, a.prod_y + (pp == 0 ? 0 : plane) + off, tid);
                if constexpr (FP)
                {
                    // round 6, double-precision primes: the products are formed in the field the transform works in (canonical words are exact
                    // doubles; a product of two of them is at most 0.875 q, field.h) - 10 vector instructions a product instead of the
                    // integer Barrett form's ~30, and no conversion back before the first butterflies
#pragma unroll
                    for (int e = 0; e < 16; e++)
                        x[e] = fp_mulmod(fp_from_u52(raw[e]), fp_from_u52(rb[e]), m.q, m.qinv);
                    if (pp == 1)
                    {
                        uint64_t rc[16];
                        load_rows(rc, lds_wave, a.prod_x + plane + off, tid);
                        load_rows(rb, lds_wave, a.prod_y + off, tid);
#pragma unroll
                        for (int e = 0; e < 16; e++)
                            x[e] += fp_mulmod(fp_from_u52(rc[e]), fp_from_u52(rb[e]), m.q, m.qinv);
                    }
#pragma unroll
                    for (int e = 0; e < 16; e++)
                        F::fix(x[e], m);
                    if (a.prod_out)
                    {
#pragma unroll
                        for (int e = 0; e < 16; e++)
                            raw[e] = fp_to_canon(x[e], m);
                        store_rows(raw, lds_wave, a.prod_out + (size_t)outer * a.prod_out_stride + rows, tid);
                    }
                    have_x = true;
                }
                else
                {
#pragma unroll
                    for (int e = 0; e < 16; e++)
                        raw[e] = mul_mod(raw[e], rb[e], md);
                    if (pp == 1)
                    {
                        uint64_t rc[16];
                        load_rows(rc, lds_wave, a.prod_x + plane + off, tid);
                        load_rows(rb, lds_wave, a.prod_y + off, tid);
#pragma unroll
                        for (int e = 0; e < 16; e++)
                            raw[e] = add_mod(raw[e], mul_mod(rc[e], rb[e], md), md.q);
                    }
                    if (a.prod_out)
                        store_rows(raw, lds_wave, a.prod_out + (size_t)outer * a.prod_out_stride + rows, tid);
                }
            }
            else if (a.src_gal)
                load_rows_galois(raw, lds_wave, a.src + (size_t)outer * a.src_outer_stride + ((size_t)comp << G::n), (hg * 16 + (tid >> 6) * 4) << 8, a.src_gal, G::n, tid);
            else
                load_rows(raw, lds_wave, a.src + (size_t)outer * a.src_outer_stride + rows, tid);
            if (!have_x)
            {
#pragma unroll
                for (int e = 0; e < 16; e++)
                {
                    x[e] = F::from_canon(raw[e], m);
                    F::fix(x[e], m);
                }
            }
            {
                TwRegs<FP> tw;
                load_tw<FP, 4>(tw, tab, [&](int t) { return (1u << (D1 + 4 + t)) + ((h * 16 + v) << t); });
                phase_inv<FP, 4, 0, ICLS, 0>(x, m, [&](int t, int g) { return tw.get((1 << t) + g); });
            }
#pragma unroll
            for (int e = 0; e < 16; e++)
                F::fix(x[e], m);
            // wave-local exchange: (v', e') -> (e, v)
#pragma unroll
            for (int e = 0; e < 16; e++)
                lds_wave[ul * kRowWords + v * 18 + e] = F::raw(x[e]);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int e = 0; e < 16; e++)
                x[e] = F::unraw(lds_wave[ul * kRowWords + e * 18 + v]);
            {
                TwRegs<FP> tw;
                load_tw<FP, 4>(tw, tab, [&](int t) { return (1u << (D1 + t)) + (h << t); });
                phase_inv<FP, 4, 0, ICLS, kInvE1<ICLS>>(x, m, [&](int t, int g) { return tw.get((1 << t) + g); });
            }
#pragma unroll
            for (int e = 0; e < 16; e++)
                F::fix(x[e], m);
            uint64_t *mid_tr = a.mid + (((size_t)outer * a.ncomp + comp) << G::n) + ((size_t)hg << 12);
#pragma unroll
            for (int e = 0; e < 16; e++)
                mid_st<8>(mid_tr + e * 256 + tid, F::raw(x[e]));
        }

        template <int D1, int CLS>
        __global__ void __launch_bounds__(kThreads) ntt2_inv_pa(InvArgs a)
        {
            HIP_DYNAMIC_SHARED(uint64_t, lds)
            const Blk blk = spread_blocks();
            const unsigned comp = blk.y + a.comp0, outer = blk.z;
            const unsigned prime = SHL_UNIFORM(a.comp_prime ? a.comp_prime[comp] : a.prime_first + comp);
            if constexpr (CLS == 1)
                inv_pa_body<true, D1>(a, prime, comp, outer, lds, blk.tile);
            else if constexpr (CLS == 0)
                with_int_class(a.t, prime, [&](auto ic) { inv_pa_body<false, D1, decltype(ic)::value>(a, prime, comp, outer, lds, blk.tile); });
            else if (a.t.fpd[prime].qi)
                inv_pa_body<true, D1>(a, prime, comp, outer, lds, blk.tile);
            else
                inv_pa_body<false, D1, 2>(a, prime, comp, outer, lds, blk.tile); // mixed launches keep the guarded butterflies
        }

        template <bool FP, int D1, int ICLS = 2>
        __device__ __forceinline__ void inv_pb_body(const InvArgs &a, unsigned prime, unsigned comp, unsigned outer, uint64_t *lds, unsigned tile)
        {
            typedef Field<FP> F;
            typedef Geo<D1> G;
            static_assert(G::rA >= 1, "the N^-1 stage is handled in phase A");
            const unsigned tid = threadIdx.x, cg = tile;
            const unsigned c = tid & (G::C - 1), hi = tid >> G::LC;
            const typename F::Mod m = F::make_mod(ld_uniform_mod(&a.t.mods[prime]), ld_uniform_fpd(&a.t.fpd[prime]));
            const typename F::tw_t *tab = tw_table<FP>(a.t, true, prime);
            const unsigned col = cg * G::C + c;
            const uint64_t *i = a.mid + (((size_t)outer * a.ncomp + comp) << G::n) + ((size_t)(hi * 16 + (col >> 4)) << 8) + (col & 15);
            typename F::elem x[16];
#pragma unroll
            for (int rb = 0; rb < 16; rb++)
                x[rb] = F::unraw(mid_ld<8>(i + rb * 16));
            {
                TwRegs<FP> tw;
                load_tw<FP, 4>(tw, tab, [&](int t) { return (1u << (G::rA + t)) + (hi << t); });
                phase_inv<FP, 4, 0, ICLS, kInvE2<ICLS>>(x, m, [&](int t, int g) { return tw.get((1 << t) + g); });
            }
#pragma unroll
            for (int e = 0; e < 16; e++)
                F::fix(x[e], m);
#pragma unroll
            for (int rb = 0; rb < 16; rb++)
                lds[(hi * 16 + rb) * G::CP + c] = F::raw(x[rb]);
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 16; e++)
            {
                const unsigned ra = e >> (4 - G::rA), rbh = e & ((1 << (4 - G::rA)) - 1);
                const unsigned R = ra * 16 + (rbh << G::rA) + hi;
                x[e] = F::unraw(lds[R * G::CP + c]);
            }
            // phase A undone: stages rA-1 .. 1 with uniform twiddles, then stage 0 with N^-1 folded in
            phase_inv<FP, G::rA, 1, ICLS, kInvE3<ICLS>>(x, m, [&](int t, int g) { return ld_uniform(tab, (1u << t) + g); });
            {
                typename F::tw_t ni, nw;
                if constexpr (FP)
                {
                    ni = ld_uniform(a.t.ninv_d, 2 * prime);
                    nw = ld_uniform(a.t.ninv_d, 2 * prime + 1);
                }
                else
                {
                    ni = ld_uniform(a.t.ninv, 2 * prime);
                    nw = ld_uniform(a.t.ninv, 2 * prime + 1);
                }
                inv_last_stage<FP, G::rA, ICLS, kInvE3<ICLS>>(x, m, ni, nw);
            }
            uint64_t *o = a.data + (size_t)outer * a.outer_stride + ((size_t)comp << G::n);
#pragma unroll
            for (int e = 0; e < 16; e++)
            {
                F::fix(x[e], m);
                const unsigned ra = e >> (4 - G::rA), rbh = e & ((1 << (4 - G::rA)) - 1);
                const unsigned R = ra * 16 + (rbh << G::rA) + hi;
                uint64_t v = a.lazy ? F::inv_to_lazy(x[e], m) : F::inv_to_canon(x[e], m);
                if (a.out_add)
                    v = csub(v + a.out_add, a.t.mods[prime].q);
                mid_st<16>(o + (size_t)R * 256 + col, v);
            }
        }

        template <int D1, int CLS>
        __global__ void __launch_bounds__(kThreads) ntt2_inv_pb(InvArgs a)
        {
            HIP_DYNAMIC_SHARED(uint64_t, lds)
            const Blk blk = spread_blocks();
            const unsigned comp = blk.y + a.comp0, outer = blk.z;
            const unsigned prime = SHL_UNIFORM(a.comp_prime ? a.comp_prime[comp] : a.prime_first + comp);
            if constexpr (CLS == 1)
                inv_pb_body<true, D1>(a, prime, comp, outer, lds, blk.tile);
            else if constexpr (CLS == 0)
                with_int_class(a.t, prime, [&](auto ic) { inv_pb_body<false, D1, decltype(ic)::value>(a, prime, comp, outer, lds, blk.tile); });
            else if (a.t.fpd[prime].qi)
                inv_pb_body<true, D1>(a, prime, comp, outer, lds, blk.tile);
            else
                inv_pb_body<false, D1, 2>(a, prime, comp, outer, lds, blk.tile);
        }

        // ---------------------------------------------------------------------------------------
        // Single-launch transforms, second generation: N = 2^13 (two teams of 256 threads) and N = 2^14
        // (four teams), double-precision back end.  Every coefficient crosses HBM once in each direction.
        // Against a first generation that kept the intermediate and the exchange buffers side by side (140 KiB, one workgroup per CU):
        //  * the tile-order intermediate, pass 1's exchange buffer and pass 2's wave-local exchange buffers
        //    are the SAME LDS area used one after the other (one more barrier per transform): 76 KiB per
        //    workgroup at N = 2^13, so two workgroups share a CU (four waves per SIMD) instead of one;
        //    152 KiB at N = 2^14 (one 1024-thread workgroup per CU, also four waves per SIMD);
        //  * for D1 <= 6 the row index of pass 1's phase B is wave-uniform (tid >> LC with LC >= 6), so ALL
        //    twiddles of pass 1 are scalar loads; pass 2's row-shared phase-A twiddles are staged once in LDS
        //    and its per-thread phase-B twiddles stay in registers for the whole loop: no twiddle is re-read
        //    per transform and the kernel fits 128 VGPRs;
        //  * the inverse transform gets the same treatment (ntt2_inv_fused2).
        // ---------------------------------------------------------------------------------------
        template <int D1, int TWW = 1> // TWW = 64-bit words per twiddle: 1 double-precision back end, 2 integer back end (Shoup pairs)
        struct FusedGeo
        {
            typedef Geo<D1> G;
            static_assert(G::LC >= 6, "pass 1's phase-B row index must be wave-uniform");
            static_assert(G::lds1_words <= kLds2Words, "pass 1's exchange buffer lives in the team's pass-2 area");
            static constexpr int TEAMS = G::TILES;
            static constexpr int BS = 272; // padded 256-word block of the LDS intermediate: the 16-lane runs of pass 1 fall on different banks
            static constexpr size_t mid_words = (size_t)TEAMS * 16 * BS;
            static constexpr size_t xch_words = (size_t)TEAMS * kLds2Words;
            static constexpr size_t main_words = mid_words > xch_words ? mid_words : xch_words;
            static constexpr size_t lds_bytes = (main_words + (size_t)TEAMS * 240 * TWW) * 8;
        };
        // Integer back end in one launch (round 3): the same structure with Shoup pairs, budgeted for 128 VGPRs like the
        // double-precision kernels - two workgroups per CU at N = 2^13 (or one of each class, which is what a chain with both
        // kinds of primes launches side by side: 76 + 77 KiB of LDS, 4 x 128 registers per SIMD lane) and the one 1024-thread
        // workgroup a CU holds at N = 2^14.  Measured by hipcc's resource remarks (tools/quick/resources.py):
        //  * TWB_REGS: pass 2's fifteen per-thread twiddles resident for the workgroup's loop are 60 VGPRs of Shoup pairs -> 183
        //    registers (two waves per SIMD) or 120-230 bytes of scratch at 128; they are re-read per transform instead (L2:
        //    the table of one prime is 16 N bytes), each where it is used;
        //  * PF: the next transform's sixteen words in flight during this one (32 VGPRs) fits the forward kernel at 2^14 only
        //    (124 registers); elsewhere it spills 28-76 bytes, and at 2^13 the second workgroup of the CU covers the load.
// forward: where pass 2's fifteen per-thread twiddles come from (0 fetched where used, 1 requested at the start of pass 2 of every
// transform, 2 resident for the workgroup's loop), prefetch of the next transform at 2^13 / 2^14, waves per SIMD asked of the compiler
#ifndef SEALHIP_FINT_FWD_TWB
#define SEALHIP_FINT_FWD_TWB 0
#endif
#ifndef SEALHIP_FINT_FWD_PF13
#define SEALHIP_FINT_FWD_PF13 0
#endif
#ifndef SEALHIP_FINT_FWD_PF14
#define SEALHIP_FINT_FWD_PF14 1
#endif
#ifndef SEALHIP_FINT_FWD_WAVES
#define SEALHIP_FINT_FWD_WAVES 4
#endif
// inverse: the first phase's per-thread twiddles fetched where used (0) or requested together with the rows (1); prefetch; waves
#ifndef SEALHIP_FINT_INV_TWB
#define SEALHIP_FINT_INV_TWB 1
#endif
#ifndef SEALHIP_FINT_INV_PF
#define SEALHIP_FINT_INV_PF 0
#endif
#ifndef SEALHIP_FINT_INV_WAVES
#define SEALHIP_FINT_INV_WAVES 4
#endif

        template <bool FP, int D1, int ICLS = 0>
        __device__ __forceinline__ void fwd_fused2_body(const FwdArgs &a, unsigned prime, unsigned comp, unsigned outer, uint64_t *lds)
        {
            typedef Field<FP> F;
            typedef Geo<D1> G;
            typedef FusedGeo<D1, F::tw_words> FG;
            typedef typename F::tw_t tw_t;
            constexpr bool TWB_REGS = FP || (D1 == 5 && SEALHIP_FINT_FWD_TWB == 2); // 2^14: one 1024-thread workgroup, 128 registers
            constexpr bool TWB_EARLY = !FP && D1 == 5 && SEALHIP_FINT_FWD_TWB == 1;
#ifndef SEALHIP_FP_FWD_PF13
#define SEALHIP_FP_FWD_PF13 0 // as SEALHIP_FP_INV_PF: +2 % on the mixed chain, +5 % on a chain of primes below 2^50 (4.3 -> 4.5 TB/s)
#endif
            constexpr bool PF = FP ? (D1 != 5 || SEALHIP_FP_FWD_PF13) : ((D1 == 6 && SEALHIP_FINT_FWD_PF14) || (D1 == 5 && SEALHIP_FINT_FWD_PF13));
            const unsigned team = threadIdx.x >> 8, tid = threadIdx.x & 255;
            const typename F::Mod m = F::make_mod(ld_uniform_mod(&a.t.mods[prime]), ld_uniform_fpd(&a.t.fpd[prime]));
            const tw_t *tab = tw_table<FP>(a.t, false, prime);
            uint64_t *mid = lds;
            uint64_t *xch = lds + team * kLds2Words;
            uint64_t *lds_wave = xch + (tid >> 6) * (4 * kRowWords);
            tw_t *twa = reinterpret_cast<tw_t *>(lds + FG::main_words) + team * 240;
            stage_twa<D1>(twa, tab, team, tid); // read after several workgroup barriers
            const unsigned c = tid & (G::C - 1);
            const unsigned hi = SHL_UNIFORM(tid >> G::LC); // rbl in phase A, ra in phase B: the same in every lane of a wave
            TwRegs<FP> twb, unused;
            if constexpr (TWB_REGS)
            {
                const unsigned h = team * 16 + (tid >> 4), v = tid & 15;
                load_tw<FP, 4>(twb, tab, [&](int t) { return (1u << (D1 + 4 + t)) + ((h * 16 + v) << t); });
            }
            uint64_t *base = a.data + ((size_t)comp << G::n);
            // out of place (NttBatch::src with mode 0: the same residues, read from another buffer)
            const uint64_t *in_base = a.src ? a.src + ((size_t)(comp % a.src_ncomp) << G::n) : base;
            const size_t in_stride = a.src ? a.src_outer_stride : a.outer_stride;
            uint64_t nxt[16];
            auto fetch = [&](unsigned z) {
                const uint64_t *in = in_base + (size_t)z * in_stride + team * G::C + c;
#pragma unroll
                for (int e = 0; e < 16; e++)
                {
                    const unsigned ra = e >> (4 - G::rA), rbh = e & ((1 << (4 - G::rA)) - 1);
                    const unsigned R = ra * 16 + (rbh << G::rA) + hi;
                    nxt[e] = in[(size_t)R * 256];
                }
            };
            const unsigned ostride = gridDim.z;
            if constexpr (PF)
                fetch(outer);
            for (; outer < a.nouter; outer += ostride)
            {
                typename F::elem x[16];
                if constexpr (!PF)
                    fetch(outer);
#pragma unroll
                for (int e = 0; e < 16; e++)
                    x[e] = F::from_canon(nxt[e], m);
                if constexpr (PF)
                {
                    if (outer + ostride < a.nouter)
                        fetch(outer + ostride);
                }
                // ---- pass 1 on column tile `team`: every twiddle is wave-uniform (scalar loads)
                phase_fwd_end<FP, G::rA, true, ICLS, 4>(x, m, [&](int t, int g) { return ld_uniform(tab, (1u << t) + g); });
                __syncthreads(); // the previous transform's wave-local buffers are free
#pragma unroll
                for (int e = 0; e < 16; e++)
                {
                    const unsigned ra = e >> (4 - G::rA), rbh = e & ((1 << (4 - G::rA)) - 1);
                    const unsigned R = ra * 16 + (rbh << G::rA) + hi;
                    xch[R * G::CP + c] = F::raw(x[e]);
                }
                __syncthreads();
#pragma unroll
                for (int rb = 0; rb < 16; rb++)
                    x[rb] = F::unraw(xch[(hi * 16 + rb) * G::CP + c]);
                phase_fwd_end<FP, 4, true, ICLS, IntBounds<ICLS>::fwd_after(4, G::rA)>(
                    x, m, [&](int t, int g) { return ld_uniform(tab, (1u << (G::rA + t)) + (hi << t) + g); });
                __syncthreads(); // every team has read its exchange data: the area becomes the intermediate
                {
                    const unsigned col = team * G::C + c;
                    uint64_t *o = mid + (size_t)(hi * 16 + (col >> 4)) * FG::BS + (col & 15);
#pragma unroll
                    for (int rb = 0; rb < 16; rb++)
                        o[rb * 16] = F::raw(x[rb]);
                }
                __syncthreads();
                // ---- pass 2 on row tile `team`
                {
                    const uint64_t *mp = mid + (size_t)(team * 16) * FG::BS + tid;
#pragma unroll
                    for (int e = 0; e < 16; e++)
                        x[e] = F::unraw(mp[e * FG::BS]);
                }
                if constexpr (TWB_EARLY)
                {
                    // requested before phase A so that they arrive during it (fetched where they are used, every stage of phase B
                    // waits for a round trip to L2)
                    const unsigned h = team * 16 + (tid >> 4), v = tid & 15;
                    load_tw<FP, 4>(twb, tab, [&](int t) { return (1u << (D1 + 4 + t)) + ((h * 16 + v) << t); });
                }
                __syncthreads(); // the intermediate is consumed: the area becomes the wave-local exchange buffers
                if constexpr (TWB_REGS || TWB_EARLY)
                    p2_tile<FP, D1, false, false, true, true, false, ICLS, kP1Out<ICLS, D1>>(x, m, tab, twa, nullptr, lds_wave, team, tid, &unused, &twb);
                else
                    p2_tile<FP, D1, false, true, false, true, false, ICLS, kP1Out<ICLS, D1>>(x, m, tab, twa, nullptr, lds_wave, team, tid);
                uint64_t val[16];
#pragma unroll
                for (int e = 0; e < 16; e++)
                    val[e] = a.lazy ? fwd_out_lazy<FP, ICLS, kP2Out<ICLS, D1>>(x[e], m) : fwd_out_canon<FP, ICLS, kP2Out<ICLS, D1>>(x[e], m);
                store_rows(val, lds_wave, base + (size_t)outer * a.outer_stride + ((size_t)(team * 16 + (tid >> 6) * 4) << 8), tid);
            }
        }

        // CLS: 1 double-precision back end, 0 integer back end (one modulus class per workgroup)
        template <int D1, int CLS>
        __global__ void __launch_bounds__(FusedGeo<D1>::TEAMS *kThreads, CLS == 1 ? 4 : SEALHIP_FINT_FWD_WAVES) ntt2_fwd_fused2(FwdArgs a)
        {
            HIP_DYNAMIC_SHARED(uint64_t, lds)
            const unsigned comp = blockIdx.y + a.comp0, outer = blockIdx.z;
            const unsigned prime = SHL_UNIFORM(a.comp_prime ? a.comp_prime[comp] : a.prime_first + comp);
            if constexpr (CLS == 1)
                fwd_fused2_body<true, D1>(a, prime, comp, outer, lds);
            else
                with_int_class(a.t, prime, [&](auto ic) { fwd_fused2_body<false, D1, decltype(ic)::value>(a, prime, comp, outer, lds); });
        }

        // Inverse: team k undoes pass 2 on row tile k (rows -> intermediate in LDS), the teams meet, team k undoes
        // pass 1 on column tile k (N^-1 folded into the last stage) and stores natural order.
        template <bool FP, int D1, int ICLS = 2>
        __device__ __forceinline__ void inv_fused2_body(const InvArgs &a, unsigned prime, unsigned comp, unsigned outer, uint64_t *lds)
        {
            typedef Field<FP> F;
            typedef Geo<D1> G;
            typedef FusedGeo<D1, F::tw_words> FG;
            typedef typename F::tw_t tw_t;
            static_assert(G::rA >= 1, "the N^-1 stage is handled in phase A");
#ifndef SEALHIP_FP_INV_PF
#define SEALHIP_FP_INV_PF 0 // 2^13, double precision: two workgroups per CU cover the loads; without the 32 prefetch registers nothing spills: +10 %
#endif
            constexpr bool PF = FP ? (SEALHIP_FP_INV_PF || D1 != 5) : (D1 == 5 && SEALHIP_FINT_INV_PF);
            const unsigned team = threadIdx.x >> 8, tid = threadIdx.x & 255;
            const unsigned v = tid & 15, u = tid >> 4, ul = u & 3, lane = tid & 63;
            const unsigned h = team * 16 + u;
            const typename F::Mod m = F::make_mod(ld_uniform_mod(&a.t.mods[prime]), ld_uniform_fpd(&a.t.fpd[prime]));
            const tw_t *tab = tw_table<FP>(a.t, true, prime);
            uint64_t *mid = lds;
            uint64_t *xch = lds + team * kLds2Words;
            uint64_t *lds_wave = xch + (tid >> 6) * (4 * kRowWords);
            tw_t *twa = reinterpret_cast<tw_t *>(lds + FG::main_words) + team * 240;
            stage_twa<D1>(twa, tab, team, tid);
            const unsigned c = tid & (G::C - 1);
            const unsigned hi = SHL_UNIFORM(tid >> G::LC);
            const unsigned col = team * G::C + c;
            tw_t ni, nw;
            if constexpr (FP)
            {
                ni = ld_uniform(a.t.ninv_d, 2 * prime);
                nw = ld_uniform(a.t.ninv_d, 2 * prime + 1);
            }
            else
            {
                ni = ld_uniform(a.t.ninv, 2 * prime);
                nw = ld_uniform(a.t.ninv, 2 * prime + 1);
            }
            const size_t comp_off = (size_t)comp << G::n;
            const size_t row_off = comp_off + ((size_t)(team * 16 + (tid >> 6) * 4) << 8);
            uint64_t nxt[16];
            auto fetch = [&](unsigned z) {
                const uint64_t *rows = a.src + (size_t)z * a.src_outer_stride + row_off;
#pragma unroll
                for (int k = 0; k < 16; k++)
                    nxt[k] = rows[(k >> 2) * 256 + (k & 3) * 64 + lane];
            };
            const unsigned ostride = gridDim.z;
            if constexpr (PF)
                fetch(outer);
            __syncthreads(); // twa staged
            for (; outer < a.nouter; outer += ostride)
            {
                if constexpr (!PF)
                    fetch(outer);
                TwRegs<FP> twb_early;
                if constexpr (!FP && SEALHIP_FINT_INV_TWB == 1) // the first phase's per-thread twiddles travel with the rows
                    load_tw<FP, 4>(twb_early, tab, [&](int t) { return (1u << (D1 + 4 + t)) + ((h * 16 + v) << t); });
                // ---- rows of tile `team`: coalesced words -> wave-local transposition -> (row u, cols 16 v + e)
#pragma unroll
                for (int k = 0; k < 16; k++)
                {
                    const unsigned row = k >> 2, cc = (k & 3) * 64 + lane;
                    lds_wave[row * kRowWords + cc + 2 * (cc >> 4)] = nxt[k];
                }
                __builtin_amdgcn_wave_barrier();
                typename F::elem x[16];
#pragma unroll
                for (int e = 0; e < 16; e++)
                {
                    x[e] = F::from_canon(lds_wave[ul * kRowWords + v * 18 + e], m);
                    F::fix(x[e], m);
                }
                __builtin_amdgcn_wave_barrier();
                {
                    // the 15 per-thread twiddles of this phase are re-read (L2) per transform: holding them next to the
                    // prefetched rows does not fit 128 VGPRs (measured: 40 spilled registers)
                    if constexpr (FP)
                    {
                        TwRegs<FP> twb;
                        load_tw<FP, 4>(twb, tab, [&](int t) { return (1u << (D1 + 4 + t)) + ((h * 16 + v) << t); });
                        phase_inv<FP, 4, 0, ICLS, 0>(x, m, [&](int t, int g) { return twb.get((1 << t) + g); });
                    }
                    else if constexpr (SEALHIP_FINT_INV_TWB == 1)
                        phase_inv<FP, 4, 0, ICLS, 0>(x, m, [&](int t, int g) { return twb_early.get((1 << t) + g); });
                    else // each twiddle fetched where it is used
                        phase_inv<FP, 4, 0, ICLS, 0>(x, m, [&](int t, int g) { return tab[(1u << (D1 + 4 + t)) + ((h * 16 + v) << t) + g]; });
                }
#pragma unroll
                for (int e = 0; e < 16; e++)
                    F::fix(x[e], m);
                if constexpr (PF)
                {
                    if (outer + ostride < a.nouter)
                        fetch(outer + ostride);
                }
                // wave-local exchange: (v', e') -> (e, v)
#pragma unroll
                for (int e = 0; e < 16; e++)
                    lds_wave[ul * kRowWords + v * 18 + e] = F::raw(x[e]);
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int e = 0; e < 16; e++)
                    x[e] = F::unraw(lds_wave[ul * kRowWords + e * 18 + v]);
                phase_inv<FP, 4, 0, ICLS, kInvE1<ICLS>>(x, m, [&](int t, int g) { return twa[(16u << t) - 16u + (u << t) + g]; });
#pragma unroll
                for (int e = 0; e < 16; e++)
                    F::fix(x[e], m);
                __syncthreads(); // every wave is done with its exchange buffer: the area becomes the intermediate
#pragma unroll
                for (int e = 0; e < 16; e++)
                    mid[(size_t)(team * 16 + e) * FG::BS + tid] = F::raw(x[e]);
                __syncthreads();
                // ---- columns of tile `team`
                {
                    const uint64_t *i = mid + (size_t)(hi * 16 + (col >> 4)) * FG::BS + (col & 15);
#pragma unroll
                    for (int rb = 0; rb < 16; rb++)
                        x[rb] = F::unraw(i[rb * 16]);
                }
                phase_inv<FP, 4, 0, ICLS, kInvE2<ICLS>>(x, m, [&](int t, int g) { return ld_uniform(tab, (1u << (G::rA + t)) + (hi << t) + g); });
#pragma unroll
                for (int e = 0; e < 16; e++)
                    F::fix(x[e], m);
                __syncthreads(); // the intermediate is consumed: the area becomes pass 1's exchange buffer
#pragma unroll
                for (int rb = 0; rb < 16; rb++)
                    xch[(hi * 16 + rb) * G::CP + c] = F::raw(x[rb]);
                __syncthreads();
#pragma unroll
                for (int e = 0; e < 16; e++)
                {
                    const unsigned ra = e >> (4 - G::rA), rbh = e & ((1 << (4 - G::rA)) - 1);
                    const unsigned R = ra * 16 + (rbh << G::rA) + hi;
                    x[e] = F::unraw(xch[R * G::CP + c]);
                }
                __syncthreads(); // (the next transform's row loads reuse the area)
                phase_inv<FP, G::rA, 1, ICLS, kInvE3<ICLS>>(x, m, [&](int t, int g) { return ld_uniform(tab, (1u << t) + g); });
                inv_last_stage<FP, G::rA, ICLS, kInvE3<ICLS>>(x, m, ni, nw);
                uint64_t *o = a.data + (size_t)outer * a.outer_stride + comp_off;
#pragma unroll
                for (int e = 0; e < 16; e++)
                {
                    F::fix(x[e], m);
                    const unsigned ra = e >> (4 - G::rA), rbh = e & ((1 << (4 - G::rA)) - 1);
                    const unsigned R = ra * 16 + (rbh << G::rA) + hi;
                    uint64_t v = a.lazy ? F::inv_to_lazy(x[e], m) : F::inv_to_canon(x[e], m);
                    if (a.out_add)
                        v = csub(v + a.out_add, a.t.mods[prime].q);
                    o[(size_t)R * 256 + col] = v;
                }
            }
        }

        template <int D1, int CLS>
        __global__ void __launch_bounds__(FusedGeo<D1>::TEAMS *kThreads, CLS == 1 ? 4 : SEALHIP_FINT_INV_WAVES) ntt2_inv_fused2(InvArgs a)
        {
            HIP_DYNAMIC_SHARED(uint64_t, lds)
            const unsigned comp = blockIdx.y + a.comp0, outer = blockIdx.z;
            const unsigned prime = SHL_UNIFORM(a.comp_prime ? a.comp_prime[comp] : a.prime_first + comp);
            if constexpr (CLS == 1)
                inv_fused2_body<true, D1>(a, prime, comp, outer, lds);
            else
                with_int_class(a.t, prime, [&](auto ic) { inv_fused2_body<false, D1, decltype(ic)::value>(a, prime, comp, outer, lds); });
        }

        // The key-switch kernels of N = 2^16 (eight stages per pass) use the lean fix() placement of p1_tile / p2_tile (tile-order
        // intermediate only; the lane-order geometry keeps its own).  SEALHIP_KS_LEAN_OFF at build time restores five fix() per pair.
#ifdef SEALHIP_KS_LEAN_OFF
        template <int D1>
        constexpr bool kLeanKs = false;
#else
        template <int D1>
        constexpr bool kLeanKs = D1 == 8;
#endif
        // the lean placement takes an unfixed digit of magnitude up to kLeanEntry q_I (p1_tile)
        constexpr double kLeanEntry = 1.13;

        // ---------------------------------------------------------------------------------------
        // fused key switching, pass 1: one workgroup = (column tile cg, digit J, batch item b);
        // loops over the target moduli in `targets` (entries: slot I in the K+1 x K grid, pool prime).
        // ---------------------------------------------------------------------------------------
        struct Ks1Args
        {
            const uint64_t *t;   // [batch][K][N] coefficient form, canonical mod q_J
            uint64_t *mid;       // [batch][K+1][K][N] tile order, raw field elements
            const uint32_t *targets; // [ntargets] pairs (I, prime)
            unsigned ntargets;
            unsigned K;
            unsigned batch;
            unsigned j0, j1;     // digits handled by this call (digit-parallel key switching)
            unsigned parts;      // in-launch digit groups: workgroup group g handles the g-th slice of [j0, j1) (small batches)
            int skip_diag;       // CKKS: (I == J) is the input itself, not transformed
            NttTables tb;
        };

        // a digit's words (residues modulo data prime J, src_q = q_J) as field elements of the target modulus, ready for p1_tile
        template <bool FP, int D1>
        __device__ __forceinline__ void ks1_map(const uint64_t (&nxt)[16], typename Field<FP>::elem (&x)[16], uint64_t src_q, const typename Field<FP>::Mod &m)
        {
            typedef Field<FP> F;
            if constexpr (FP)
            {
                if (src_q >> 52)
                {
#pragma unroll
                    for (int e = 0; e < 16; e++)
                    {
                        x[e] = F::from_any(nxt[e], m); // magnitude < q + 2^32
                        if constexpr (kLeanKs<D1>)
                            F::fix(x[e], m); // the lean placement starts from |x| <= q/2
                    }
                }
                else
                {
                    // digit below 2^52: exact as a double; one fix() brings it under q_I / 2 - which the lean placement does
                    // not need when q_J <= 1.13 q_I (p1_tile: the first fix() sits after stage 6, 2.804 B + 4.811 < 8)
#pragma unroll
                    for (int e = 0; e < 16; e++)
                        x[e] = fp_from_u52(nxt[e]);
                    if (!kLeanKs<D1> || (double)src_q > kLeanEntry * m.q)
                    {
#pragma unroll
                        for (int e = 0; e < 16; e++)
                            F::fix(x[e], m);
                    }
                }
            }
            else
            {
                if (src_q < 4 * m.q)
                {
                    // already inside the butterflies' lazy input range [0, 4 q_I)
#pragma unroll
                    for (int e = 0; e < 16; e++)
                        x[e] = nxt[e];
                }
                else
                {
#pragma unroll
                    for (int e = 0; e < 16; e++)
                        x[e] = F::from_any(nxt[e], m);
                }
            }
        }

        template <bool FP, int D1, int ICLS = 0>
        __device__ __forceinline__ void ks1_body(const Ks1Args &a, uint64_t *lds, unsigned I, unsigned prime, unsigned b, unsigned cg, unsigned j0, unsigned j1)
        {
            typedef Field<FP> F;
            typedef Geo<D1> G;
            const unsigned tid = threadIdx.x;
            const typename F::Mod m = F::make_mod(ld_uniform_mod(&a.tb.mods[prime]), ld_uniform_fpd(&a.tb.fpd[prime]));
            const typename F::tw_t *tab = tw_table<FP>(a.tb, false, prime);
            TwRegs<FP> tw;
            p1_load_tw<FP, D1>(tw, tab, tid);
            const unsigned c = tid & (G::C - 1), rbl = tid >> G::LC;
            // thread's 16 source words of digit J: rows R(e), column cg*C + c
            const uint64_t *in0 = a.t + (((size_t)b * a.K) << G::n) + cg * G::C + c;
            uint64_t nxt[16];
            auto fetch = [&](unsigned J) {
                const uint64_t *in = in0 + ((size_t)J << G::n);
#pragma unroll
                for (int e = 0; e < 16; e++)
                {
                    const unsigned ra = e >> (4 - G::rA), rbh = e & ((1 << (4 - G::rA)) - 1);
                    const unsigned R = ra * 16 + (rbh << G::rA) + rbl;
#ifdef SEALHIP_KS_NOMEM
                    nxt[e] = (uint64_t)((tid << 4) + (unsigned)e + J * 4099u + R); // (any word below 2^50 does)
#else
                    nxt[e] = in[(size_t)R * 256];
#endif
                }
            };
            const unsigned Jskip = a.skip_diag ? I : ~0u;
            unsigned J = Jskip == j0 ? j0 + 1 : j0;
            if (J < j1)
                fetch(J);
            while (J < j1)
            {
                const uint64_t src_q = a.tb.mods[J].q; // digit J is a residue modulo data prime J
                typename F::elem x[16];
                ks1_map<FP, D1>(nxt, x, src_q, m);
                unsigned Jn = J + 1;
                if (Jn == Jskip)
                    Jn++;
                if (Jn < j1)
                    fetch(Jn);
                uint64_t *mid_tr = a.mid + ((((size_t)b * (a.K + 1) + I) * a.K + J) << G::n);
                p1_tile<FP, D1, 256, FP && kLeanKs<D1>, ICLS>(x, m, tab, tw, lds, mid_tr, cg, tid);
                J = Jn;
            }
        }

        // One launch per arithmetic back end: the double-precision body needs ~127 VGPRs (4 waves per
        // SIMD), the integer body ~214 (2 waves); a merged kernel would run both at the lower occupancy.
        template <bool FP, int D1>
        __global__ void __launch_bounds__(kThreads) ks1_kernel(Ks1Args a)
        {
            typedef Geo<D1> G;
            HIP_DYNAMIC_SHARED(uint64_t, lds)
            // one workgroup = (column tile cg, target modulus, batch item), looping over the digits J:
            // the modulus constants and all twiddles are loop-invariant, the next digit is prefetched.
            // Blocks that share (b, cg) - they read the same digit tiles - sit on one XCD (blockIdx % 8).
            const unsigned bid = blockIdx.x;
            const unsigned low = bid & 7, rest = bid >> 3;
            const unsigned it = rest % a.ntargets, grp = (rest / a.ntargets) * 8 + low; // grp = b*TILES + cg
            if (grp >= a.batch * a.parts * G::TILES)
                return;
            const unsigned vb = grp / G::TILES, cg = grp % G::TILES; // virtual batch item = (digit group, batch item)
            const unsigned b = vb % a.batch, dg = vb / a.batch;
            const unsigned jlen = (a.j1 - a.j0 + a.parts - 1) / a.parts;
            const unsigned j0 = a.j0 + dg * jlen, j1 = j0 + jlen < a.j1 ? j0 + jlen : a.j1;
            const unsigned I = SHL_UNIFORM(a.targets[2 * it]), prime = SHL_UNIFORM(a.targets[2 * it + 1]);
            if constexpr (FP)
                ks1_body<FP, D1>(a, lds, I, prime, b, cg, j0, j1);
            else
                with_int_class(a.tb, prime, [&](auto ic) { ks1_body<FP, D1, decltype(ic)::value>(a, lds, I, prime, b, cg, j0, j1); });
        }

        // The same pass with the loops exchanged (round 4): one workgroup = (column tile cg, DIGIT J, batch item), holding the digit's
        // tile in registers and looping over the target moduli of its class.  Every digit tile is then fetched by two workgroups (one
        // per arithmetic class) instead of sixteen - in the order above the sixteen targets of a tile meet in an XCD's L2, and that
        // traffic, as large as the intermediate itself, was what pass 1 waited for (tools/microbench/ks_flow: pass 1 without its digit
        // loads 1.5 instead of 2.35 ms per 64 items).  What is reloaded per target instead - modulus constants (scalar) and fifteen
        // per-thread twiddles with four distinct addresses per wave - is small.  Used when (tiles x digits x batch) fills the chip;
        // small batches keep the order above, whose in-launch digit groups make the workgroups they need.
        // SMALL (double-precision targets only): every digit of the launch is below 2^52 - the words are converted to doubles once and
        // the general mapping (from_any, whose component-independent half the compiler hoists out of the target loop: +68 VGPRs
        // for every wave of a kernel that contains it) is not compiled in: 4 waves per SIMD instead of 3.  The launcher cuts the
        // digit range into runs of one kind (C5: digit 0, a 60-bit prime, and digits 1..14).
        template <bool FP, int D1, bool SMALL = false>
        __global__ void __launch_bounds__(kThreads, 2) ks1t_kernel(Ks1Args a)
        {
            static_assert(FP || !SMALL, "SMALL is a property of the double-precision launch");
            typedef Field<FP> F;
            typedef Geo<D1> G;
            HIP_DYNAMIC_SHARED(uint64_t, lds)
            const unsigned tid = threadIdx.x;
            const unsigned ndig = a.j1 - a.j0;
            const unsigned grp = blockIdx.x; // (b * ndig + dj) * TILES + cg
            if (grp >= a.batch * ndig * G::TILES)
                return;
            const unsigned cg = grp % G::TILES, bj = grp / G::TILES;
            const unsigned J = a.j0 + bj % ndig, b = bj / ndig;
            const unsigned c = tid & (G::C - 1), rbl = tid >> G::LC;
            const uint64_t *in = a.t + (((size_t)b * a.K + J) << G::n) + cg * G::C + c;
            uint64_t raw[16];
#pragma unroll
            for (int e = 0; e < 16; e++)
            {
                const unsigned ra = e >> (4 - G::rA), rbh = e & ((1 << (4 - G::rA)) - 1);
                const unsigned R = ra * 16 + (rbh << G::rA) + rbl;
                raw[e] = in[(size_t)R * 256];
            }
            const uint64_t src_q = a.tb.mods[J].q; // digit J is a residue modulo data prime J
            // double-precision targets, digit below 2^52: the words are converted ONCE and kept as doubles (left to the compiler, the
            // loop-invariant conversion is hoisted next to the integers: 178 VGPRs, two waves per SIMD instead of four)
            const bool as_doubles = SMALL || (FP && !(src_q >> 52));
            if (as_doubles)
            {
#pragma unroll
                for (int e = 0; e < 16; e++)
                    raw[e] = fp_to_bits(fp_from_u52(raw[e]));
            }
#pragma nounroll
            for (unsigned it = 0; it < a.ntargets; it++)
            {
                const unsigned I = SHL_UNIFORM(a.targets[2 * it]), prime = SHL_UNIFORM(a.targets[2 * it + 1]);
                if (a.skip_diag && I == J)
                    continue;
                uint64_t *mid_tr = a.mid + ((((size_t)b * (a.K + 1) + I) * a.K + J) << G::n);
                auto one = [&](auto ic) {
                    constexpr int ICLS = decltype(ic)::value;
                    const typename F::Mod m = F::make_mod(ld_uniform_mod(&a.tb.mods[prime]), ld_uniform_fpd(&a.tb.fpd[prime]));
                    const typename F::tw_t *tab = tw_table<FP>(a.tb, false, prime);
                    TwRegs<FP> tw;
                    p1_load_tw<FP, D1>(tw, tab, tid);
                    typename F::elem x[16];
                    if constexpr (FP)
                    {
                        if (as_doubles)
                        {
#pragma unroll
                            for (int e = 0; e < 16; e++)
                                x[e] = fp_from_bits(raw[e]);
                            if (!kLeanKs<D1> || (double)src_q > kLeanEntry * m.q) // as ks1_map
                            {
#pragma unroll
                                for (int e = 0; e < 16; e++)
                                    F::fix(x[e], m);
                            }
                        }
                        else if constexpr (!SMALL)
                            ks1_map<FP, D1>(raw, x, src_q, m);
                    }
                    else
                        ks1_map<FP, D1>(raw, x, src_q, m);
                    p1_tile<FP, D1, 256, FP && kLeanKs<D1>, ICLS>(x, m, tab, tw, lds, mid_tr, cg, tid);
                };
                if constexpr (FP)
                    one(std::integral_constant<int, 0>());
                else
                    with_int_class(a.tb, prime, one);
            }
        }

        // ---------------------------------------------------------------------------------------
        // fused key switching, pass 2 + inner product with the key.
        // one workgroup = (target modulus I, row tile hg, batch item b); loops over the digits J.
        //   acc[b][k][I][natural order] = sum_J NTT_I(t_J mod q_I) * key[J][k][comp(I)]   (canonical)
        // key layout: [J][2][L][N] with every component in "register order"
        //   pos(hg, e', tid) = hg*4096 + e'*256 + tid   <->   natural hg*4096 + (tid>>4)*256 + (tid&15)*16 + e'
        // and stored as doubles for double-precision primes (KSwitchKeys::set_key).
        // ---------------------------------------------------------------------------------------
        struct Ks2Args
        {
            const uint64_t *mid;     // [batch][K+1][K][N]
            const uint64_t *target;  // [batch][K][N] NTT form (CKKS diagonal shortcut) or null
            const uint64_t *key;     // [digits][key_digit_words] register order (ntt2_kernels.h: key_to_register_order)
            size_t key_digit_words;  // words of one digit
            uint64_t *acc;           // [batch][2][K+1][N] natural order, canonical
            const uint32_t *targets; // [ntargets] quads (I, prime, key component, its offset inside a digit in units of N words)
            unsigned ntargets;
            unsigned K, L;
            unsigned batch;
            unsigned j0, j1, key_digit0; // digits handled by this call; first digit resident in `key`
            unsigned parts;              // in-launch digit groups (as Ks1Args); group g writes acc + g * batch*2*(K+1)*N
            // KsFusedArgs::fold_c0: components I < K leave as c_k + S_k P^-1 instead of S_k (fold_pm = P^-1 mod q_I); null = plain sums
            const uint64_t *fold_c0, *fold_c1;
            const ShoupOp *fold_pm;
            // KsFusedArgs::fold_x: the addend is the product of two size-2 operands, formed here (c0 = x0 y0, c1 = x0 y1 + x1 y0)
            const uint64_t *fold_x, *fold_y;
            size_t fold_plane;
            // KsFusedArgs::galois_elt: `target` (the diagonal terms) and fold_c0 are read through the NTT-domain automorphism
            uint32_t gal;
            NttTables tb;
        };

#ifndef SEALHIP_KS2_INT_TWB3
#define SEALHIP_KS2_INT_TWB3 1
#endif
        template <bool FP, int D1, int ICLS = 0>
        __device__ __forceinline__ void ks2_body(const Ks2Args &a, uint64_t *lds, unsigned I, unsigned prime, unsigned koff, unsigned b, unsigned hg,
                                                 unsigned j0, unsigned j1, uint64_t *acc_part)
        {
            typedef Field<FP> F;
            typedef Geo<D1> G;
            const unsigned tid = threadIdx.x;
            const typename F::Mod m = F::make_mod(ld_uniform_mod(&a.tb.mods[prime]), ld_uniform_fpd(&a.tb.fpd[prime]));
            const typename F::tw_t *tab = tw_table<FP>(a.tb, false, prime);

            uint64_t *lds_wave = lds + (tid >> 6) * (4 * kRowWords);
            const typename F::tw_t *twa = nullptr, *twb = nullptr;
            if constexpr (!FP)
            {
                // integer back end: phase A's 240 row-shared Shoup pairs (3.8 KiB) in LDS; phase B's stay in L2
                typename F::tw_t *la = reinterpret_cast<typename F::tw_t *>(lds + kLds2Words);
                if (tid < 240)
                {
                    const unsigned t = 31 - __builtin_clz(tid / 16 + 1);
                    const unsigned r = tid - ((16u << t) - 16u);
                    la[tid] = tab[(1u << (D1 + t)) + ((hg * 16) << t) + r];
                }
                twa = la;
#if SEALHIP_KS2_INT_TWB3
                // ... and the last stage's eight per-thread pairs (32 KiB: two workgroups per CU still fit): the same for every digit
                typename F::tw_t *lb = la + 240;
#pragma unroll
                for (int g = 0; g < 8; g++)
                    lb[g * 256 + tid] = tab[(1u << (D1 + 7)) + ((hg * 256 + tid) << 3) + g];
                twb = lb;
#endif
                __syncthreads();
            }
            if constexpr (FP)
            {
                // stage this tile's twiddles in LDS once: reused by all K digits
                typename F::tw_t *la = reinterpret_cast<typename F::tw_t *>(lds + kLds2Words);
                typename F::tw_t *lb = la + 240;
                if (tid < 240)
                {
                    // entry (t, r): t = floor(log2(tid/16 + 1)), r = tid - (16<<t) + 16
                    const unsigned t = 31 - __builtin_clz(tid / 16 + 1);
                    const unsigned r = tid - ((16u << t) - 16u);
                    la[tid] = tab[(1u << (D1 + t)) + ((hg * 16) << t) + r];
                }
#pragma unroll
                for (int t = 0; t < 4; t++)
#pragma unroll
                    for (int g = 0; g < (1 << t); g++)
                        lb[((256u << t) - 256u) + g * 256 + tid] = tab[(1u << (D1 + 4 + t)) + ((hg * 256 + tid) << t) + g];
                twa = la;
                twb = lb;
                __syncthreads();
            }

            // double precision: sums of doubles, fixed every seven terms.  Integer back end (round 3): the key is the precomputed
            // operand of a Shoup product (its quotient plane: key_to_register_order), so a term is x k mod q in [0, 4q) for any
            // 64-bit x and the sum stays a 64-bit word - 2 VGPRs instead of the 4 of a 128-bit sum, 12 instructions a term with the
            // addition riding in the remainder chain (field.h: mul_rem) - brought under 4 q every kAccRun terms.
            typedef typename std::conditional<FP, typename F::Acc, uint64_t>::type acc_t;
            constexpr unsigned kAccRun = ICLS == 2 ? 1 : IntBounds<ICLS>::lim / 4 - 1; // 4 (n + 1) q <= lim q
            acc_t acc0[16], acc1[16];
#pragma unroll
            for (int e = 0; e < 16; e++)
            {
                if constexpr (FP)
                {
                    acc0[e] = F::acc_zero();
                    acc1[e] = F::acc_zero();
                }
                else
                {
                    acc0[e] = 0;
                    acc1[e] = 0;
                }
            }
            const size_t N = (size_t)1 << G::n;
            // Every load of the digit loop goes through a wave-uniform window (modarith.h: UniformView): the tile's base is the
            // same in every lane, the lane adds tid words, row e sits e * 2 KiB (intermediate) or e * 4 KiB (key pairs) further.
            const uint64_t *mid0 = a.mid + ((((size_t)b * (a.K + 1) + I) * a.K) << G::n) + ((size_t)hg << 12);
            // diagonal digit (CKKS): NTT_J(INTT_J(target_J)) = target_J (evaluator.cpp:2682-2685); read the
            // thread's 16 contiguous coefficients of row 16 hg + u straight from the input polynomial
            const uint64_t *mid0_lane = mid0 + mid_lane(tid);
            // (round 6) a deferred product's diagonal terms are x1 y1, formed here from the operands: diag_prod = the product is not stored
            const bool diag_prod = a.fold_x && !a.target && I < a.K;
            const bool has_diag = (a.target || diag_prod) && I < a.K;
            const size_t diag_off = (((size_t)b * a.K + I) << G::n) + ((size_t)hg << 12);
            const UniformView diag_view = uniform_view(diag_prod ? a.fold_x + a.fold_plane + diag_off : has_diag ? a.target + diag_off : a.mid);
            uint64_t nxt[16]; // digit J+1 is in flight while digit J is transformed
            auto fetch = [&](unsigned J) {
                if (has_diag && J == I)
                {
                    if (a.gal)
                    {
                        // a rotation's target is pi(c1), never stored: the thread's sixteen words come through the index map
                        const uint64_t *poly = a.target + (((size_t)b * a.K + I) << G::n);
#pragma unroll
                        for (int e = 0; e < 16; e++)
                            nxt[e] = poly[galois_src_index((hg << 12) + tid * 16 + e, a.gal, G::n)];
                    }
                    else
                    {
#pragma unroll
                        for (int e = 0; e < 16; e += 2)
                            view_load128(diag_view, tid * 128, e * 8, nxt[e], nxt[e + 1]);
                    }
                }
                else if constexpr (FP)
                {
                    const UniformView mv = uniform_view(mid0 + ((size_t)J << G::n));
#pragma unroll
                    for (int e = 0; e < 16; e++)
                    {
#ifdef SEALHIP_KS_NOMEM
                        nxt[e] = fp_to_bits((double)(int)((tid * 16 + e + J * 4099u) & 0xFFFFF) - 524288.0);
#else
                        if constexpr ((SEALHIP_KS_NT & 2) != 0)
                            nxt[e] = view_load64_nt(mv, mid_lane(tid) * 8, e * (kMidRow * 8));
                        else
                            nxt[e] = view_load64(mv, mid_lane(tid) * 8, e * (kMidRow * 8));
#endif
                    }
                }
                else
                {
                    // (the integer back end's loop is bound by latency, not by its instruction count: plain loads measured
                    // 0.6 % faster at BFV configs[3])
                    const uint64_t *mp = mid0_lane + ((size_t)J << G::n);
#pragma unroll
                    for (int e = 0; e < 16; e++)
                    {
#ifdef SEALHIP_KS_NOMEM
                        nxt[e] = ((uint64_t)(tid * 16 + e + J * 4099u) * 0x9E3779B97F4A7C15ull) >> 6;
#else
                        nxt[e] = mid_ld<4>(mp + e * kMidRow);
#endif
                    }
                }
            };
            // Integer back end: with 128-bit sums there were no registers for the prefetch (round 2: the overflow spilled to
            // scratch gained nothing, one wave per SIMD lost 10 %); with 64-bit sums (round 3) digit J + 1 travels during digit J
#ifndef SEALHIP_KS2_INT_PF
#define SEALHIP_KS2_INT_PF 1
#endif
            constexpr bool PF = FP || SEALHIP_KS2_INT_PF;
            if constexpr (PF)
            {
                if (j0 < j1)
                    fetch(j0);
            }
            for (unsigned J = j0; J < j1; J++)
            {
                typename F::elem x[16];
                const bool is_diag = has_diag && J == I;
                if constexpr (!PF)
                    fetch(J);
                if (is_diag && diag_prod)
                {
                    // nxt holds x1's sixteen words of this thread; y1's are loaded here, once per tile (the product's round trip through
                    // HBM - stored by the inverse transform, read back here - is what this replaces)
                    const UniformView yv = uniform_view(a.fold_y + a.fold_plane + diag_off);
                    uint64_t yw[16];
#pragma unroll
                    for (int e = 0; e < 16; e += 2)
                        view_load128(yv, tid * 128, e * 8, yw[e], yw[e + 1]);
                    [[maybe_unused]] const ModDesc mdd = ld_uniform_mod(&a.tb.mods[prime]);
#pragma unroll
                    for (int e = 0; e < 16; e++)
                    {
                        if constexpr (FP)
                        {
                            x[e] = fp_mulmod(fp_from_u52(nxt[e]), fp_from_u52(yw[e]), m.q, m.qinv); // <= 0.875 q (field.h)
                            F::fix(x[e], m);
                        }
                        else
                            x[e] = F::from_canon(mul_mod(nxt[e], yw[e], mdd), m);
                    }
                }
                else if (is_diag)
                {
#pragma unroll
                    for (int e = 0; e < 16; e++)
                        x[e] = F::from_canon(nxt[e], m);
                }
                else
                {
#pragma unroll
                    for (int e = 0; e < 16; e++)
                        x[e] = F::unraw(nxt[e]);
                }
                // this component of digit J: N pairs of doubles (double-precision primes) or 2 x N (word, Shoup quotient) pairs
                const size_t kslab = (size_t)(J - a.key_digit0) * a.key_digit_words + (size_t)koff * N;
                typename F::key_t kr0[16], kr1[16];
                prio_phase<2>();
                if constexpr (FP)
                {
                    // the key words of this digit and the next digit travel while this digit is transformed
                    const UniformView kv = uniform_view(a.key + kslab + ((size_t)hg << 13));
#pragma unroll
                    for (int e = 0; e < 16; e++)
                    {
                        uint64_t w0, w1; // (first, second) key polynomial of this coefficient
#ifdef SEALHIP_KS_NOMEM
                        w0 = fp_to_bits((double)(int)((tid * 7 + e * 31 + J) & 0xFFFFF) - 500000.0), w1 = fp_to_bits((double)(int)((tid * 3 + e * 17 + J) & 0xFFFFF) - 400000.0);
#else
                        view_load128(kv, tid * 16, e * 4096, w0, w1);
#endif
                        kr0[e] = fp_from_bits(w0);
                        kr1[e] = fp_from_bits(w1);
                    }
                    if (J + 1 < j1)
                        fetch(J + 1);
                }
                else if constexpr (PF)
                {
                    if (J + 1 < j1)
                        fetch(J + 1);
                }
                prio_phase<1>();
                if (!is_diag)
                {
                    p2_tile<FP, D1, FP, true, false, !FP, FP && kLeanKs<D1>, ICLS, kP1Out<ICLS, D1>, !FP && SEALHIP_KS2_INT_TWB3>(x, m, tab, twa, twb, lds_wave, hg, tid);
                    // integer back end: the Shoup product takes any 64-bit x (the transform's results are below kP2Out q); the
                    // guarded class (moduli of 2^60 and above: never a key-switch target, kept for completeness) works in [0, 2q)
                }
                if constexpr (FP)
                {
#pragma unroll
                    for (int e = 0; e < 16; e++)
                    {
                        F::mac(acc0[e], x[e], kr0[e], m);
                        F::mac(acc1[e], x[e], kr1[e], m);
                    }
                }
                else
                {
                    // the word and its quotient are neighbours: one 16-byte load each (register order in units of pairs)
                    const ShoupOp *p0 = reinterpret_cast<const ShoupOp *>(a.key + kslab) + ((size_t)hg << 12) + tid;
                    const ShoupOp *p1 = p0 + N; // the second key polynomial's pairs follow the first's
#pragma unroll
                    for (int e = 0; e < 16; e++)
                    {
#ifdef SEALHIP_KS_NOMEM
                        const ShoupOp w0{ (uint64_t)(tid * 7 + e * 31 + J) * 0x9E3779B97F4A7C15ull >> 5, (uint64_t)(tid + e) * 0xD1B54A32D192ED03ull }, w1{ w0.wq >> 4, w0.w << 3 };
#else
                        const ShoupOp w0 = p0[e * 256], w1 = p1[e * 256];
#endif
                        if constexpr (ICLS == 2)
                        {
                            acc0[e] = F::guard(acc0[e] + F::mul_lazy(x[e], w0, m), m);
                            acc1[e] = F::guard(acc1[e] + F::mul_lazy(x[e], w1, m), m);
                        }
                        else
                        {
                            acc0[e] = F::mul_rem(x[e], w0, F::mul_hi_approx(x[e], w0), acc0[e], m);
                            acc1[e] = F::mul_rem(x[e], w1, F::mul_hi_approx(x[e], w1), acc1[e], m);
                        }
                    }
                }
                if constexpr (FP)
                {
                    if ((J - j0) % 7 == 6)
                    {
#pragma unroll
                        for (int e = 0; e < 16; e++)
                        {
                            F::acc_fix(acc0[e], m);
                            F::acc_fix(acc1[e], m);
                        }
                    }
                }
                else if constexpr (ICLS != 2)
                {
                    if ((J - j0) % kAccRun == kAccRun - 1)
                    {
#pragma unroll
                        for (int e = 0; e < 16; e++)
                        {
                            F::template fix4<IntBounds<ICLS>::hi32>(acc0[e], m);
                            F::template fix4<IntBounds<ICLS>::hi32>(acc1[e], m);
                        }
                    }
                }
            }
            uint64_t val[16];
            uint64_t *out = acc_part + ((((size_t)b * 2 + 0) * (a.K + 1) + I) << G::n) + ((size_t)(hg * 16 + (tid >> 6) * 4) << 8);
            auto sum_to_canon = [&](const acc_t &s) {
                if constexpr (FP)
                    return F::acc_to_canon(s, m);
                else if constexpr (ICLS == 2)
                    return csub(s, m.q);
                else
                    return F::template canon_any<IntBounds<ICLS>::hi32>(s, m);
            };
            if (a.fold_x && I < a.K)
            {
                // round 6: the ciphertext being relinearised is a product that was never stored (Evaluator::multiply deferred it): its two
                // leading polynomials are formed here from the four operand polynomials, where they are needed - once
                const size_t crow = ((((size_t)b * a.K + I) << G::n)) + ((size_t)(hg * 16 + (tid >> 6) * 4) << 8);
                const uint64_t *X0 = a.fold_x + crow, *X1 = X0 + a.fold_plane, *Y0 = a.fold_y + crow, *Y1 = Y0 + a.fold_plane;
                const ShoupOp pm = a.fold_pm[I];
                const uint64_t q = a.tb.mods[prime].q;
                [[maybe_unused]] const ModDesc md = ld_uniform_mod(&a.tb.mods[prime]);
                auto load16 = [&](const uint64_t *P, uint64_t (&v)[16]) {
#pragma unroll
                    for (int k = 0; k < 16; k++)
                        v[k] = mid_ld<16>(P + (k >> 2) * 256 + (k & 3) * 64 + (tid & 63));
                };
                uint64_t p0[16], p1[16];
                uint64_t *out1 = out + ((size_t)(a.K + 1) << G::n);
                if constexpr (FP)
                {
                    // double-precision primes: the epilogue stays in the field the sums were formed in (as the tail's, round 3).  The
                    // sum is fixed (|.| <= q/2) and multiplied by the balanced P^-1: <= 0.6 q; a product of two canonical residues is
                    // <= 0.875 q (field.h: fp_mulmod, multiplier in [0, q)); c1 adds two of them: at most 2.35 q before the last fix()
                    const double pmd = pm.w > q / 2 ? -(double)(q - pm.w) : (double)pm.w;
                    double cd[16];
                    load16(X0, p0);
                    load16(Y0, p1);
#pragma unroll
                    for (int k = 0; k < 16; k++)
                        cd[k] = fp_mulmod(fp_from_u52(p0[k]), fp_from_u52(p1[k]), m.q, m.qinv);
#pragma unroll
                    for (int e = 0; e < 16; e++)
                        val[e] = fp_to_bits(fp_mulmod(fp_fix(acc0[e], m.q, m.qinv), pmd, m.q, m.qinv));
                    emit_rows_k(val, lds_wave, tid, [&](int k, unsigned off, uint64_t sv) {
                        mid_st<16>(out + off, fp_to_canon(fp_fix(fp_from_bits(sv) + cd[k], m.q, m.qinv), m));
                    });
                    load16(Y1, p1);
#pragma unroll
                    for (int k = 0; k < 16; k++)
                        cd[k] = fp_mulmod(fp_from_u52(p0[k]), fp_from_u52(p1[k]), m.q, m.qinv);
                    load16(X1, p0);
                    load16(Y0, p1);
#pragma unroll
                    for (int k = 0; k < 16; k++)
                        cd[k] += fp_mulmod(fp_from_u52(p0[k]), fp_from_u52(p1[k]), m.q, m.qinv);
#pragma unroll
                    for (int e = 0; e < 16; e++)
                        val[e] = fp_to_bits(fp_mulmod(fp_fix(acc1[e], m.q, m.qinv), pmd, m.q, m.qinv));
                    emit_rows_k(val, lds_wave, tid, [&](int k, unsigned off, uint64_t sv) {
                        mid_st<16>(out1 + off, fp_to_canon(fp_fix(fp_from_bits(sv) + cd[k], m.q, m.qinv), m));
                    });
                    return;
                }
                uint64_t cv[16];
                load16(X0, p0);
                load16(Y0, p1);
#pragma unroll
                for (int k = 0; k < 16; k++)
                    cv[k] = mul_mod(p0[k], p1[k], md);
#pragma unroll
                for (int e = 0; e < 16; e++)
                    val[e] = sum_to_canon(acc0[e]);
                emit_rows_k(val, lds_wave, tid, [&](int k, unsigned off, uint64_t sv) {
                    mid_st<16>(out + off, add_mod(mul_shoup(sv, pm.w, pm.wq, q), cv[k], q));
                });
                load16(Y1, p1);
#pragma unroll
                for (int k = 0; k < 16; k++)
                    cv[k] = mul_mod(p0[k], p1[k], md);
                load16(X1, p0);
                load16(Y0, p1);
#pragma unroll
                for (int k = 0; k < 16; k++)
                    cv[k] = add_mod(cv[k], mul_mod(p0[k], p1[k], md), q);
#pragma unroll
                for (int e = 0; e < 16; e++)
                    val[e] = sum_to_canon(acc1[e]);
                emit_rows_k(val, lds_wave, tid, [&](int k, unsigned off, uint64_t sv) {
                    mid_st<16>(out1 + off, add_mod(mul_shoup(sv, pm.w, pm.wq, q), cv[k], q));
                });
                return;
            }
            if (a.fold_c0 && I < a.K)
            {
                // the key-switch tail's "c + S P^-1" happens here, where S is in registers: the tail then reads one operand, not two
                // (evaluator.cpp:2845-2863 computes ct += (S - NTT(v)) P^-1 = (c + S P^-1) - NTT(v) P^-1; exact residue arithmetic)
                const size_t crow = ((((size_t)b * a.K + I) << G::n)) + ((size_t)(hg * 16 + (tid >> 6) * 4) << 8);
                const uint64_t *C0 = a.fold_c0 + crow, *C1 = a.fold_c1 ? a.fold_c1 + crow : nullptr; // null: that addend is zero (rotations)
                const ShoupOp pm = a.fold_pm[I];
                const uint64_t q = a.tb.mods[prime].q;
                uint64_t cv0[16], cv1[16];
#pragma unroll
                for (int k = 0; k < 16; k++)
                {
                    const unsigned off = (k >> 2) * 256 + (k & 3) * 64 + (tid & 63);
                    if (a.gal) // the first addend is pi(c0) of a rotation, read through the index map
                        cv0[k] = mid_ld<16>(a.fold_c0 + (((size_t)b * a.K + I) << G::n) + galois_src_index(((hg * 16 + (tid >> 6) * 4) << 8) + off, a.gal, G::n));
                    else
                        cv0[k] = mid_ld<16>(C0 + off);
                    cv1[k] = C1 ? mid_ld<16>(C1 + off) : 0;
                }
                uint64_t *out1 = out + ((size_t)(a.K + 1) << G::n);
                if constexpr (FP)
                {
                    // (as above: S P^-1 in the field, <= 0.6 q, plus the canonical addend: <= 1.6 q before the last fix())
                    const double pmd = pm.w > q / 2 ? -(double)(q - pm.w) : (double)pm.w;
#pragma unroll
                    for (int e = 0; e < 16; e++)
                        val[e] = fp_to_bits(fp_mulmod(fp_fix(acc0[e], m.q, m.qinv), pmd, m.q, m.qinv));
                    emit_rows_k(val, lds_wave, tid, [&](int k, unsigned off, uint64_t sv) {
                        mid_st<16>(out + off, fp_to_canon(fp_fix(fp_from_bits(sv) + fp_from_u52(cv0[k]), m.q, m.qinv), m));
                    });
#pragma unroll
                    for (int e = 0; e < 16; e++)
                        val[e] = fp_to_bits(fp_mulmod(fp_fix(acc1[e], m.q, m.qinv), pmd, m.q, m.qinv));
                    emit_rows_k(val, lds_wave, tid, [&](int k, unsigned off, uint64_t sv) {
                        mid_st<16>(out1 + off, fp_to_canon(fp_fix(fp_from_bits(sv) + fp_from_u52(cv1[k]), m.q, m.qinv), m));
                    });
                    return;
                }
#pragma unroll
                for (int e = 0; e < 16; e++)
                    val[e] = sum_to_canon(acc0[e]);
                emit_rows_k(val, lds_wave, tid, [&](int k, unsigned off, uint64_t sv) {
                    mid_st<16>(out + off, add_mod(mul_shoup(sv, pm.w, pm.wq, q), cv0[k], q));
                });
#pragma unroll
                for (int e = 0; e < 16; e++)
                    val[e] = sum_to_canon(acc1[e]);
                emit_rows_k(val, lds_wave, tid, [&](int k, unsigned off, uint64_t sv) {
                    mid_st<16>(out1 + off, add_mod(mul_shoup(sv, pm.w, pm.wq, q), cv1[k], q));
                });
                return;
            }
#pragma unroll
            for (int e = 0; e < 16; e++)
                val[e] = sum_to_canon(acc0[e]);
            store_rows(val, lds_wave, out, tid);
#pragma unroll
            for (int e = 0; e < 16; e++)
                val[e] = sum_to_canon(acc1[e]);
            store_rows(val, lds_wave, out + ((size_t)(a.K + 1) << G::n), tid);
        }

        // CLS: 0 integer-back-end targets only, 1 double-precision targets only
        template <int D1, int CLS>
#ifndef SEALHIP_KS2_INT_WAVES
#define SEALHIP_KS2_INT_WAVES 2
#endif
        __global__ void __launch_bounds__(kThreads, CLS == 0 ? SEALHIP_KS2_INT_WAVES : 2) ks2_kernel(Ks2Args a)
        {
            typedef Geo<D1> G;
            HIP_DYNAMIC_SHARED(uint64_t, lds)
            // XCD-aware order: blockIdx % 8 selects the XCD; keep every batch item of one (I, hg)
            // on one XCD and adjacent in time so the key tile is served by that XCD's L2.
            const unsigned ntile = a.ntargets * G::TILES;
            const unsigned bid = blockIdx.x;
            const unsigned xcd = bid & 7, rest = bid >> 3;
            const unsigned vbatch = a.batch * a.parts;
            const unsigned vb = rest % vbatch, tile_hi = rest / vbatch;
            const unsigned b = vb % a.batch, dg = vb / a.batch; // virtual batch item = (digit group, batch item)
            const unsigned jlen = (a.j1 - a.j0 + a.parts - 1) / a.parts;
            const unsigned j0 = a.j0 + dg * jlen, j1 = j0 + jlen < a.j1 ? j0 + jlen : a.j1;
            uint64_t *acc_part = a.acc + (((size_t)dg * a.batch * 2 * (a.K + 1)) << G::n);
            const unsigned tile = tile_hi * 8 + xcd;
            if (tile >= ntile)
                return;
            const unsigned it = tile / G::TILES, hg = tile % G::TILES;
            const unsigned I = SHL_UNIFORM(a.targets[4 * it]), prime = SHL_UNIFORM(a.targets[4 * it + 1]), koff = SHL_UNIFORM(a.targets[4 * it + 3]);
            if constexpr (CLS == 1)
                ks2_body<true, D1>(a, lds, I, prime, koff, b, hg, j0, j1, acc_part);
            else
                with_int_class(a.tb, prime, [&](auto ic) { ks2_body<false, D1, decltype(ic)::value>(a, lds, I, prime, koff, b, hg, j0, j1, acc_part); });
        }

        // per-component offsets of a digit in units of N words (ntt2_kernels.h: key_comp_offset_units), one table per workgroup
        __device__ __forceinline__ void key_offsets(unsigned *off, const FpDesc *fpd, unsigned L)
        {
            if (threadIdx.x == 0)
            {
                unsigned u = 0;
                for (unsigned c = 0; c < L; c++)
                {
                    off[c] = u;
                    u += fpd[c].qi ? 2u : 4u;
                }
                off[L] = u;
            }
            __syncthreads();
        }
        // natural order (u64) [digits][2][L][N] -> register order (ntt2_kernels.h)
        __global__ void __launch_bounds__(kThreads) key_layout_kernel(
            const uint64_t *in, uint64_t *out, const FpDesc *fpd, const ModDesc *mods, unsigned L, unsigned n_log, size_t polys)
        {
            __shared__ unsigned off[kMaxKeyComps + 1];
            key_offsets(off, fpd, L);
            const size_t N = (size_t)1 << n_log;
            const size_t total = polys * L * N;
            for (size_t i = blockIdx.x * (size_t)kThreads + threadIdx.x; i < total; i += (size_t)gridDim.x * kThreads)
            {
                const size_t p = i & (N - 1), slab = i >> n_log;
                const unsigned comp = (unsigned)(slab % L);
                const size_t poly = slab / L, digit = poly >> 1;
                const unsigned k = (unsigned)(poly & 1);
                // destination position p = hg*4096 + e*256 + tid  <-  natural hg*4096 + (tid>>4)*256 + (tid&15)*16 + e
                const size_t hg = p >> 12;
                const unsigned e = (unsigned)(p >> 8) & 15, tid = (unsigned)p & 255;
                size_t nat = (hg << 12) + ((size_t)(tid >> 4) << 8) + ((tid & 15) << 4) + e;
                const uint64_t v = in[(slab << n_log) + nat];
                uint64_t *o = out + ((digit * off[L] + off[comp]) << n_log); // this component of this digit
                if (fpd[comp].qi)
                {
                    // balanced representative in (-q/2, q/2] (see Context: the same halving of the bound for the key products);
                    // the two polynomials of a digit side by side: ks2 reads both with one 16-byte load
                    double d = fp_from_u52(v);
                    if (v > fpd[comp].qi / 2)
                        d -= fpd[comp].q;
                    o[2 * p + k] = fp_to_bits(d);
                }
                else
                {
                    // integer back end: the word and, next to it, floor(v 2^64 / q) (exact: estimate from the
                    // Barrett ratio floor(2^128 / q), then at most two corrections)
                    typedef unsigned __int128 u128;
                    const ModDesc md = mods[comp];
                    const u128 vr = (u128)v * md.ratio_hi + (((u128)v * md.ratio_lo) >> 64); // floor(v * ratio / 2^64), below 2^64 + 1
                    uint64_t est = vr > (u128)~(uint64_t)0 ? ~(uint64_t)0 : (uint64_t)vr;
                    u128 rem = ((u128)v << 64) - (u128)est * md.q;
                    while (rem >= md.q)
                    {
                        rem -= md.q;
                        est++;
                    }
                    o += (size_t)k << (n_log + 1); // the second polynomial's N pairs follow the first's
                    o[2 * p] = v;
                    o[2 * p + 1] = est;
                }
            }
        }

        // register order -> natural order, canonical words: what KSwitchKeys::save writes (kswitchkeys.cpp:47-90)
        __global__ void __launch_bounds__(kThreads) key_unlayout_kernel(
            const uint64_t *in, uint64_t *out, const FpDesc *fpd, unsigned L, unsigned n_log, size_t polys)
        {
            __shared__ unsigned off[kMaxKeyComps + 1];
            key_offsets(off, fpd, L);
            const size_t N = (size_t)1 << n_log;
            const size_t total = polys * L * N;
            for (size_t i = blockIdx.x * (size_t)kThreads + threadIdx.x; i < total; i += (size_t)gridDim.x * kThreads)
            {
                const size_t p = i & (N - 1), slab = i >> n_log;
                const unsigned comp = (unsigned)(slab % L);
                const size_t poly = slab / L, digit = poly >> 1;
                const unsigned k = (unsigned)(poly & 1);
                const size_t hg = p >> 12;
                const unsigned e = (unsigned)(p >> 8) & 15, tid = (unsigned)p & 255;
                const size_t nat = (hg << 12) + ((size_t)(tid >> 4) << 8) + ((tid & 15) << 4) + e;
                const uint64_t *o = in + ((digit * off[L] + off[comp]) << n_log);
                uint64_t v;
                if (fpd[comp].qi)
                {
                    double d = fp_from_bits(o[2 * p + k]);
                    if (d < 0)
                        d += fpd[comp].q; // balanced -> [0, q)
                    v = (uint64_t)d;
                }
                else
                    v = o[((size_t)k << (n_log + 1)) + 2 * p];
                out[(slab << n_log) + nat] = v;
            }
        }

        // Components [c0, c0 + nc) of one launch, all of arithmetic class cls (0 int, 1 fp, 2 mixed)
        struct CompRun
        {
            unsigned c0, nc;
            int cls;
        };
        // Split the components of a batch into runs of one arithmetic class (known on the host only when
        // the prime of a component is prime_first + comp): the single-class kernels need about half the
        // registers of the mixed one (128 vs 256 VGPRs for the double-precision forward passes).
        // Measured on MI355X (bench.py, C5): +2 % on the multiply+relinearize+rescale pipeline.  Two
        // alternatives were measured and dropped: a (tile, outer, comp) grid order (-5 % on the NTT) and
        // splitting a batch so that the intermediate stays below 128 MiB (-2 %).
        std::vector<CompRun> comp_runs(const NttTables &t, const uint32_t *comp_prime, unsigned prime_first, unsigned ncomp, int cls_hint = -1)
        {
            std::vector<CompRun> runs;
            if (comp_prime && (cls_hint == 0 || cls_hint == 1))
            {
                runs.push_back(CompRun{ 0, ncomp, cls_hint }); // the caller vouches for the class of every mapped prime (NttBatch::cls_hint)
                return runs;
            }
            static const bool split = !shl_ab_getenv("SEALHIP_NTT_NOSPLIT");
            unsigned c = 0;
            while (c < ncomp)
            {
                int cls = 2;
                unsigned e = ncomp;
                if (split && !comp_prime && t.fp_host)
                {
                    cls = t.fp_host[prime_first + c] ? 1 : 0;
                    e = c + 1;
                    while (e < ncomp && (t.fp_host[prime_first + e] ? 1 : 0) == cls)
                        e++;
                }
                runs.push_back(CompRun{ c, e - c, cls });
                c = e;
            }
            return runs;
        }

        // Runs of different classes touch disjoint components (and disjoint parts of the intermediate):
        // the short ones go to a side stream between a fork and a join event so that their few
        // workgroups share the chip with the long run instead of running alone before it.
        struct SideStream
        {
            hipStream_t stream = nullptr;
            hipEvent_t fork = nullptr, join = nullptr;
            bool ok = false;
            SideStream()
            {
                ok = !shl_ab_getenv("SEALHIP_NTT_NOFORK") && hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) == hipSuccess &&
                     hipEventCreateWithFlags(&fork, hipEventDisableTiming) == hipSuccess &&
                     hipEventCreateWithFlags(&join, hipEventDisableTiming) == hipSuccess;
            }
        };
        SideStream &side_stream()
        {
            static thread_local SideStream s;
            return s;
        }
        // run(r, stream) launches the kernels of one run; the longest run stays on `s`
        template <class RunFn>
        hipError_t launch_runs(const std::vector<CompRun> &runs, hipStream_t s, RunFn run, bool may_fork = true)
        {
            size_t longest = 0;
            for (size_t i = 1; i < runs.size(); i++)
                if (runs[i].nc > runs[longest].nc)
                    longest = i;
            SideStream &ss = side_stream();
            const bool fork = may_fork && runs.size() > 1 && ss.ok;
            hipError_t e;
            if (fork)
            {
                if ((e = hipEventRecord(ss.fork, s)) != hipSuccess || (e = hipStreamWaitEvent(ss.stream, ss.fork, 0)) != hipSuccess)
                    return e;
                for (size_t i = 0; i < runs.size(); i++)
                    if (i != longest && (e = run(runs[i], ss.stream)) != hipSuccess)
                        return e;
                if ((e = hipEventRecord(ss.join, ss.stream)) != hipSuccess)
                    return e;
            }
            for (size_t i = 0; i < runs.size(); i++)
                if ((!fork || i == longest) && (e = run(runs[i], s)) != hipSuccess)
                    return e;
            if (fork && (e = hipStreamWaitEvent(s, ss.join, 0)) != hipSuccess)
                return e;
            return hipSuccess;
        }

#ifndef SEALHIP_FUSED_FORK14
#define SEALHIP_FUSED_FORK14 1 // 2^14: the single-launch kernels of the two classes side by side (a CU holds ONE 1024-thread workgroup of either)
#endif
        // single-launch kernels for the integer back end (N = 2^13, 2^14); SEALHIP_NTT_NOFUSED_INT=1 keeps its two-launch engine (A/B runs)
        inline bool fused_int()
        {
            static const bool on = !shl_ab_getenv("SEALHIP_NTT_NOFUSED_INT");
            return on;
        }

        template <int D1>
        hipError_t launch_fwd(const FwdArgs &a, unsigned nouter, hipStream_t s)
        {
            typedef Geo<D1> G;
            // enough workgroups to fill the chip several times over, each looping over its share of
            // the outer items with the next tile in flight
            unsigned per = G::TILES * a.ncomp;
#ifndef SEALHIP_NTT_WG_TARGET
#define SEALHIP_NTT_WG_TARGET 8192 // round 5: 8192 workgroups per launch +1.3 % on the 2^16 leg over 4096 (four same-box rounds), 2048 -4 %, 16384 +1 % (profiles/r05_ntt_grid_variants.txt)
#endif
            unsigned chunks = (SEALHIP_NTT_WG_TARGET + per - 1) / per;
            if (const char *f = shl_ab_getenv("SEALHIP_NTT_CHUNKS")) // development / emulated builds: force the per-workgroup loop at small batches
                chunks = (unsigned)std::atoi(f) ? (unsigned)std::atoi(f) : 1;
            if (chunks > nouter)
                chunks = nouter;
            if (chunks > 65535)
                chunks = 65535;
            size_t l1 = G::rA > 0 ? G::lds1_words * 8 : 8;
            // single-launch kernels (plain in-place transforms, N = 2^13, 2^14); SEALHIP_NTT_NOFUSED=1 keeps the two-launch engine
            bool fused = false;
            unsigned fchunks = 1;
            if constexpr (D1 == 5 || D1 == 6)
            {
                static const bool fused_ok = !shl_ab_getenv("SEALHIP_NTT_NOFUSED");
                if (fused_ok && (!a.src || a.src_mode == 0) && a.epi == 0)
                {
                    static bool raised = false;
                    if (!raised)
                    {
                        // above the default 64 KiB of dynamic LDS
                        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&ntt2_fwd_fused2<D1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)FusedGeo<D1>::lds_bytes) != hipSuccess ||
                            hipFuncSetAttribute(reinterpret_cast<const void *>(&ntt2_fwd_fused2<D1, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(FusedGeo<D1, 2>::lds_bytes)) != hipSuccess)
                            return hipErrorInvalidValue;
                        raised = true;
                    }
                    fused = true;
                    // two workgroups per CU (N = 2^13) or one 1024-thread workgroup (N = 2^14);
                    // a few loop iterations per workgroup so that the prefetch and the hoisted twiddles pay
                    const unsigned want = D1 == 5 ? 2048 : 1024;
                    fchunks = (want + a.ncomp - 1) / a.ncomp;
                    if (const char *f = std::getenv("SEALHIP_NTT_FCHUNKS")) // tests: force the per-workgroup loop at small batches
                        fchunks = (unsigned)std::atoi(f) ? (unsigned)std::atoi(f) : 1;
                    if (fchunks > nouter)
                        fchunks = nouter;
                    if (fchunks > 65535)
                        fchunks = 65535;
                }
            }
            return launch_runs(comp_runs(a.t, a.comp_prime, a.prime_first, a.ncomp, a.cls_hint), s, [&](const CompRun &r, hipStream_t st) {
                FwdArgs g = a;
                g.comp0 = r.c0;
#ifndef SEALHIP_MID_PACK
#define SEALHIP_MID_PACK 1
#endif
                // N = 2^16, plain transform, double-precision components: the intermediate in 13 words per 16 values (kPackWords)
                // (only together with the hoisted pass 2, i.e. when the workgroups loop: the small-batch kernels stay as they are)
                g.mid_pack = SEALHIP_MID_PACK && D1 == 8 && r.cls == 1 && a.epi == 0 && chunks < nouter ? 1 : 0;
                if constexpr (D1 == 5 || D1 == 6)
                {
                    if (fused && r.cls == 1)
                    {
                        hipLaunchKernelGGL((ntt2_fwd_fused2<D1, 1>), dim3(1, r.nc, fchunks), dim3(FusedGeo<D1>::TEAMS * kThreads), FusedGeo<D1>::lds_bytes, st, g);
                        return hipGetLastError();
                    }
                    if (fused && r.cls == 0 && fused_int())
                    {
                        constexpr size_t lds_int = FusedGeo<D1, 2>::lds_bytes;
                        // (workgroups per component of the integer class alone, 256 ... 4096 at 4096 polynomials: the default is the best
                        // or equal, profiles/r05_configs1_bisect.txt)
                        hipLaunchKernelGGL((ntt2_fwd_fused2<D1, 0>), dim3(1, r.nc, fchunks), dim3(FusedGeo<D1>::TEAMS * kThreads), lds_int, st, g);
                        return hipGetLastError();
                    }
                }
                dim3 grid(G::TILES, r.nc, chunks);
                if constexpr (D1 == 8)
                {
                    // one launch, intermediate in a re-used ring (ntt2_ring.hip): plain in-place transforms of double-precision components
                    if (r.cls == 1 && !a.src && a.epi == 0 && a.ring)
                    {
                        const unsigned z = ntt2_ring_teams(r.nc, nouter);
                        if (z && ntt2_ring_run_words(r.nc, z) <= a.ring_words && ntt2_ring_stream_ok(st))
                        {
                            const NttRingRun run{ a.data, a.outer_stride, a.comp_prime, a.prime_first, a.ncomp, r.c0, r.nc, nouter, a.lazy, z, a.ring, a.ring_words };
                            return ntt2_ring_launch(a.t, run, st);
                        }
                    }
                }
                if constexpr (D1 == 8)
                {
                    if (g.mid_pack)
                    {
                        hipLaunchKernelGGL((ntt2_fwd_p1<D1, 5>), grid, dim3(kThreads), l1, st, g);
                        hipError_t ep = hipGetLastError();
                        if (ep != hipSuccess)
                            return ep;
                        hipLaunchKernelGGL((ntt2_fwd_p2<D1, 5>), grid, dim3(kThreads), (kLds2Words + 240) * 8, st, g);
                        return hipGetLastError();
                    }
                }
                if (r.cls == 1)
                    hipLaunchKernelGGL((ntt2_fwd_p1<D1, 1>), grid, dim3(kThreads), l1, st, g);
                else if (r.cls == 0)
                    hipLaunchKernelGGL((ntt2_fwd_p1<D1, 0>), grid, dim3(kThreads), l1, st, g);
                else
                    hipLaunchKernelGGL((ntt2_fwd_p1<D1, 2>), grid, dim3(kThreads), l1, st, g);
                hipError_t e = hipGetLastError();
                if (e != hipSuccess)
                    return e;
                static const int p2_hoist = shl_ab_getenv("SEALHIP_P2_HOIST") ? std::atoi(shl_ab_getenv("SEALHIP_P2_HOIST")) : 4;
                if (r.cls == 1 && a.epi == 0 && chunks < nouter && p2_hoist == 4)
                    hipLaunchKernelGGL((ntt2_fwd_p2<D1, 4>), grid, dim3(kThreads), (kLds2Words + 240) * 8, st, g);
                else if (r.cls == 1 && a.epi == 0 && chunks < nouter && p2_hoist == 3)
                    hipLaunchKernelGGL((ntt2_fwd_p2<D1, 3>), grid, dim3(kThreads), kLds2Words * 8, st, g);
                else if (r.cls == 1)
                    hipLaunchKernelGGL((ntt2_fwd_p2<D1, 1>), grid, dim3(kThreads), kLds2Words * 8, st, g);
                else if (r.cls == 0)
                    hipLaunchKernelGGL((ntt2_fwd_p2<D1, 0>), grid, dim3(kThreads), kLds2Words * 8, st, g);
                else
                    hipLaunchKernelGGL((ntt2_fwd_p2<D1, 2>), grid, dim3(kThreads), kLds2Words * 8, st, g);
                return hipGetLastError();
            }, !(fused && fused_int()) || (D1 == 6 && SEALHIP_FUSED_FORK14));
            // single-launch kernels of both classes at 2^13: one after the other (at 2^14 a CU holds ONE 1024-thread workgroup of either class,
            // side by side costs nothing and the mixed chain {60,6x50,60} runs at 0.32 / 0.34 instead of 0.29 / 0.29).  Side by side at 2^13 a CU holds one workgroup of each (LDS), and
            // each class loses the partner workgroup that covers its latencies: measured at configs[1]'s chain {60,40,40,60}, 8192
            // polynomials: 2.83 / 3.02 TB/s forked, 2.97 / 3.12 in sequence (profiles/r03_configs1_chain.txt)
        }

        template <int D1>
        hipError_t launch_tail2(const Tail2Args &a, unsigned nouter, hipStream_t s)
        {
            typedef Geo<D1> G;
            unsigned per = G::TILES * a.f.ncomp;
#ifndef SEALHIP_TAIL2_WG_TARGET
#define SEALHIP_TAIL2_WG_TARGET 4096
#endif
            unsigned chunks = (SEALHIP_TAIL2_WG_TARGET + per - 1) / per;
            if (chunks > nouter)
                chunks = nouter;
            if (chunks > 65535)
                chunks = 65535;
            const size_t l1 = G::rA > 0 ? G::lds1_words * 8 : 8;
            return launch_runs(comp_runs(a.f.t, nullptr, a.f.prime_first, a.f.ncomp), s, [&](const CompRun &r, hipStream_t st) {
                Tail2Args g = a;
                g.f.comp0 = r.c0;
                dim3 grid(G::TILES, r.nc, chunks);
                if (r.cls == 1)
                    hipLaunchKernelGGL((ntt2_tail2_p1<D1, 1>), grid, dim3(kThreads), l1, st, g);
                else if (r.cls == 0)
                    hipLaunchKernelGGL((ntt2_tail2_p1<D1, 0>), grid, dim3(kThreads), l1, st, g);
                else
                    hipLaunchKernelGGL((ntt2_tail2_p1<D1, 2>), grid, dim3(kThreads), l1, st, g);
                hipError_t e = hipGetLastError();
                if (e != hipSuccess)
                    return e;
                if (r.cls == 1)
                    hipLaunchKernelGGL((ntt2_tail2_p2<D1, 1>), grid, dim3(kThreads), kLds2Words * 8, st, g);
                else if (r.cls == 0)
                    hipLaunchKernelGGL((ntt2_tail2_p2<D1, 0>), grid, dim3(kThreads), kLds2Words * 8, st, g);
                else
                    hipLaunchKernelGGL((ntt2_tail2_p2<D1, 2>), grid, dim3(kThreads), kLds2Words * 8, st, g);
                return hipGetLastError();
            });
        }

        template <int D1>
        hipError_t launch_inv(const InvArgs &a, unsigned nouter, hipStream_t s)
        {
            typedef Geo<D1> G;
            bool fused = false;
            unsigned fchunks = 1;
            if constexpr (D1 == 5 || D1 == 6)
            {
                static const bool fused_ok = !shl_ab_getenv("SEALHIP_NTT_NOFUSED");
                if (fused_ok && !a.prod_x && !a.src_gal) // (the product and automorphism sources are features of the two-pass kernels)
                {
                    static bool raised = false;
                    if (!raised)
                    {
                        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&ntt2_inv_fused2<D1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)FusedGeo<D1>::lds_bytes) != hipSuccess ||
                            hipFuncSetAttribute(reinterpret_cast<const void *>(&ntt2_inv_fused2<D1, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(FusedGeo<D1, 2>::lds_bytes)) != hipSuccess)
                            return hipErrorInvalidValue;
                        raised = true;
                    }
                    fused = true;
                    const unsigned want = D1 == 5 ? 2048 : 1024;
                    fchunks = (want + a.ncomp - 1) / a.ncomp;
                    // round 5 (profiles/r05_configs1_bisect.txt): the 2^13 inverse re-reads its phase-B twiddles per transform anyway, so
                    // long loops buy it nothing - about FOUR transforms per workgroup is its optimum (8192 polynomials: 2048 workgroups
                    // per component +3.3 % over 512; 4096 polynomials: 1024 = 512, 2048 - two transforms each - loses 5 %)
                    if (D1 == 5 && nouter / 4 > fchunks)
                        fchunks = nouter / 4;
                    if (const char *f = std::getenv("SEALHIP_NTT_FCHUNKS"))
                        fchunks = (unsigned)std::atoi(f) ? (unsigned)std::atoi(f) : 1;
                    if (fchunks > nouter)
                        fchunks = nouter;
                    if (fchunks > 65535) // gridDim.z (nouter / 4 is not bounded by `want` any more)
                        fchunks = 65535;
                }
            }
            return launch_runs(comp_runs(a.t, a.comp_prime, a.prime_first, a.ncomp, a.cls_hint), s, [&](const CompRun &r, hipStream_t st) {
                InvArgs g = a;
                g.comp0 = r.c0;
                g.nouter = nouter;
                if constexpr (D1 == 5 || D1 == 6)
                {
                    if (fused && r.cls == 1)
                    {
                        hipLaunchKernelGGL((ntt2_inv_fused2<D1, 1>), dim3(1, r.nc, fchunks), dim3(FusedGeo<D1>::TEAMS * kThreads), FusedGeo<D1>::lds_bytes, st, g);
                        return hipGetLastError();
                    }
                    if (fused && r.cls == 0 && fused_int())
                    {
                        constexpr size_t lds_int = FusedGeo<D1, 2>::lds_bytes;
                        hipLaunchKernelGGL((ntt2_inv_fused2<D1, 0>), dim3(1, r.nc, fchunks), dim3(FusedGeo<D1>::TEAMS * kThreads), lds_int, st, g);
                        return hipGetLastError();
                    }
                }
                dim3 grid(G::TILES, r.nc, nouter);
                if (r.cls == 1)
                    hipLaunchKernelGGL((ntt2_inv_pa<D1, 1>), grid, dim3(kThreads), kLds2Words * 8, st, g);
                else if (r.cls == 0)
                    hipLaunchKernelGGL((ntt2_inv_pa<D1, 0>), grid, dim3(kThreads), kLds2Words * 8, st, g);
                else
                    hipLaunchKernelGGL((ntt2_inv_pa<D1, 2>), grid, dim3(kThreads), kLds2Words * 8, st, g);
                hipError_t e = hipGetLastError();
                if (e != hipSuccess)
                    return e;
                // development builds, TIMING ONLY (wrong words): the step without the inverse transforms' last pass - the most that
                // fusing it into the next kernel's prologue could return (profiles/r05_inv_pb_fusion_bound.txt)
                // (value = number of inverse launches that still run it: the warm-up leaves realistic words in the recycled buffers -
                // with stale zeros in them the key switch's arithmetic draws less power and the chip clocks 10 % higher, which is not
                // the saving being measured)
                static const char *skip_env = shl_ab_getenv("SEALHIP_AB_SKIP_INV_PB");
                static long skip_after = skip_env ? std::atol(skip_env) : -1;
                static std::atomic<long> inv_launches{ 0 };
                if (skip_after >= 0 && inv_launches.fetch_add(1) >= skip_after)
                    return (hipError_t)hipSuccess;
                if (r.cls == 1)
                    hipLaunchKernelGGL((ntt2_inv_pb<D1, 1>), grid, dim3(kThreads), G::lds1_words * 8, st, g);
                else if (r.cls == 0)
                    hipLaunchKernelGGL((ntt2_inv_pb<D1, 0>), grid, dim3(kThreads), G::lds1_words * 8, st, g);
                else
                    hipLaunchKernelGGL((ntt2_inv_pb<D1, 2>), grid, dim3(kThreads), G::lds1_words * 8, st, g);
                return hipGetLastError();
            }, !(fused && fused_int()) || (D1 == 6 && SEALHIP_FUSED_FORK14));
        }

        template <int D1>
        hipError_t launch_ks(const Ks1Args &a1, const Ks2Args &a2, unsigned batch, unsigned n_int, hipStream_t s, int order1, bool no_class_fork)
        {
            typedef Geo<D1> G;
            if (a1.ntargets == 0)
                return hipSuccess;
            const size_t l1 = G::rA > 0 ? G::lds1_words * 8 : 8;
            const unsigned vbatch = batch * a1.parts; // (digit group, batch item) pairs
            const unsigned groups = vbatch * G::TILES;
            const unsigned n_fp = a1.ntargets - n_int;
            const size_t l2_fp = kLds2Words * 8 + (240 + 3840) * sizeof(double); // the tile's twiddles staged in LDS
            const size_t l2_int = kLds2Words * 8 + (240 + (SEALHIP_KS2_INT_TWB3 ? 2048 : 0)) * sizeof(ShoupOp);
            if (n_int && l2_int > 65536)
            {
                static bool raised_int = false;
                if (!raised_int)
                {
                    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&ks2_kernel<D1, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l2_int) != hipSuccess)
                        return hipErrorInvalidValue;
                    raised_int = true;
                }
            }
            if (n_fp && l2_fp > 65536)
            {
                // more than the default 64 KiB of dynamic LDS per workgroup (gfx950 has 160 KiB per CU)
                static bool raised = false;
                if (!raised)
                {
                    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&ks2_kernel<D1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l2_fp) != hipSuccess)
                        return hipErrorInvalidValue;
                    raised = true;
                }
            }
            // both passes for the targets of one arithmetic class (targets = [integer moduli..., double-precision moduli...])
            auto run_class = [&](bool fp, hipStream_t st) -> hipError_t {
                const unsigned t0 = fp ? n_int : 0, nt = fp ? n_fp : n_int;
                if (!nt)
                    return hipSuccess;
                Ks1Args c1 = a1;
                c1.targets = a1.targets + 2 * t0;
                c1.ntargets = nt;
                const dim3 g1(((groups + 7) / 8) * nt * 8);
                // digit-resident order (ks1t_kernel) when its grid fills the chip, and for the complete digit range only: a digit
                // group or a rank's slice keeps the order whose grid does not shrink with the slice.  SEALHIP_KS1_ORDER=0 / 1 forces
                static const char *order_env = shl_ab_getenv("SEALHIP_KS1_ORDER");
                const unsigned g1t = batch * (a1.j1 - a1.j0) * G::TILES;
                const bool digit_resident = order_env ? order_env[0] == '1' : order1 >= 0 ? (order1 == 1 && a1.parts == 1) : (a1.parts == 1 && g1t >= 4096);
                if (digit_resident && fp)
                {
                    // runs of digits of one kind: below 2^52 (known on the host as "a double-precision prime": below 2^50) or not
                    const unsigned char *small = a1.tb.fp_host;
                    for (unsigned ja = a1.j0; ja < a1.j1;)
                    {
                        const bool sm = small && small[ja];
                        unsigned jb = ja + 1;
                        while (jb < a1.j1 && (small && small[jb]) == sm)
                            jb++;
                        Ks1Args cr = c1;
                        cr.j0 = ja;
                        cr.j1 = jb;
                        const dim3 gr(batch * (jb - ja) * G::TILES);
                        if (sm)
                            hipLaunchKernelGGL((ks1t_kernel<true, D1, true>), gr, dim3(kThreads), l1, st, cr);
                        else
                            hipLaunchKernelGGL((ks1t_kernel<true, D1, false>), gr, dim3(kThreads), l1, st, cr);
                        ja = jb;
                    }
                }
                else if (digit_resident)
                    hipLaunchKernelGGL((ks1t_kernel<false, D1>), dim3(g1t), dim3(kThreads), l1, st, c1);
                else if (fp)
                    hipLaunchKernelGGL((ks1_kernel<true, D1>), g1, dim3(kThreads), l1, st, c1);
                else
                    hipLaunchKernelGGL((ks1_kernel<false, D1>), g1, dim3(kThreads), l1, st, c1);
                hipError_t e = hipGetLastError();
                if (e != hipSuccess)
                    return e;
                Ks2Args c2 = a2;
                c2.targets = a2.targets + 4 * t0;
                c2.ntargets = nt;
                const unsigned ntile = nt * G::TILES;
                if (fp)
                    hipLaunchKernelGGL((ks2_kernel<D1, 1>), dim3(((ntile + 7) / 8) * vbatch * 8), dim3(kThreads), l2_fp, st, c2);
                else
                    hipLaunchKernelGGL((ks2_kernel<D1, 0>), dim3(((ntile + 7) / 8) * vbatch * 8), dim3(kThreads), l2_int, st, c2);
                return hipGetLastError();
            };
            // The integer-back-end targets (60-bit moduli) are latency-bound at two waves per SIMD, the
            // double-precision ones are issue-bound: on two streams their workgroups share the CUs.
            SideStream &ss = side_stream();
            static const bool fork_ok = !shl_ab_getenv("SEALHIP_KS_NOFORK");
            hipError_t e;
            if (n_int && n_fp && ss.ok && fork_ok && !no_class_fork)
            {
                if ((e = hipEventRecord(ss.fork, s)) != hipSuccess || (e = hipStreamWaitEvent(ss.stream, ss.fork, 0)) != hipSuccess)
                    return e;
                if ((e = run_class(false, ss.stream)) != hipSuccess || (e = hipEventRecord(ss.join, ss.stream)) != hipSuccess)
                    return e;
                if ((e = run_class(true, s)) != hipSuccess)
                    return e;
                return hipStreamWaitEvent(s, ss.join, 0);
            }
            if ((e = run_class(false, s)) != hipSuccess)
                return e;
            return run_class(true, s);
        }
    } // namespace

    bool ntt2_supports(int log_n)
    {
        return log_n >= 13 && log_n <= 16;
    }

    size_t ntt2_ring_words(const NttTables &t, const NttBatch &b)
    {
        if (t.log_n != 16 || b.src || b.epi || b.tail2 || b.ncomp == 0 || b.nouter == 0)
            return 0;
        if (!ntt2_ring_teams(1, b.nouter)) // the kernels are opt-in (SEALHIP_NTT_RING) and need a batch that loops: nothing to plan otherwise
            return 0;
        // the double-precision runs of the batch (one after the other on the launcher's stream: they share the scratch)
        size_t words = 0;
        for (const CompRun &r : comp_runs(t, b.comp_prime, b.prime_first, b.ncomp, b.cls_hint))
            if (r.cls == 1)
            {
                const unsigned z = ntt2_ring_teams(r.nc, b.nouter);
                if (z)
                    words = std::max(words, ntt2_ring_run_words(r.nc, z));
            }
        return words;
    }

    hipError_t ntt2_forward(const NttTables &t, const NttBatch &b, int out_lazy, uint64_t *mid, hipStream_t stream, uint64_t *ring, size_t ring_words)
    {
        if (b.ncomp == 0 || b.nouter == 0)
            return hipSuccess;
        FwdArgs a;
        a.ring = ring;
        a.ring_words = ring ? ring_words : 0;
        a.data = b.data;
        a.outer_stride = b.outer_stride;
        a.mid = mid;
        a.src = b.src;
        a.src_outer_stride = b.src_outer_stride;
        a.src_ncomp = b.src_ncomp ? b.src_ncomp : 1;
        a.src_mode = b.src ? b.src_mode : 0;
        a.src_half = b.src_half;
        a.src_q = b.src_q;
        a.src_fix = b.src_fix;
        a.comp_prime = b.comp_prime;
        a.cls_hint = b.cls_hint;
        a.prime_first = b.prime_first;
        a.ncomp = b.ncomp;
        a.comp0 = 0;
        a.nouter = b.nouter;
        a.lazy = out_lazy;
        a.epi = b.epi;
        a.epi_a = b.epi_a;
        a.epi_a_stride = b.epi_a_stride;
        a.epi_mul = b.epi_mul;
        a.epi_out0 = b.epi_out0;
        a.epi_out1 = b.epi_out1;
        a.epi_out_stride = b.epi_out_stride;
        a.mid_pack = 0;
        a.t = t;
        if (b.tail2)
        {
            // NttTail2: mapped source 2 + the double-division tail; prime of a component = prime_first + comp
            if (!b.src || (b.src_mode != 2 && b.src_mode != 3) || b.comp_prime || !b.epi_a || !b.epi_mul || !b.epi_out0 || !b.epi_out1)
                return hipErrorInvalidValue;
            Tail2Args t2{ a, *b.tail2 };
            switch (t.log_n)
            {
            case 13:
                return launch_tail2<5>(t2, b.nouter, stream);
            case 14:
                return launch_tail2<6>(t2, b.nouter, stream);
            case 15:
                return launch_tail2<7>(t2, b.nouter, stream);
            case 16:
                return launch_tail2<8>(t2, b.nouter, stream);
            default:
                return hipErrorInvalidValue;
            }
        }
        switch (t.log_n)
        {
        case 13:
            return launch_fwd<5>(a, b.nouter, stream);
        case 14:
            return launch_fwd<6>(a, b.nouter, stream);
        case 15:
            return launch_fwd<7>(a, b.nouter, stream);
        case 16:
            return launch_fwd<8>(a, b.nouter, stream);
        default:
            return hipErrorInvalidValue;
        }
    }

    hipError_t ntt2_inverse(const NttTables &t, const NttBatch &b, int out_lazy, uint64_t *mid, hipStream_t stream)
    {
        if (b.ncomp == 0 || b.nouter == 0)
            return hipSuccess;
        InvArgs a;
        a.data = b.data;
        a.outer_stride = b.outer_stride;
        a.src = b.src ? b.src : b.data;
        a.src_outer_stride = b.src ? b.src_outer_stride : b.outer_stride;
        a.mid = mid;
        a.comp_prime = b.comp_prime;
        a.cls_hint = b.cls_hint;
        a.prime_first = b.prime_first;
        a.ncomp = b.ncomp;
        a.comp0 = 0;
        a.nouter = 0; // set per launch
        a.lazy = out_lazy;
        if (b.out_add && out_lazy)
            return hipErrorInvalidValue;
        a.out_add = b.out_add;
        a.prod_x = b.prod_x;
        a.prod_y = b.prod_y;
        a.prod_batch = b.prod_batch;
        a.prod_outer0 = b.prod_outer0;
        a.prod_out = b.prod_out;
        a.prod_out_stride = b.prod_out_stride;
        a.src_gal = b.src_galois_elt;
        if (b.src_galois_elt && (!b.src || b.prod_x || !(b.src_galois_elt & 1)))
            return hipErrorInvalidValue;
        if (b.prod_x && (!b.prod_y || !b.prod_batch || (size_t)b.nouter + b.prod_outer0 > (size_t)3 * b.prod_batch || !b.src_outer_stride))
            return hipErrorInvalidValue;
        if (b.prod_x && !b.prod_outer0 && b.nouter != 3 * b.prod_batch) // the whole product, as before round 6
            return hipErrorInvalidValue;
        if (b.prod_out && (!b.prod_x || !b.prod_out_stride))
            return hipErrorInvalidValue;
        if (b.prod_x)
            a.src_outer_stride = b.src_outer_stride;
        a.t = t;
        const unsigned zmax = 65535;
        for (unsigned z0 = 0; z0 < b.nouter; z0 += zmax)
        {
            unsigned nz = b.nouter - z0 < zmax ? b.nouter - z0 : zmax;
            InvArgs az = a;
            az.prod_outer0 = b.prod_outer0 + z0;
            if (az.prod_out)
                az.prod_out = a.prod_out + (size_t)z0 * a.prod_out_stride;
            az.data = a.data + (size_t)z0 * a.outer_stride;
            az.src = a.src + (size_t)z0 * a.src_outer_stride;
            az.mid = a.mid + (((size_t)z0 * a.ncomp) << t.log_n);
            hipError_t e;
            switch (t.log_n)
            {
            case 13:
                e = launch_inv<5>(az, nz, stream);
                break;
            case 14:
                e = launch_inv<6>(az, nz, stream);
                break;
            case 15:
                e = launch_inv<7>(az, nz, stream);
                break;
            case 16:
                e = launch_inv<8>(az, nz, stream);
                break;
            default:
                return hipErrorInvalidValue;
            }
            if (e != hipSuccess)
                return e;
        }
        return hipSuccess;
    }

    hipError_t ks_fused(const NttTables &t, const KsFusedArgs &k, hipStream_t stream)
    {
        Ks1Args a1;
        a1.t = k.t;
        a1.mid = k.mid;
        a1.targets = k.targets1;
        a1.ntargets = k.ntargets;
        a1.K = k.K;
        a1.batch = k.batch;
        a1.skip_diag = k.target_ntt != nullptr || k.fold_x != nullptr; // (a deferred product: the diagonal terms are x1 y1, formed in ks2)
        a1.j0 = k.j0;
        a1.j1 = k.j1;
        a1.parts = k.parts ? k.parts : 1;
        a1.tb = t;
        if (k.fold_c0 && (a1.parts > 1 || !k.fold_pm || k.j0 != 0 || k.j1 != k.K))
            return hipErrorInvalidValue; // the addend may only join the COMPLETE sum of a component
        if (k.fold_x && (k.fold_c0 || !k.fold_y || !k.fold_plane || !k.fold_pm || a1.parts > 1 || k.j0 != 0 || k.j1 != k.K))
            return hipErrorInvalidValue;
        Ks2Args a2;
        a2.mid = k.mid;
        a2.target = k.target_ntt;
        a2.key = k.key;
        a2.key_digit_words = (size_t)key_digit_units(t, k.L) << t.log_n;
        a2.acc = k.acc;
        a2.targets = k.targets2;
        a2.ntargets = k.ntargets;
        a2.K = k.K;
        a2.L = k.L;
        a2.batch = k.batch;
        a2.j0 = k.j0;
        a2.j1 = k.j1;
        a2.key_digit0 = k.key_digit0;
        a2.parts = a1.parts;
        a2.fold_c0 = k.fold_c0;
        a2.fold_c1 = k.fold_c1;
        a2.fold_pm = k.fold_pm;
        a2.fold_x = k.fold_x;
        a2.fold_y = k.fold_y;
        a2.fold_plane = k.fold_plane;
        a2.gal = k.galois_elt;
        if (k.galois_elt && (!(k.galois_elt & 1) || !k.target_ntt || !k.fold_c0 || k.fold_c1 || k.fold_x))
            return hipErrorInvalidValue; // a rotation: target and first addend through the index map, the second addend zero
        a2.tb = t;
        switch (t.log_n)
        {
        case 13:
            return launch_ks<5>(a1, a2, k.batch, k.n_int, stream, k.order1, k.no_class_fork);
        case 14:
            return launch_ks<6>(a1, a2, k.batch, k.n_int, stream, k.order1, k.no_class_fork);
        case 15:
            return launch_ks<7>(a1, a2, k.batch, k.n_int, stream, k.order1, k.no_class_fork);
        case 16:
            return launch_ks<8>(a1, a2, k.batch, k.n_int, stream, k.order1, k.no_class_fork);
        default:
            return hipErrorInvalidValue;
        }
    }

    hipError_t key_to_register_order(const NttTables &t, const uint64_t *in, uint64_t *out, unsigned L, size_t digits, hipStream_t stream)
    {
        if (L > kMaxKeyComps)
            return hipErrorInvalidValue;
        const size_t polys = 2 * digits;
        size_t total = (polys * L) << t.log_n;
        size_t blocks = (total + kThreads - 1) / kThreads;
        if (blocks > 4096)
            blocks = 4096;
        hipLaunchKernelGGL(key_layout_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, stream, in, out, t.fpd, t.mods, L, (unsigned)t.log_n, polys);
        return hipGetLastError();
    }
    hipError_t key_from_register_order(const NttTables &t, const uint64_t *in, uint64_t *out, unsigned L, size_t digits, hipStream_t stream)
    {
        if (L > kMaxKeyComps)
            return hipErrorInvalidValue;
        const size_t polys = 2 * digits;
        size_t total = (polys * L) << t.log_n;
        size_t blocks = (total + kThreads - 1) / kThreads;
        if (blocks > 4096)
            blocks = 4096;
        hipLaunchKernelGGL(key_unlayout_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, stream, in, out, t.fpd, L, (unsigned)t.log_n, polys);
        return hipGetLastError();
    }
} // namespace sealhip
