// Device-side building blocks of the two-pass NTT engine, shared by ntt2_kernels.hip (the two-launch and key-switching kernels) and
// ntt2_ring.hip (the one-launch ring kernel of N = 2^16): geometry, the packed intermediate, butterfly stages and phases, resident
// twiddles, the bodies of the two passes' tiles, the wave-local transposes, and the argument block of the forward kernels.
// Everything lives in an anonymous namespace: each translation unit gets its own copy, nothing here is an interface.
// See ntt2_kernels.hip for the decomposition and field.h for the two arithmetic back ends.
#pragma once
#include "ntt2_kernels.h"
#include <cstdlib>
#include <type_traits>

// minimum waves per SIMD requested for the double-precision-only forward kernels (register budget
// 512 / waves): 4 keeps them at 128 VGPRs with a handful of spilled words
#ifndef SEALHIP_FP_WAVES_P1
#define SEALHIP_FP_WAVES_P1 4
#endif
#ifndef SEALHIP_FP_WAVES_P2
#define SEALHIP_FP_WAVES_P2 4
#endif

namespace sealhip
{
    namespace
    {
        constexpr int kThreads = 256;
        constexpr unsigned kMaxKeyComps = 64; // SEAL_COEFF_MOD_COUNT_MAX
        // Where word (e, tid) of a 4096-word row tile of the forward passes' intermediate sits (e = column block, tid = (row u = tid >> 4,
        // column-in-block v = tid & 15)).  Tile order (default): e*256 + tid - the 16 rows of a workgroup interleaved in 2 KiB blocks, a
        // workgroup of pass 2 reads one 32 KiB run.  -DSEALHIP_MID_WAVE_MAJOR (measured in round 4, profiles/r04_ntt_wave_major.txt): the
        // four rows of a WAVE contiguous - wave*1024 + e*64 + lane - so that every wave of pass 2 reads its own 8 KiB run.
        // Streaming hints (round 4): the intermediate of a two-pass transform is written once and read once, gigabytes later - it has no
        // business displacing what the L2 holds for reuse (key tiles, digit tiles, twiddles).  SEALHIP_KS_NT is a bit mask of where the
        // non-temporal form is used: 1 pass-1 stores, 2 ks2's double-precision loads, 4 the other forward pass-2 loads, 8 the inverse
        // passes' intermediate (store and load), 16 single-use operands and results of the passes (plain sources, tail operands, final stores).  Measured in profiles/r04_nontemporal.txt; the default below is what won.
#ifndef SEALHIP_KS_NT
#define SEALHIP_KS_NT 31 // headline step 8.82 -> 9.18 k ct/s same-box (+4.0 %); 15: +2.5 %; with the tensor product's loads (32): +2.9 %
#endif
        // Wave priorities (round 5 experiment, profiles/r05_ldsdma_setprio.txt): a CU holds two to four workgroups of these kernels in
        // different phases.  SEALHIP_PRIO = 0 none (default); 1: s_setprio 1 while a wave does arithmetic, 0 while it issues its
        // loads; 2: the other way round (memory instructions first).
#ifndef SEALHIP_PRIO
#define SEALHIP_PRIO 0
#endif
        template <int WHEN> // 1 = entering the arithmetic, 2 = entering the load issue
        __device__ __forceinline__ void prio_phase()
        {
#if defined(__HIP_DEVICE_COMPILE__)
            if constexpr (SEALHIP_PRIO != 0)
                __builtin_amdgcn_s_setprio(SEALHIP_PRIO == WHEN ? 1 : 0);
#endif
        }
        template <int BIT>
        __device__ __forceinline__ uint64_t mid_ld(const uint64_t *p)
        {
#if defined(__HIP_DEVICE_COMPILE__)
            if constexpr ((SEALHIP_KS_NT & BIT) != 0)
                return __builtin_nontemporal_load(p);
#endif
            return *p;
        }
        template <int BIT>
        __device__ __forceinline__ void mid_st(uint64_t *p, uint64_t v)
        {
#if defined(__HIP_DEVICE_COMPILE__)
            if constexpr ((SEALHIP_KS_NT & BIT) != 0)
            {
                __builtin_nontemporal_store(v, p);
                return;
            }
#endif
            *p = v;
        }
        // two adjacent words (16-byte aligned) with one 16-byte access
        template <int BIT>
        __device__ __forceinline__ void mid_ld2(const uint64_t *p, uint64_t &a, uint64_t &b)
        {
#if defined(__HIP_DEVICE_COMPILE__)
            typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));
            const u64x2 *q = reinterpret_cast<const u64x2 *>(p);
            const u64x2 v = (SEALHIP_KS_NT & BIT) != 0 ? __builtin_nontemporal_load(q) : *q;
            a = v.x;
            b = v.y;
#else
            a = p[0];
            b = p[1];
#endif
        }
        template <int BIT>
        __device__ __forceinline__ void mid_st2(uint64_t *p, uint64_t a, uint64_t b)
        {
#if defined(__HIP_DEVICE_COMPILE__)
            typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));
            u64x2 *q = reinterpret_cast<u64x2 *>(p);
            const u64x2 v = { a, b };
            if constexpr ((SEALHIP_KS_NT & BIT) != 0)
                __builtin_nontemporal_store(v, q);
            else
                *q = v;
#else
            p[0] = a;
            p[1] = b;
#endif
        }
#ifdef SEALHIP_MID_WAVE_MAJOR
        __device__ __forceinline__ unsigned mid_lane(unsigned tid) { return (tid >> 6) * 1024 + (tid & 63); }
        constexpr unsigned kMidRow = 64;
#else
        __device__ __forceinline__ unsigned mid_lane(unsigned tid) { return tid; }
        constexpr unsigned kMidRow = 256;
#endif
        template <int D1>
        struct Geo
        {
            static constexpr int rA = D1 - 4;                 // stages in phase A of pass 1
            static constexpr int LC = 12 - D1;                // log2(columns per pass-1 tile)
            static constexpr int C = 1 << LC;
            static constexpr int CP = C + (C == 16 ? 1 : 0);  // padded LDS row (words)
            static constexpr int ROWS = 1 << D1;
            static constexpr int TILES = 1 << (D1 - 4);       // tiles per transform, both passes
            static constexpr int n = D1 + 8;
            static constexpr size_t lds1_words = (size_t)ROWS * CP;
        };
        // pass-2 wave-local exchange buffer: 16 rows of 256 words, 2 pad words per 16
        constexpr int kRowWords = 16 * 18;
        constexpr size_t kLds2Words = 16 * kRowWords;

        // ---- Packed intermediate of the plain double-precision forward transform at N = 2^16 (round 5, SEALHIP_MID_PACK).
        // Between the passes a residue of a prime below 2^50 is an integer with |x| <= q/2 < 2^49 held in a double (1.80 q < 2^51 with
        // the lean fix() placement): 64 bits for at most 52.
        // x + 1.5 * 2^52 has x + 2^51 in its 52 mantissa bits (offset binary; one v_add_f64 each way), and the 16 values a pass-1
        // thread owns - rows h_lo = 0..15 of one column of a 16 x 16 block - are stored as 13 words instead of 16:
        //   words 0..7   the low 32 bits of values 2k, 2k+1           words 8..11  bits 32..47 of values 4(k-8) .. 4(k-8)+3
        //   word 12      bits 48..51 of value r at bit 4r
        // block (row tile hg, column block cg) = 13 x 16 words at mid + (hg*16 + cg) * 208, word k of column v at k*16 + v: pass 1
        // stores 13 instead of 16 128-byte runs per thread, pass 2 loads a block's 208 words with coalesced 8-byte loads (thread
        // (e, v) takes pack (cg = e, v)), decodes all sixteen rows and hands them to their owners (u, v) through LDS.
        // 13/16 of the intermediate's bytes in both directions; the values are the same doubles, so the results are the same words.
        constexpr unsigned kPackWords = 13, kPackBlock = kPackWords * 16; // words per pack / per 16 x 16 block
        // SEALHIP_PACK_VEC16 (round 6): the twelve words 0..11 of a pack travel as six 16-byte pairs - pair j of column v at j*32 + v*2,
        // word 12 at 192 + v - so that either pass issues 7 instead of 13 memory instructions per thread and the runs are 256 bytes
#ifndef SEALHIP_PACK_VEC16
#define SEALHIP_PACK_VEC16 1
#endif
        constexpr double kPackMagic = 6755399441055744.0;                  // 2^52 + 2^51
        constexpr unsigned kPackLdsRow = 272;                              // words between the rows u of the hand-over buffer
        __device__ __forceinline__ void pack52(const double (&x)[16], uint64_t (&w)[13])
        {
            uint32_t lo[16], hi[16];
#pragma unroll
            for (int r = 0; r < 16; r++)
            {
                const uint64_t b = fp_to_bits(x[r] + kPackMagic);
                lo[r] = (uint32_t)b;
                hi[r] = (uint32_t)(b >> 32) & 0xFFFFFu;
            }
#pragma unroll
            for (int k = 0; k < 8; k++)
                w[k] = (uint64_t)lo[2 * k] | ((uint64_t)lo[2 * k + 1] << 32);
#pragma unroll
            for (int k = 0; k < 4; k++)
            {
                const uint32_t d0 = (hi[4 * k] & 0xFFFFu) | (hi[4 * k + 1] << 16), d1 = (hi[4 * k + 2] & 0xFFFFu) | (hi[4 * k + 3] << 16);
                w[8 + k] = (uint64_t)d0 | ((uint64_t)d1 << 32);
            }
            uint32_t t0 = 0, t1 = 0;
#pragma unroll
            for (int r = 0; r < 8; r++)
            {
                t0 |= (hi[r] >> 16) << (4 * r);
                t1 |= (hi[r + 8] >> 16) << (4 * r);
            }
            w[12] = (uint64_t)t0 | ((uint64_t)t1 << 32);
        }
        __device__ __forceinline__ void unpack52(const uint64_t (&w)[13], double (&x)[16])
        {
#pragma unroll
            for (int r = 0; r < 16; r++)
            {
                const uint32_t lo = (uint32_t)(w[r >> 1] >> (32 * (r & 1)));
                const uint32_t mid = (uint32_t)(w[8 + (r >> 2)] >> (16 * (r & 3))) & 0xFFFFu;
                const uint32_t top = (uint32_t)(w[12] >> (4 * r)) & 0xFu;
                const uint64_t b = (uint64_t)lo | ((uint64_t)(0x43300000u | (top << 16) | mid) << 32);
                x[r] = fp_from_bits(b) - kPackMagic;
            }
        }

        // modulus class of the integer back end (field.h, IntBounds): 0 tight (2^58 <= q < 2^60), 1 roomy (q < 2^58), 2 wide
        // (q >= 2^60: SEAL's 61-bit internal moduli, the BEHZ auxiliary base, keep the reference's guarded butterflies); wave-uniform
        __device__ __forceinline__ int int_class(const NttTables &t, unsigned prime)
        {
            const uint64_t q = SHL_UCONST(reinterpret_cast<const uint64_t *>(&t.mods[prime]))[0];
            return (q >> 60) ? 2 : (q >> 58) ? 0 : 1;
        }
        // runs body(std::integral_constant<int, ICLS>) for the class of `prime`: one wave-uniform branch per kernel
        template <class Body>
        __device__ __forceinline__ void with_int_class(const NttTables &t, unsigned prime, Body body)
        {
            const int c = int_class(t, prime);
            if (c == 0)
                body(std::integral_constant<int, 0>());
            else if (c == 1)
                body(std::integral_constant<int, 1>());
            else
                body(std::integral_constant<int, 2>());
        }
        // Which (tile, component, outer item) a workgroup of the two-pass kernels works on.  The hardware hands consecutive
        // workgroups to consecutive XCDs (workgroup i -> XCD i mod 8).  With the tile index fastest, the eight XCDs stream eight
        // NEIGHBOURING 32 KiB tiles at the same time - eight streams 32 KiB apart - and that is the one arrangement the memory
        // system dislikes: a plain copy whose XCDs take runs of 8 / 16 / 32 KiB moves 6.0 / 5.9 / 5.5 TB/s where 4 KiB
        // interleave, 256 KiB runs or one eighth of the buffer per XCD all move 6.3 (round 3,
        // tools/microbench/copy_locality.hip, profiles/r03_microbench_copy_locality.txt).  So the XCDs are given different
        // TRANSFORMS (N * 8 bytes apart) and each walks the tiles of its own: workgroup L -> XCD x = L mod 8, r = L / 8,
        // tile = r mod TILES, slot = (r / TILES) * 8 + x, (component, outer) = slot; the last, incomplete group of 8 * TILES keeps
        // the plain order.  MEASURED AND NOT KEPT (default 0): the 2^16 NTT leg 2.51 -> 2.49 TB/s, the headline step -0.6 %, BFV
        // configs[3] -1 % (gpurun_out r3i, same box, alternating): the passes' workgroups are persistent loops that drift apart,
        // so their concurrent streams are not the lock-step runs of the copy; what the copy shows is that the ceiling of a
        // tile-shaped pass is ~5.4 TB/s, not the 6.3 of a 4 KiB-per-workgroup stream.
#ifndef SEALHIP_XCD_SPREAD
#define SEALHIP_XCD_SPREAD 0
#endif
        struct Blk
        {
            unsigned tile, y, z;
        };
        __device__ __forceinline__ Blk spread_blocks()
        {
#if SEALHIP_XCD_SPREAD
            const unsigned tiles = gridDim.x, slots = gridDim.y * gridDim.z;
            const unsigned L = blockIdx.x + tiles * (blockIdx.y + gridDim.y * blockIdx.z);
            const unsigned group = 8 * tiles, full = (slots / 8) * group;
            unsigned tile, slot;
            if (L < full)
            {
                const unsigned x = L & 7, r = L >> 3;
                tile = r % tiles;
                slot = (r / tiles) * 8 + x;
            }
            else
            {
                tile = L % tiles;
                slot = L / tiles;
            }
            return Blk{ tile, slot % gridDim.y, slot / gridDim.y };
#else
            return Blk{ blockIdx.x, blockIdx.y, blockIdx.z };
#endif
        }

        template <bool FP>
        __device__ __forceinline__ const typename Field<FP>::tw_t *tw_table(const NttTables &t, bool inverse, unsigned prime)
        {
            if constexpr (FP)
                return (inverse ? t.inv_d : t.fwd_d) + ((size_t)prime << t.log_n);
            else
                return (inverse ? t.inv : t.fwd) + ((size_t)prime << t.log_n);
        }

        template <int ICLS>
        __device__ __forceinline__ void int_fix_all(uint64_t (&x)[16], const Field<false>::Mod &m)
        {
#pragma unroll
            for (int a = 0; a < 16; a++)
                Field<false>::fix4<IntBounds<ICLS>::hi32>(x[a], m);
        }

        // One radix-2 stage over the 16 registers of a thread, pairing register-index bit BIT.
        // tw(g) supplies the twiddle of group g = e >> (BIT+1).
        // Integer back end: ICLS = modulus class, B = bound of every value before the stage in units of q (compile time); classes
        // 0 / 1 run the unguarded 15-instruction butterfly (+ 4 q per stage) and bring all sixteen values back under 4 q first when
        // the stage could pass the class's limit; class 2 runs the reference's butterfly with its per-butterfly range guard.
        template <bool FP, int BIT, int ICLS = 0, int B = 4, class TwFn>
        __device__ __forceinline__ void stage_fwd(typename Field<FP>::elem (&x)[16], const typename Field<FP>::Mod &m, TwFn tw)
        {
            if constexpr (!FP)
            {
                if constexpr (IntBounds<ICLS>::fwd_fix_before(B))
                    int_fix_all<ICLS>(x, m);
            }
#pragma unroll
            for (int g = 0; g < (8 >> BIT); g++)
            {
                const auto w = tw(g);
#pragma unroll
                for (int k = 0; k < (1 << BIT); k++)
                {
                    const int e0 = (g << (BIT + 1)) | k;
                    if constexpr (!FP && ICLS == 2)
                        Field<FP>::bfly_fwd_guarded(x[e0], x[e0 | (1 << BIT)], w, m);
                    else
                        Field<FP>::bfly_fwd(x[e0], x[e0 | (1 << BIT)], w, m);
                }
            }
        }

        // R (<= 4) consecutive stages on register bits 3, 2, ...; twiddle of (stage t, group g) = tw(t, g).  Integer back end:
        // values enter below B q and leave below IntBounds<ICLS>::fwd_after(B, R) q.
        template <bool FP, int R, int ICLS = 0, int B = 4, class TwFn>
        __device__ __forceinline__ void phase_fwd(typename Field<FP>::elem (&x)[16], const typename Field<FP>::Mod &m, TwFn tw)
        {
            typedef IntBounds<ICLS> IB;
            if constexpr (R >= 1)
                stage_fwd<FP, 3, ICLS, B>(x, m, [&](int g) { return tw(0, g); });
            if constexpr (R >= 2)
                stage_fwd<FP, 2, ICLS, IB::fwd_after(B, 1)>(x, m, [&](int g) { return tw(1, g); });
            if constexpr (R >= 3)
                stage_fwd<FP, 1, ICLS, IB::fwd_after(B, 2)>(x, m, [&](int g) { return tw(2, g); });
            if constexpr (R >= 4)
                stage_fwd<FP, 0, ICLS, IB::fwd_after(B, 3)>(x, m, [&](int g) { return tw(3, g); });
        }

        // A phase and the reduction that ends it.  Double precision: fix() of all 16 values when FIX.  Integer back end: FIX is
        // ignored - the reductions are placed by the compile-time bound (stage_fwd), and the values leave below
        // IntBounds<ICLS>::fwd_after(B, R) q, which the caller hands to whatever consumes them.
        template <bool FP, int R, bool FIX, int ICLS = 0, int B = 4, class TwFn>
        __device__ __forceinline__ void phase_fwd_end(typename Field<FP>::elem (&x)[16], const typename Field<FP>::Mod &m, TwFn tw)
        {
            phase_fwd<FP, R, ICLS, B>(x, m, tw);
            if constexpr (FP && FIX)
            {
#pragma unroll
                for (int a = 0; a < 16; a++)
                    Field<FP>::fix(x[a], m);
            }
        }
        // integer back end: a forward result below B q -> [0, 4q) (the reference's lazy output range) / -> [0, q)
        template <bool FP, int ICLS, int B>
        __device__ __forceinline__ uint64_t fwd_out_lazy(typename Field<FP>::elem x, const typename Field<FP>::Mod &m)
        {
            if constexpr (!FP && B > 4)
                Field<FP>::template fix4<IntBounds<ICLS>::hi32>(x, m);
            return Field<FP>::fwd_to_lazy(x, m);
        }
        template <bool FP, int ICLS, int B>
        __device__ __forceinline__ uint64_t fwd_out_canon(typename Field<FP>::elem x, const typename Field<FP>::Mod &m)
        {
            if constexpr (!FP && ICLS != 2)
                return Field<FP>::template canon_any<IntBounds<ICLS>::hi32>(x, m);
            else
                return Field<FP>::fwd_to_canon(x, m);
        }
        // bound (units of q) of the integer back end's values after pass 1 / after both passes of a forward transform whose
        // input is below 4 q
        template <int ICLS, int D1>
        constexpr int kP1Out = IntBounds<ICLS>::fwd_after(4, D1);
        template <int ICLS, int D1>
        constexpr int kP2Out = IntBounds<ICLS>::fwd_after(kP1Out<ICLS, D1>, 8);

        // the same with one fix() of all 16 values after the first FIXAT stages of the phase (double-precision back end: the
        // "lean" placement of the key-switch kernels, see p1_tile)
        template <bool FP, int R, int FIXAT, class TwFn>
        __device__ __forceinline__ void phase_fwd_fix(typename Field<FP>::elem (&x)[16], const typename Field<FP>::Mod &m, TwFn tw)
        {
            static_assert(FP && R == 4 && FIXAT >= 1 && FIXAT <= 3, "a four-stage double-precision phase with the fix inside it");
            stage_fwd<FP, 3>(x, m, [&](int g) { return tw(0, g); });
            if constexpr (FIXAT == 1)
            {
#pragma unroll
                for (int a = 0; a < 16; a++)
                    Field<FP>::fix(x[a], m);
            }
            stage_fwd<FP, 2>(x, m, [&](int g) { return tw(1, g); });
            if constexpr (FIXAT == 2)
            {
#pragma unroll
                for (int a = 0; a < 16; a++)
                    Field<FP>::fix(x[a], m);
            }
            stage_fwd<FP, 1>(x, m, [&](int g) { return tw(2, g); });
            if constexpr (FIXAT == 3)
            {
#pragma unroll
                for (int a = 0; a < 16; a++)
                    Field<FP>::fix(x[a], m);
            }
            stage_fwd<FP, 0>(x, m, [&](int g) { return tw(3, g); });
        }

        // Inverse (Gentleman-Sande) counterparts: the stages of a phase are undone last-to-first.
        // Integer back end (classes 0 / 1): IDX = number of stages of this phase already undone, FB = register bit of the
        // phase's first stage, EIN = exponent every register entered the phase with (values below 2^EIN q); the exponent of a
        // register before this stage is IntBounds::inv_exp() - the two operands of a butterfly share it -, operands at the limit
        // are brought under 4 q first, the difference is offset by 2^E q.
        template <bool FP, int BIT, int ICLS = 2, int IDX = 0, int FB = 0, int EIN = 0, class TwFn>
        __device__ __forceinline__ void stage_inv(typename Field<FP>::elem (&x)[16], const typename Field<FP>::Mod &m, TwFn tw)
        {
#pragma unroll
            for (int g = 0; g < (8 >> BIT); g++)
            {
                const auto w = tw(g);
#pragma unroll
                for (int k = 0; k < (1 << BIT); k++)
                {
                    const int e0 = (g << (BIT + 1)) | k, e1 = e0 | (1 << BIT);
                    if constexpr (!FP && ICLS != 2)
                    {
                        typedef IntBounds<ICLS> IB;
                        int E = IB::inv_exp(e0, IDX, FB, EIN);
                        if (E + 1 > IB::lim_exp)
                        {
                            Field<FP>::template fix4<IB::hi32>(x[e0], m);
                            Field<FP>::template fix4<IB::hi32>(x[e1], m);
                            E = 2;
                        }
                        Field<FP>::bfly_inv_lazy(x[e0], x[e1], w, m.q << E, m);
                    }
                    else
                        Field<FP>::bfly_inv(x[e0], x[e1], w, m);
                }
            }
        }
        // exponent the integer back end's registers leave an inverse phase with: R stages from exponent EIN, first register bit FB,
        // everything above IntBounds::inv_phase_out fixed at the end of the phase
        template <int ICLS>
        constexpr int inv_phase_exp(int ein, int stages, int fb)
        {
            if (ICLS == 2)
                return 1; // the guarded butterflies keep [0, 2q)
            int mx = 0;
            for (int r = 0; r < 16; r++)
            {
                const int e = IntBounds<ICLS>::inv_exp(r, stages, fb, ein);
                mx = e > mx ? e : mx;
            }
            return mx > IntBounds<ICLS>::inv_phase_out ? IntBounds<ICLS>::inv_phase_out : mx;
        }
        // undo stages t = R-1 .. FIRST of a phase (FIRST = 1 leaves stage 0 to the caller, together with the reduction it needs).
        // Integer back end, FIRST = 0: the registers leave below 2^inv_phase_exp<ICLS>(EIN, R, 4 - R) q.
        template <bool FP, int R, int FIRST, int ICLS = 2, int EIN = 0, class TwFn>
        __device__ __forceinline__ void phase_inv(typename Field<FP>::elem (&x)[16], const typename Field<FP>::Mod &m, TwFn tw)
        {
            constexpr int FB = 4 - R;
            if constexpr (R >= 4 && FIRST <= 3)
                stage_inv<FP, 0, ICLS, R - 4, FB, EIN>(x, m, [&](int g) { return tw(3, g); });
            if constexpr (R >= 3 && FIRST <= 2)
                stage_inv<FP, 1, ICLS, R - 3, FB, EIN>(x, m, [&](int g) { return tw(2, g); });
            if constexpr (R >= 2 && FIRST <= 1)
                stage_inv<FP, 2, ICLS, R - 2, FB, EIN>(x, m, [&](int g) { return tw(1, g); });
            if constexpr (R >= 1 && FIRST <= 0)
                stage_inv<FP, 3, ICLS, R - 1, FB, EIN>(x, m, [&](int g) { return tw(0, g); });
            if constexpr (!FP && ICLS != 2 && FIRST == 0)
            {
                typedef IntBounds<ICLS> IB;
#pragma unroll
                for (int r = 0; r < 16; r++)
                    if (IB::inv_exp(r, R, FB, EIN) > IB::inv_phase_out)
                        Field<FP>::template fix4<IB::hi32>(x[r], m);
            }
        }
        // the last inverse stage (register bit 3, N^-1 folded in) after a phase_inv<.., R, 1, ICLS, EIN>: operands at the limit are
        // fixed first; results in [0, 2q) (exact quotients)
        template <bool FP, int R, int ICLS, int EIN>
        __device__ __forceinline__ void inv_last_stage(typename Field<FP>::elem (&x)[16], const typename Field<FP>::Mod &m,
                                                       const typename Field<FP>::tw_t &ni, const typename Field<FP>::tw_t &nw)
        {
#pragma unroll
            for (int k = 0; k < 8; k++)
            {
                if constexpr (!FP && ICLS != 2)
                {
                    typedef IntBounds<ICLS> IB;
                    int E = IB::inv_exp(k, R - 1, 4 - R, EIN);
                    if (E + 1 > IB::lim_exp)
                    {
                        Field<FP>::template fix4<IB::hi32>(x[k], m);
                        Field<FP>::template fix4<IB::hi32>(x[k | 8], m);
                        E = 2;
                    }
                    Field<FP>::bfly_inv_last(x[k], x[k | 8], ni, nw, m.q << E, m);
                }
                else
                    Field<FP>::bfly_inv_last(x[k], x[k | 8], ni, nw, m);
            }
        }

        // The 15 twiddles of a 4-stage phase held in registers: slot (1<<t)+g.
        template <bool FP>
        struct TwRegs;
        template <>
        struct TwRegs<true>
        {
            double w[16];
            __device__ __forceinline__ void set(int slot, double v) { w[slot] = v; }
            __device__ __forceinline__ double get(int slot) const { return w[slot]; }
        };
        template <>
        struct TwRegs<false>
        {
            uint64_t w[16], wq[16];
            __device__ __forceinline__ void set(int slot, const ShoupOp &v)
            {
                w[slot] = v.w;
                wq[slot] = v.wq;
            }
            __device__ __forceinline__ ShoupOp get(int slot) const { return ShoupOp{ w[slot], wq[slot] }; }
        };

        // Load the twiddles of R stages whose table rows start at base(t) = first index of stage t
        // for this thread; stage t needs 2^t consecutive entries.
        template <bool FP, int R, class BaseFn>
        __device__ __forceinline__ void load_tw(TwRegs<FP> &r, const typename Field<FP>::tw_t *tab, BaseFn base)
        {
#pragma unroll
            for (int t = 0; t < R; t++)
            {
                const unsigned b = base(t);
#pragma unroll
                for (int g = 0; g < (1 << t); g++)
                    r.set((1 << t) + g, tab[b + g]);
            }
        }

        // ---------------------------------------------------------------------------------------
        // source mapping of pass 1 (NttBatch::src_mode)
        // ---------------------------------------------------------------------------------------
        struct SrcMap
        {
            int mode; // 0 own residue, 1 foreign residue (x mod q), 2 ((x + half) mod src_q) mod q + fix, 3 x mod q + fix
            uint64_t half, src_q, fix;
        };
        template <bool FP>
        __device__ __forceinline__ typename Field<FP>::elem map_src(uint64_t v, const SrcMap &s, const typename Field<FP>::Mod &m)
        {
            typedef Field<FP> F;
            if (s.mode == 0)
                return F::from_canon(v, m);
            if (s.mode == 1)
                return F::from_any(v, m);
            const uint64_t r = s.mode == 3 ? v : csub(v + s.half, s.src_q); // 3: the producer has added `half` (NttBatch::out_add)
            typename F::elem x = F::from_any(r, m) + F::from_canon(s.fix, m);
            F::fix(x, m); // the sum may reach 1.5q: bring it back before four butterfly stages
            return x;
        }

        template <bool FP, int D1>
        __device__ __forceinline__ void p1_load_tw(TwRegs<FP> &tw, const typename Field<FP>::tw_t *tab, unsigned tid)
        {
            const unsigned hi = tid >> Geo<D1>::LC;
            load_tw<FP, 4>(tw, tab, [&](int t) { return (1u << (Geo<D1>::rA + t)) + (hi << t); });
        }

        // ---------------------------------------------------------------------------------------
        // pass 1 body: src (natural order, column tile cg) -> D1 stages -> mid (tile order)
        // ---------------------------------------------------------------------------------------
        // BS = words between consecutive 256-word blocks of the tile-order intermediate (256 in HBM; 272 when the
        // intermediate lives in LDS, so that the 16-lane runs of one wave instruction fall on different banks)
        // LEAN (double-precision back end, N = 2^16, key switching): with balanced twiddles a magnitude B q grows to
        // (1.1875 B + 0.5) q per stage (field.h), 0.5 -> 1.09 -> 1.80 -> 2.64 -> 3.63 -> 4.81 -> 6.21 -> 7.88 over seven stages
        // (< 8 q <= 2^53: still exact).  Round 3: the fix() sit after global stage 6 (here, inside phase B) and after stage 13
        // (p2_tile), so that a digit may come in UNFIXED with |x| <= kLeanEntry q (a residue of a digit modulus of about the
        // target's size: 1.13 -> 1.84 -> 2.69 -> 3.69 -> 4.88 -> 6.30 -> 7.98, then 0.5 -> 1.09 -> 1.80): the intermediate
        // leaves with |x| <= 1.80 q, unfixed, and the values leave p2_tile with |x| <= 2.64 q.
        // Integer back end: ICLS = modulus class; the sixteen values enter below 4 q and leave below kP1Out<ICLS, D1> q.
        template <bool FP, int D1, int BS = 256, bool LEAN = false, int ICLS = 0, bool PACK = false>
        __device__ __forceinline__ void p1_tile(
            typename Field<FP>::elem (&x)[16], const typename Field<FP>::Mod &m, const typename Field<FP>::tw_t *tab,
            const TwRegs<FP> &tw, uint64_t *lds, uint64_t *mid_tr, unsigned cg, unsigned tid)
        {
            typedef Field<FP> F;
            typedef Geo<D1> G;
            const unsigned c = tid & (G::C - 1);
            const unsigned hi = tid >> G::LC; // rbl in phase A, ra in phase B
            if constexpr (G::rA > 0)
            {
                // phase A: register a = ra*2^(4-rA) + rbh; stage s pairs register bit 3-s; twiddle 2^s + group: uniform
                phase_fwd_end<FP, G::rA, !LEAN, ICLS, 4>(x, m, [&](int t, int g) { return ld_uniform(tab, (1u << t) + g); });
                __syncthreads(); // previous users of the exchange buffer are done
#pragma unroll
                for (int a = 0; a < 16; a++)
                {
                    const unsigned ra = a >> (4 - G::rA), rbh = a & ((1 << (4 - G::rA)) - 1);
                    const unsigned R = ra * 16 + (rbh << G::rA) + hi;
                    lds[R * G::CP + c] = F::raw(x[a]);
                }
                __syncthreads();
#pragma unroll
                for (int rb = 0; rb < 16; rb++)
                    x[rb] = F::unraw(lds[(hi * 16 + rb) * G::CP + c]);
            }
            // phase B: thread (c, ra = hi); register rb; stage rA+t pairs rb bit 3-t; twiddle 2^(rA+t) + ra*2^t + group
            // (tw = p1_load_tw(), loop-invariant for callers that transform many tiles with one prime)
            if constexpr (FP && LEAN)
            {
                static_assert(!LEAN || G::rA == 4, "the lean placement is worked out for eight stages per pass");
                phase_fwd_fix<FP, 4, 2>(x, m, [&](int t, int g) { return tw.get((1 << t) + g); });
            }
            else
            {
                // the intermediate is stored with |x| <= q/2 resp. below kP1Out<ICLS, D1> q
                phase_fwd_end<FP, 4, true, ICLS, IntBounds<ICLS>::fwd_after(4, G::rA)>(x, m, [&](int t, int g) { return tw.get((1 << t) + g); });
            }
            if constexpr (PACK)
            {
                static_assert(!PACK || (FP && D1 == 8 && BS == 256), "the packed intermediate is defined for the plain double-precision pass at N = 2^16");
                // packed intermediate (above): this thread's sixteen rows of column c of block (hg = hi, cg)
                uint64_t w[13];
                pack52(x, w);
                uint64_t *o = mid_tr + (size_t)(hi * 16 + cg) * kPackBlock + c;
#ifdef SEALHIP_P1_NOSTORE
                uint64_t sink = 0; // measurement build: one word per thread keeps the arithmetic alive
#pragma unroll
                for (int k = 0; k < 13; k++)
                    sink ^= w[k];
                mid_st<1>(o, sink);
#else
#if SEALHIP_PACK_VEC16
                uint64_t *o2 = o + c; // column c of the block: pair j at j*32 + c*2
#pragma unroll
                for (int j = 0; j < 6; j++)
                    mid_st2<1>(o2 + j * 32, w[2 * j], w[2 * j + 1]);
                mid_st<1>(o + 192, w[12]);
#else
#pragma unroll
                for (int k = 0; k < 13; k++)
                    mid_st<1>(o + k * 16, w[k]);
#endif
#endif
                return;
            }
            // tile order: ((hg*16 + col_hi)*16 + h_lo)*16 + col_lo, hg = ra, h_lo = rb, col = cg*C + c
            const unsigned col = cg * G::C + c;
#ifdef SEALHIP_MID_WAVE_MAJOR
            if constexpr (BS == 256)
            {
                // row hi*16 + rb of column col: wave rb >> 2 of tile hi, row-in-wave rb & 3, column block col >> 4
                uint64_t *w = mid_tr + (size_t)hi * 4096 + (col >> 4) * 64 + (col & 15);
#pragma unroll
                for (int rb = 0; rb < 16; rb++)
                    w[(rb >> 2) * 1024 + (rb & 3) * 16] = F::raw(x[rb]);
                return;
            }
#endif
            uint64_t *o = mid_tr + (size_t)(hi * 16 + (col >> 4)) * BS + (col & 15);
            if constexpr (BS == 256 && (SEALHIP_KS_NT & 1) != 0)
            {
#pragma unroll
                for (int rb = 0; rb < 16; rb++)
                    mid_st<1>(o + rb * 16, F::raw(x[rb]));
                return;
            }
#ifdef SEALHIP_KS_NOMEM
            // measurement build (tools/ab.sh nomem): the tile is not stored - one word per thread keeps the arithmetic alive
            uint64_t sink = 0;
#pragma unroll
            for (int rb = 0; rb < 16; rb++)
                sink ^= F::raw(x[rb]);
            o[0] = sink;
#else
#pragma unroll
            for (int rb = 0; rb < 16; rb++)
                o[rb * 16] = F::raw(x[rb]);
#endif
        }

        // ---------------------------------------------------------------------------------------
        // pass 2 body: 16 registers loaded from mid (tile order) -> 8 stages -> 16 contiguous
        // coefficients (col = 16 v' + e') of row h = 16 hg + u in registers.  lds = this wave's rows.
        // TW_LDS: twiddles of both phases are taken from LDS tables staged by the caller:
        //   twa[t][u][g] (t<4, g<2^t) at twa[(16 << t) - 16 + (u << t) + g]
        //   twb[t][g][tid]            at twb[((256 << t) - 256) + g*256 + tid]
        // ---------------------------------------------------------------------------------------
        // stage the 240 row-shared twiddles of pass 2's phase A (row tile hg) at twa[(16 << t) - 16 + (u << t) + g]
        template <int D1, class TW>
        __device__ __forceinline__ void stage_twa(TW *twa, const TW *tab, unsigned hg, unsigned tid)
        {
            if (tid < 240)
            {
                const unsigned t = 31 - __builtin_clz(tid / 16 + 1);
                const unsigned r = tid - ((16u << t) - 16u);
                twa[tid] = tab[(1u << (D1 + t)) + ((hg * 16) << t) + r];
            }
        }

        // twiddles of the two phases of pass 2 for tile hg: they depend on (prime, hg, thread) only, so a
        // workgroup that transforms the same tile of many polynomials can load them once
        template <bool FP, int D1>
        __device__ __forceinline__ void p2_load_tw(TwRegs<FP> &ta, TwRegs<FP> &tb, const typename Field<FP>::tw_t *tab, unsigned hg, unsigned tid)
        {
            const unsigned v = tid & 15, u = tid >> 4;
            const unsigned h = hg * 16 + u;
            load_tw<FP, 4>(ta, tab, [&](int t) { return (1u << (D1 + t)) + (h << t); });
            load_tw<FP, 4>(tb, tab, [&](int t) { return (1u << (D1 + 4 + t)) + ((h * 16 + v) << t); });
        }

        // TWA_LDS: only phase A's row-shared twiddles come from LDS (twa), phase B's per-thread ones from global memory
        // (LOWREG) or from the caller's registers (HOIST && TWA_LDS: pre_b only)
        // LEAN (see p1_tile): the input arrives with |x| <= 1.80 q; no fix() after phase A (-> 6.21 q), one after the first stage
        // of phase B (global stage 13: 7.88 q -> q/2), none at the end: the values leave with |x| <= 2.64 q, which the key
        // products take (|x k mod q| <= q (1/2 + 3/16 * 2.64) = 0.995 q with balanced key words; seven terms on top of a fixed
        // accumulator stay below 7.5 q)
        // Integer back end: ICLS = modulus class, BIN = bound of the loaded values in units of q (kP1Out<ICLS, D1> for an
        // intermediate written by p1_tile); they leave below IntBounds<ICLS>::fwd_after(BIN, 8) q.
        // TWB3_LDS (integer ks2, round 3): the eight per-thread twiddles of the LAST stage are read from twb[g * 256 + tid] (staged
        // once per workgroup by the caller: they are the same for every digit and batch item), the seven of the stages before it
        // from global memory where they are used
        template <bool FP, int D1, bool TW_LDS, bool LOWREG = false, bool HOIST = false, bool TWA_LDS = false, bool LEAN = false, int ICLS = 0,
                  int BIN = 4, bool TWB3_LDS = false>
        __device__ __forceinline__ void p2_tile(
            typename Field<FP>::elem (&x)[16], const typename Field<FP>::Mod &m, const typename Field<FP>::tw_t *tab,
            const typename Field<FP>::tw_t *twa, const typename Field<FP>::tw_t *twb, uint64_t *lds_wave, unsigned hg, unsigned tid,
            const TwRegs<FP> *pre_a = nullptr, const TwRegs<FP> *pre_b = nullptr)
        {
            typedef Field<FP> F;
            const unsigned v = tid & 15, u = tid >> 4;
            const unsigned h = hg * 16 + u;
            const unsigned ul = u & 3; // row inside this wave's buffer
            if constexpr (HOIST && !TWA_LDS)
            {
                phase_fwd_end<FP, 4, !LEAN, ICLS, BIN>(x, m, [&](int t, int g) { return pre_a->get((1 << t) + g); });
            }
            else if constexpr ((LOWREG || HOIST) && (TW_LDS || TWA_LDS))
            {
                phase_fwd_end<FP, 4, !LEAN, ICLS, BIN>(x, m, [&](int t, int g) { return twa[(16u << t) - 16u + (u << t) + g]; });
            }
            else if constexpr (LOWREG)
            {
                // register-lean variant: each twiddle is fetched where it is used
                phase_fwd_end<FP, 4, !LEAN, ICLS, BIN>(x, m, [&](int t, int g) { return tab[(1u << (D1 + t)) + (h << t) + g]; });
            }
            else
            {
                TwRegs<FP> tw;
                if constexpr (TW_LDS)
                    load_tw<FP, 4>(tw, twa, [&](int t) { return (16u << t) - 16u + (u << t); });
                else
                    load_tw<FP, 4>(tw, tab, [&](int t) { return (1u << (D1 + t)) + (h << t); });
                phase_fwd_end<FP, 4, !LEAN, ICLS, BIN>(x, m, [&](int t, int g) { return tw.get((1 << t) + g); });
            }
            // wave-local exchange: (e, v) -> (v', e')
#pragma unroll
            for (int e = 0; e < 16; e++)
                lds_wave[ul * kRowWords + e * 18 + v] = F::raw(x[e]);
            __builtin_amdgcn_wave_barrier();
            {
                const ulonglong2 *rp = reinterpret_cast<const ulonglong2 *>(lds_wave + ul * kRowWords + v * 18);
#pragma unroll
                for (int e = 0; e < 8; e++)
                {
                    const ulonglong2 pr = rp[e];
                    x[2 * e] = F::unraw(pr.x);
                    x[2 * e + 1] = F::unraw(pr.y);
                }
            }
            __builtin_amdgcn_wave_barrier(); // the buffer may be rewritten by the caller's next tile
            constexpr int BMID = IntBounds<ICLS>::fwd_after(BIN, 4);
            if constexpr (HOIST)
            {
                phase_fwd_end<FP, 4, !LEAN, ICLS, BMID>(x, m, [&](int t, int g) { return pre_b->get((1 << t) + g); });
            }
            else if constexpr (LOWREG && TW_LDS)
            {
                auto twf = [&](int t, int g) { return twb[((256u << t) - 256u) + g * 256 + tid]; };
                if constexpr (LEAN)
                    phase_fwd_fix<FP, 4, 1>(x, m, twf);
                else
                    phase_fwd_end<FP, 4, true, ICLS, BMID>(x, m, twf);
            }
            else if constexpr (LOWREG && TWB3_LDS)
            {
                phase_fwd_end<FP, 4, !LEAN, ICLS, BMID>(x, m, [&](int t, int g) {
                    return t == 3 ? twb[g * 256 + tid] : tab[(1u << (D1 + 4 + t)) + ((h * 16 + v) << t) + g];
                });
            }
            else if constexpr (LOWREG)
            {
                phase_fwd_end<FP, 4, !LEAN, ICLS, BMID>(x, m, [&](int t, int g) { return tab[(1u << (D1 + 4 + t)) + ((h * 16 + v) << t) + g]; });
            }
            else
            {
                TwRegs<FP> tw;
                if constexpr (TW_LDS)
                {
#pragma unroll
                    for (int t = 0; t < 4; t++)
#pragma unroll
                        for (int g = 0; g < (1 << t); g++)
                            tw.set((1 << t) + g, twb[((256u << t) - 256u) + g * 256 + tid]);
                }
                else
                    load_tw<FP, 4>(tw, tab, [&](int t) { return (1u << (D1 + 4 + t)) + ((h * 16 + v) << t); });
                phase_fwd_end<FP, 4, !LEAN, ICLS, BMID>(x, m, [&](int t, int g) { return tw.get((1 << t) + g); });
            }
            static_assert(!LEAN || (FP && LOWREG && TW_LDS), "the lean placement is wired for the double-precision ks2 variant only");
        }

        // registers (row u, cols 16 v' + e') -> wave-local transpose -> 16 coalesced 512-byte stores
        // of this wave's 4 rows to natural order at `rows` (= address of row 16*hg + 4*wave).
        __device__ __forceinline__ void store_rows(const uint64_t (&val)[16], uint64_t *lds_wave, uint64_t *rows, unsigned tid)
        {
            const unsigned v = tid & 15, ul = (tid >> 4) & 3, lane = tid & 63;
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int e = 0; e < 16; e++)
                lds_wave[ul * kRowWords + v * 18 + e] = val[e];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < 16; k++)
            {
                const unsigned row = k >> 2, col = (k & 3) * 64 + lane;
                mid_st<16>(rows + row * 256 + col, lds_wave[row * kRowWords + col + 2 * (col >> 4)]);
            }
            __builtin_amdgcn_wave_barrier(); // a looping caller's next tile rewrites the buffer (program order in hardware; the emulator's lanes need it said)
        }
        // the same transposition, handing each coalesced (offset, value) pair to `sink`
        template <class Sink>
        __device__ __forceinline__ void emit_rows(const uint64_t (&val)[16], uint64_t *lds_wave, unsigned tid, Sink sink)
        {
            const unsigned v = tid & 15, ul = (tid >> 4) & 3, lane = tid & 63;
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int e = 0; e < 16; e++)
                lds_wave[ul * kRowWords + v * 18 + e] = val[e];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < 16; k++)
            {
                const unsigned row = k >> 2, col = (k & 3) * 64 + lane;
                sink(row * 256 + col, lds_wave[row * kRowWords + col + 2 * (col >> 4)]);
            }
            __builtin_amdgcn_wave_barrier();
        }
        // emit_rows with the index k of the (offset, value) pair: k-th pair = offset (k >> 2) * 256 + (k & 3) * 64 + lane, so that a
        // caller can have loaded its other operands of these offsets ahead of time
        template <class Sink>
        __device__ __forceinline__ void emit_rows_k(const uint64_t (&val)[16], uint64_t *lds_wave, unsigned tid, Sink sink)
        {
            const unsigned v = tid & 15, ul = (tid >> 4) & 3, lane = tid & 63;
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int e = 0; e < 16; e++)
                lds_wave[ul * kRowWords + v * 18 + e] = val[e];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < 16; k++)
            {
                const unsigned row = k >> 2, col = (k & 3) * 64 + lane;
                sink(k, row * 256 + col, lds_wave[row * kRowWords + col + 2 * (col >> 4)]);
            }
            __builtin_amdgcn_wave_barrier();
        }
        __device__ __forceinline__ void load_rows(uint64_t (&val)[16], uint64_t *lds_wave, const uint64_t *rows, unsigned tid)
        {
            const unsigned v = tid & 15, ul = (tid >> 4) & 3, lane = tid & 63;
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < 16; k++)
            {
                const unsigned row = k >> 2, col = (k & 3) * 64 + lane;
                lds_wave[row * kRowWords + col + 2 * (col >> 4)] = mid_ld<16>(rows + row * 256 + col);
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int e = 0; e < 16; e++)
                val[e] = lds_wave[ul * kRowWords + v * 18 + e];
            __builtin_amdgcn_wave_barrier();
        }

        // NTT-domain automorphism as an index map (galois.cpp:18-51; poly_kernels.hip: galois_ntt_kernel): result[j] = operand[T(j)],
        // T(j) = bitrev_n(((elt * bitrev_{n+1}(j + N)) >> 1) & (N - 1)).  The low bits of j are the high bits of the reversed word, a
        // product with an odd element only carries upwards, so an aligned block of 2^m positions maps onto an aligned block of 2^m
        // positions: a gather through T touches the lines a plain read of the same block would.
        __device__ __forceinline__ unsigned galois_src_index(unsigned j, uint32_t elt, unsigned n_log)
        {
            const unsigned N = 1u << n_log;
            const unsigned rev = __brev(j + N) >> (32 - (n_log + 1));
            const unsigned idx = (unsigned)(((uint64_t)elt * rev) >> 1) & (N - 1);
            return __brev(idx) >> (32 - n_log);
        }
        // load_rows through the automorphism: `poly` = the polynomial's first word, j0 = natural index of this wave's first row
        __device__ __forceinline__ void load_rows_galois(uint64_t (&val)[16], uint64_t *lds_wave, const uint64_t *poly, unsigned j0, uint32_t elt,
                                                         unsigned n_log, unsigned tid)
        {
            const unsigned v = tid & 15, ul = (tid >> 4) & 3, lane = tid & 63;
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < 16; k++)
            {
                const unsigned row = k >> 2, col = (k & 3) * 64 + lane;
                lds_wave[row * kRowWords + col + 2 * (col >> 4)] = mid_ld<16>(poly + galois_src_index(j0 + row * 256 + col, elt, n_log));
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int e = 0; e < 16; e++)
                val[e] = lds_wave[ul * kRowWords + v * 18 + e];
            __builtin_amdgcn_wave_barrier();
        }

        // ---------------------------------------------------------------------------------------
        // generic forward transform kernels
        // ---------------------------------------------------------------------------------------
        struct FwdArgs
        {
            uint64_t *data;
            size_t outer_stride;
            uint64_t *mid; // [nouter][ncomp][N], tile order
            const uint64_t *src;
            size_t src_outer_stride;
            unsigned src_ncomp;
            int src_mode;
            uint64_t src_half, src_q;
            const uint64_t *src_fix;
            const uint32_t *comp_prime;
            int cls_hint; // NttBatch::cls_hint (host side only: picks the kernels)
            unsigned prime_first;
            unsigned ncomp;
            unsigned comp0; // this launch covers components [comp0, comp0 + gridDim.y)
            unsigned nouter;  // outer items; a workgroup handles y, y + gridDim.y, ... (twiddles stay in registers)
            int lazy;
            int epi;
            const uint64_t *epi_a;
            size_t epi_a_stride;
            const ShoupOp *epi_mul;
            uint64_t *epi_out0, *epi_out1;
            size_t epi_out_stride;
            int mid_pack; // the double-precision components' intermediate is packed (N = 2^16, plain transform: launch_fwd decides)
            uint64_t *ring;    // scratch of the one-launch ring kernel (ntt2_ring_words() words) or null: launch_fwd decides
            size_t ring_words;
            NttTables t;
        };
    } // namespace
} // namespace sealhip
