// Two arithmetic back ends for the residue field Z_q inside the gfx950 NTT / key-switch kernels.
//
// gfx950 issues every VALU instruction at the same rate (tools/microbench/valu_rates.hip: a
// v_mad_u64_u32, a 32-bit add and a v_fma_f64 all cost one ~4-cycle issue slot per wave64), so a
// kernel's speed is its instruction count.  A 64x64->128 product costs four v_mad_u64_u32 plus
// carries, which makes the reference's Shoup/Harvey butterfly (uintarithsmallmod.h:292-326,
// ntt.h:20-67) ~27 instructions per butterfly in 32-bit limbs.  For primes below 2^50 — every
// "scaling" prime of a typical CKKS chain — the same residue arithmetic can be done EXACTLY in
// double precision with error-free transformations at 8 instructions per butterfly:
//
//     h = fl(y*w); l = fma(y,w,-h)            y*w = h + l exactly (the product error is a double)
//     k = rint(fl(h*qinv))                    an integer near y*w/q
//     r = fma(-k,q,h) + l                     = y*w - k*q exactly: both steps are exact because
//                                               the results are integers of magnitude < 2^53
//
// r is congruent to y*w mod q whatever k was; only |r| depends on how good the quotient estimate
// is: |r| <= q*(1/2 + 3*2^-53*|y*w/q|) (three roundings: the product, 1/q, their product).  The
// multipliers - twiddles, N^-1 constants, key words - are stored as BALANCED representatives,
// |w| <= q/2 (Context, key_layout_kernel), so |y*w/q| <= |y|/2 and, with |y| <= B*q and q < 2^50,
//     |r| <= q*(0.5 + 0.1875*B)        (0.375*B for a multiplier in [0, q), e.g. 2^32 mod q in fp_from_u64).
// Values are kept as signed integers in doubles; sums stay exact while every magnitude is below
// 2^53 = 8*2^50, which the kernels guarantee by calling fix() (x - rint(x*qinv)*q, |result| <= q/2 + eps):
//     forward (CT):  B -> 1.1875*B + 0.5 per stage: 0.5 -> 1.09 -> 1.80 -> 2.64 -> 3.63 -> 4.81 -> 6.21 -> 7.88  (< 8)
//                    seven stages between two fix(); the transform kernels fix after every four-stage phase (3.63 q), the
//                    key-switch kernels of N = 2^16 after stages 7 and 14 only (ntt2_kernels.hip: p1_tile / p2_tile, LEAN)
//     inverse (GS):  B -> 2*B per stage:            0.5 -> 1 -> 2 -> 4 -> 8 (sum < 8q <= 2^53, exact)
// Only canonical residues in [0,q) ever leave a kernel, so results are bit-identical to the
// integer path and to the reference: the representation is internal.
//
// IntField is the general path (any q < 2^61).  Values between kernels and phases are in the reference's lazy
// ranges - forward [0,4q), inverse [0,2q) -; inside a forward phase moduli below 2^60 run unguarded up to 16 q
// (bfly_fwd / fwd_fix below), 61-bit moduli keep the reference's guard per butterfly.
#pragma once
#include "modarith.h"
#if defined(SEALHIP_CHECK_BOUNDS)
#include <cstdio>
#include <cstdlib>
#endif

namespace sealhip
{
    // q < 2^kFpMaxBits is eligible for the double-precision back end.
    constexpr int kFpMaxBits = 50;

    // Per-prime constants of the double-precision back end.
    struct __attribute__((aligned(16))) FpDesc
    {
        double q;      // the prime
        double qinv;   // fl(1/q)
        double two32;  // 2^32 mod q
        uint64_t qi;   // the prime as an integer (0 = prime not eligible)
    };

#if defined(SEALHIP_CHECK_BOUNDS) && !defined(__HIP_DEVICE_COMPILE__)
#define SEALHIP_BOUND(v)                                                                                        \
    do                                                                                                          \
    {                                                                                                           \
        const double sealhip_b_ = (v);                                                                          \
        if (!(sealhip_b_ < 9007199254740992.0 && sealhip_b_ > -9007199254740992.0))                             \
        {                                                                                                       \
            std::fprintf(stderr, "sealhip: magnitude bound 2^53 violated (%g) at %s:%d\n", sealhip_b_, __FILE__, __LINE__); \
            std::abort();                                                                                       \
        }                                                                                                       \
    } while (0)
// integer back end: a + b must not wrap (the unguarded forward butterflies rely on 16 q fitting a word)
#define SEALHIP_NOWRAP(a, b)                                                                                    \
    do                                                                                                          \
    {                                                                                                           \
        if ((uint64_t)(a) > ~(uint64_t)(b))                                                                     \
        {                                                                                                       \
            std::fprintf(stderr, "sealhip: 64-bit sum wraps at %s:%d\n", __FILE__, __LINE__);                   \
            std::abort();                                                                                       \
        }                                                                                                       \
    } while (0)
#else
#define SEALHIP_BOUND(v) ((void)0)
#define SEALHIP_NOWRAP(a, b) ((void)0)
#endif

    SHL_HD double fp_from_bits(uint64_t b)
    {
        return __builtin_bit_cast(double, b);
    }
    SHL_HD uint64_t fp_to_bits(double d)
    {
        return __builtin_bit_cast(uint64_t, d);
    }

    // x*w - k*q with k = rint(x*w/q): exact residue of the product, |result| <= q*(0.5 + 0.375*|x*w|/q^2)
    // (= 0.5 + 0.1875*|x|/q for a balanced multiplier |w| <= q/2).  SEALHIP_CHECK_BOUNDS (emulated build only): every
    // operand and result is checked against 2^53.
    SHL_HD double fp_mulmod(double x, double w, double q, double qinv)
    {
        SEALHIP_BOUND(x);
        double h = x * w;
        double l = __builtin_fma(x, w, -h);
        double k = __builtin_rint(h * qinv);
        double v = __builtin_fma(-k, q, h);
        SEALHIP_BOUND(v + l);
        return v + l;
    }
    // |x| < 2^53 -> congruent value with |result| <= q/2 (+ a few ulp of q)
    SHL_HD double fp_fix(double x, double q, double qinv)
    {
        double k = __builtin_rint(x * qinv);
        return __builtin_fma(-k, q, x);
    }
    // integer x < 2^52 -> the same value as a double (exponent trick: 2^52 + x has x in its mantissa)
    SHL_HD double fp_from_u52(uint64_t x)
    {
        return fp_from_bits(x | 0x4330000000000000ull) - 4503599627370496.0;
    }
    // any 64-bit x -> value congruent to x mod q with magnitude < q + 2^32
    SHL_HD double fp_from_u64(uint64_t x, const FpDesc &m)
    {
        double hi = (double)(uint32_t)(x >> 32), lo = (double)(uint32_t)x; // v_cvt_f64_u32: exact, one instruction each
        return fp_mulmod(hi, m.two32, m.q, m.qinv) + lo;
    }
    // |x| < q (integer-valued) -> canonical residue in [0,q) as an integer
    SHL_HD uint64_t fp_to_canon(double x, const FpDesc &m)
    {
        double y = x < 0.0 ? x + m.q : x;
        return fp_to_bits(y + 4503599627370496.0) & 0x000fffffffffffffull;
    }

    template <bool FP>
    struct Field;

    // per-prime constants of a wave-uniform prime through the scalar cache
    __device__ __forceinline__ ModDesc ld_uniform_mod(const ModDesc *p)
    {
        shl_uconst_ptr u = SHL_UCONST(reinterpret_cast<const uint64_t *>(p));
        return ModDesc{ u[0], u[1], u[2], u[3] };
    }
    __device__ __forceinline__ FpDesc ld_uniform_fpd(const FpDesc *p)
    {
        shl_uconst_ptr u = SHL_UCONST(reinterpret_cast<const uint64_t *>(p));
        return FpDesc{ __builtin_bit_cast(double, u[0]), __builtin_bit_cast(double, u[1]), __builtin_bit_cast(double, u[2]), u[3] };
    }
    // wave-uniform twiddle (the index is the same in every lane): scalar load.  The distinct type tells the integer back end
    // that the four words live in SGPRs (its instruction wrappers below need to know the register class of every operand).
    struct ShoupOpU : ShoupOp
    {};
    __device__ __forceinline__ ShoupOpU ld_uniform(const ShoupOp *tab, unsigned idx)
    {
        shl_uconst_ptr u = SHL_UCONST(reinterpret_cast<const uint64_t *>(tab));
        ShoupOpU r;
        r.w = u[2 * (size_t)idx];
        r.wq = u[2 * (size_t)idx + 1];
        return r;
    }
    __device__ __forceinline__ double ld_uniform(const double *tab, unsigned idx)
    {
        return __builtin_bit_cast(double, SHL_UCONST(reinterpret_cast<const uint64_t *>(tab))[idx]);
    }

    // ---- single gfx950 instructions for the integer back end
    //
    // hipcc's instruction selection works against 32-bit-limb arithmetic here: it rewrites every 32-bit `a * b + c` into
    // v_mad_u64_u32 with a freshly assembled {c, 0} register pair (two v_mov_b32 per product), and expands the high half of a
    // 64 x 64 product through the same pairs - a butterfly written as 15 operations came out as 24 instructions, a quarter of them
    // v_mov_b32 (round 3, llvm-objdump of ntt2_fwd_p2 / ks2_kernel).  The products are therefore issued through these wrappers:
    // one instruction each, opaque to the combiner, still scheduled and register-allocated by the compiler (not volatile, no
    // fixed registers).  SRC1 of the `_s` forms is an SGPR (wave-uniform twiddles and per-prime constants; gfx950 reads one SGPR
    // per VALU instruction).  The carry-out of v_mad_u64_u32 goes to a scratch SGPR pair nobody reads.
#if defined(__HIP_DEVICE_COMPILE__)
    namespace gfx
    {
#define SHL_GFX_PRODUCTS(SFX, C1)                                                                                          \
    __device__ __forceinline__ uint32_t mul_hi_##SFX(uint32_t a, uint32_t b)                                              \
    {                                                                                                                      \
        uint32_t r;                                                                                                        \
        asm("v_mul_hi_u32 %0, %1, %2" : "=v"(r) : "v"(a), C1(b));                                                          \
        return r;                                                                                                          \
    }                                                                                                                      \
    __device__ __forceinline__ uint32_t mul_lo_##SFX(uint32_t a, uint32_t b)                                              \
    {                                                                                                                      \
        uint32_t r;                                                                                                        \
        asm("v_mul_lo_u32 %0, %1, %2" : "=v"(r) : "v"(a), C1(b));                                                          \
        return r;                                                                                                          \
    }                                                                                                                      \
    /* a * b + c, all 64 bits */                                                                                           \
    __device__ __forceinline__ uint64_t mad64_##SFX(uint32_t a, uint32_t b, uint64_t c)                                   \
    {                                                                                                                      \
        uint64_t r, cy;                                                                                                    \
        asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(r), "=s"(cy) : "v"(a), C1(b), "v"(c));                               \
        return r;                                                                                                          \
    }                                                                                                                      \
    /* a * b, all 64 bits */                                                                                               \
    __device__ __forceinline__ uint64_t mul64_##SFX(uint32_t a, uint32_t b)                                               \
    {                                                                                                                      \
        uint64_t r, cy;                                                                                                    \
        asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(r), "=s"(cy) : "v"(a), C1(b));                                        \
        return r;                                                                                                          \
    }
        SHL_GFX_PRODUCTS(v, "v")
        SHL_GFX_PRODUCTS(s, "s")
#undef SHL_GFX_PRODUCTS
        // a + b in 32 bits, kept as one v_add_u32 on the high half of a register pair (the compiler would rather widen it to a
        // shifted 64-bit addition: a v_mov_b32 to build the pair plus a v_lshl_add_u64)
        __device__ __forceinline__ uint32_t add32(uint32_t a, uint32_t b)
        {
            uint32_t r;
            asm("v_add_u32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
            return r;
        }
        // 2 x + c with c in an SGPR pair: one v_lshl_add_u64 (the compiler factors 2 (x + c / 2) into two instructions when it
        // knows c is even)
        __device__ __forceinline__ uint64_t twice_plus(uint64_t x, uint64_t c)
        {
            uint64_t r;
            asm("v_lshl_add_u64 %0, %1, 1, %2" : "=v"(r) : "v"(x), "s"(c));
            return r;
        }
        // the register class of a twiddle's words: ShoupOpU -> SGPRs, ShoupOp -> VGPRs
        template <class TW>
        struct Tw
        {
            static __device__ __forceinline__ uint32_t mul_hi(uint32_t a, uint32_t b) { return mul_hi_v(a, b); }
            static __device__ __forceinline__ uint64_t mad64(uint32_t a, uint32_t b, uint64_t c) { return mad64_v(a, b, c); }
            static __device__ __forceinline__ uint64_t mul64(uint32_t a, uint32_t b) { return mul64_v(a, b); }
        };
        template <>
        struct Tw<ShoupOpU>
        {
            static __device__ __forceinline__ uint32_t mul_hi(uint32_t a, uint32_t b) { return mul_hi_s(a, b); }
            static __device__ __forceinline__ uint64_t mad64(uint32_t a, uint32_t b, uint64_t c) { return mad64_s(a, b, c); }
            static __device__ __forceinline__ uint64_t mul64(uint32_t a, uint32_t b) { return mul64_s(a, b); }
        };
    } // namespace gfx
#endif

    // ---- 64-bit integer back end (Shoup multiplication, lazy butterflies with compile-time range tracking)
    //
    // Round 3.  A butterfly's cost on gfx950 is its instruction count (every VALU instruction is one issue slot), and the
    // reference's Harvey butterfly (ntt.h:20-67) costs 20 (forward) / 26 (inverse) instructions in 32-bit limbs.  Three changes
    // bring that to 15 / 16 without changing a single residue:
    //  * the Shoup quotient h = floor(x * wq / 2^64) is only needed to within a few units - the remainder x*w - h*q is computed
    //    modulo 2^64 and a 64-bit word holds 16 q (q < 2^60) - so it is taken from three of the four 32 x 32 partial products
    //    (mul_hi_approx: 4 instructions instead of 8), which leaves the product in [0, 4q) instead of [0, 2q);
    //  * the sum X + t rides in the 64-bit addend of the first v_mad_u64_u32 of the remainder chain, and the difference is
    //    2X + 4q - (X + t): one v_lshl_add_u64 and one 64-bit subtraction;
    //  * no per-butterfly guard: values grow by 4 q per forward stage (they double per inverse stage) and the kernels place
    //    fix4() - six (generic) or five (q >= 2^40) instructions, result in [0, 4q) - where a COMPILE-TIME bound says the next
    //    stage could wrap (IntBounds below; ntt2_kernels.hip threads the bound through phases, passes and launches).
    // Moduli of 2^60 and above (SEAL's 61-bit internal moduli, the BEHZ auxiliary base; user primes have at most 60 bits,
    // defines.h:33-40) keep the reference's guarded butterflies with the exact quotient.
    template <>
    struct Field<false>
    {
        typedef uint64_t elem;
        typedef ShoupOp tw_t;
        struct Mod
        {
            uint64_t q, two_q, four_q;
            ModDesc md;
            // rcp = floor(2^(31 + bits) / q) in (2^31, 2^32], bits = bit_length(q): the 32-bit reciprocal of q and of 2q
            // sx = max(bits - 26, 0) aligns any value below 64 q to 32 bits; floor(x / 2q) ~ ((x >> sx) * rcp >> 32) >> (bits - sx),
            // floor(x / q) ~ the same shifted by one less; sh_hi = bits - 32: the same estimates from the high word alone (q >= 2^40)
            uint32_t rcp, sx, sh, sh_hi;
            uint32_t n0, n1; // the two halves of -q mod 2^64
            uint32_t m0, m1; // the two halves of -2q mod 2^64
        };
        static constexpr int tw_words = 2;

        static SHL_HD Mod make_mod(const ModDesc &md, const FpDesc &)
        {
            // floor(2^(31 + bits) / q) = floor(2^128 / q) >> (97 - bits); wave-uniform: scalar instructions on the device
            const unsigned bits = 64u - (unsigned)__builtin_clzll(md.q | 1);
            const unsigned s128 = 97u - bits; // 36 .. 95 for 2 <= bits <= 61
            const uint32_t rcp = s128 >= 64 ? (uint32_t)(md.ratio_hi >> (s128 - 64)) : (uint32_t)((md.ratio_hi << (64 - s128)) | (md.ratio_lo >> s128));
            const unsigned sx = bits > 26 ? bits - 26 : 0;
            const uint64_t nq = 0 - md.q, nq2 = 0 - md.two_q;
            return Mod{ md.q,           md.two_q,           md.two_q << 1,  md,
                        rcp,            sx,                 bits - sx,      bits > 32 ? bits - 32 : 0,
                        (uint32_t)nq,   (uint32_t)(nq >> 32), (uint32_t)nq2, (uint32_t)(nq2 >> 32) };
        }
        static SHL_HD elem from_canon(uint64_t x, const Mod &)
        {
            return x;
        }
        static SHL_HD elem from_any(uint64_t x, const Mod &m)
        {
            return barrett64(x, m.md);
        }
        // x*w - floor(x*wq / 2^64)*q in 32-bit limbs, written so that it compiles to 8 + 8 VALU
        // instructions: the quotient word h = hi64(x*wq), then the low 64 bits of x*w + h*(-q) as one
        // v_mad_u64_u32 chain for the low limbs and four v_mul_lo_u32 feeding two v_add3_u32 for the
        // high limb (no carry chain, no separate subtraction).  Result in [0,2q) for any 64-bit x.
        static SHL_HD uint64_t mul_lazy(uint64_t x, const tw_t &w, const Mod &m)
        {
            return mul_rem0(x, w, mul_hi64(x, w.wq), m);
        }
        // h in {floor(x * wq / 2^64) - 2, ..., floor(x * wq / 2^64)}: x wq / 2^64 = x1 wq1 + (x1 wq0 + x0 wq1) / 2^32 + x0 wq0 / 2^64; keep
        // x1 wq1 + hi32(x1 wq0) + hi32(x0 wq1), the dropped (lo32(x1 wq0) + lo32(x0 wq1)) 2^32 + x0 wq0 is below 3 * 2^64.
        // Five instructions: two v_mul_hi_u32, their 33-bit sum (v_add_co_u32 + v_addc_co_u32: a register pair that can be the addend
        // of) one v_mad_u64_u32.
        template <class TW>
        static SHL_HD uint64_t mul_hi_approx(uint64_t x, const TW &w)
        {
            const uint32_t x0 = (uint32_t)x, x1 = (uint32_t)(x >> 32), w0 = (uint32_t)w.wq, w1 = (uint32_t)(w.wq >> 32);
#if defined(__HIP_DEVICE_COMPILE__)
            const uint32_t a = gfx::Tw<TW>::mul_hi(x0, w1), c = gfx::Tw<TW>::mul_hi(x1, w0);
            return gfx::Tw<TW>::mad64(x1, w1, (uint64_t)a + c);
#else
            const uint32_t a = (uint32_t)(((uint64_t)x0 * w1) >> 32), c = (uint32_t)(((uint64_t)x1 * w0) >> 32);
            return ((uint64_t)x1 * w1 + a) + c;
#endif
        }
        // low 64 bits of add + x*w + h*(-q), seven instructions: a v_mad_u64_u32 chain carrying `add` for the low limbs, a second
        // chain of four whose low word alone is used (the four cross products of the high limb), one 32-bit addition
        template <class TW>
        static SHL_HD uint64_t mul_rem(uint64_t x, const TW &w, uint64_t h, uint64_t add, const Mod &m)
        {
            const uint32_t x0 = (uint32_t)x, x1 = (uint32_t)(x >> 32), w0 = (uint32_t)w.w, w1 = (uint32_t)(w.w >> 32);
            const uint32_t h0 = (uint32_t)h, h1 = (uint32_t)(h >> 32);
#if defined(__HIP_DEVICE_COMPILE__)
            uint64_t lo = gfx::Tw<TW>::mad64(x0, w0, add);
            lo = gfx::mad64_s(h0, m.n0, lo);
            uint64_t cr = gfx::Tw<TW>::mul64(x0, w1);
            cr = gfx::Tw<TW>::mad64(x1, w0, cr);
            cr = gfx::mad64_s(h0, m.n1, cr);
            cr = gfx::mad64_s(h1, m.n0, cr);
            const uint32_t hi = gfx::add32((uint32_t)(lo >> 32), (uint32_t)cr);
#else
            uint64_t lo = (uint64_t)x0 * w0 + add;
            lo += (uint64_t)h0 * m.n0;
            const uint32_t hi = (uint32_t)(lo >> 32) + x0 * w1 + x1 * w0 + h0 * m.n1 + h1 * m.n0;
#endif
            return ((uint64_t)hi << 32) | (uint32_t)lo;
        }
        // the same with add = 0 (the first product has no addend to carry)
        template <class TW>
        static SHL_HD uint64_t mul_rem0(uint64_t x, const TW &w, uint64_t h, const Mod &m)
        {
#if defined(__HIP_DEVICE_COMPILE__)
            const uint32_t x0 = (uint32_t)x, x1 = (uint32_t)(x >> 32), w0 = (uint32_t)w.w, w1 = (uint32_t)(w.w >> 32);
            const uint32_t h0 = (uint32_t)h, h1 = (uint32_t)(h >> 32);
            uint64_t lo = gfx::Tw<TW>::mul64(x0, w0);
            lo = gfx::mad64_s(h0, m.n0, lo);
            uint64_t cr = gfx::Tw<TW>::mul64(x0, w1);
            cr = gfx::Tw<TW>::mad64(x1, w0, cr);
            cr = gfx::mad64_s(h0, m.n1, cr);
            cr = gfx::mad64_s(h1, m.n0, cr);
            const uint32_t hi = gfx::add32((uint32_t)(lo >> 32), (uint32_t)cr);
            return ((uint64_t)hi << 32) | (uint32_t)lo;
#else
            return mul_rem(x, w, h, 0, m);
#endif
        }
        // x * w mod q in [0, 4q) for ANY 64-bit x: with h = floor(x wq / 2^64) - e, e <= 2, the remainder is
        // x w - h q = (x w - floor(x wq / 2^64) q) + e q < 2q + 2q.  12 instructions.
        template <class TW>
        static SHL_HD uint64_t mul_lazy4(uint64_t x, const TW &w, const Mod &m)
        {
            return mul_rem0(x, w, mul_hi_approx(x, w), m);
        }
        // [0,4q) -> [0,2q) without a carry chain: the sign of x - 2q selects
        static SHL_HD uint64_t guard(uint64_t x, const Mod &m)
        {
            const uint64_t d = x - m.two_q;
            return (int64_t)d < 0 ? x : d;
        }
        // Forward butterfly, 15 instructions: X, Y below B q -> below (B + 4) q, for any B with (B + 4) q <= 2^64.
        //   X' = X + t (the sum is the addend of the remainder chain), Y' = X + 4q - t = 2X + 4q - X', t = Y w mod q in [0, 4q)
        // The callers keep B + 4 <= the limit of the modulus class with fix4() (IntBounds).
        template <class TW>
        static SHL_HD void bfly_fwd(elem &X, elem &Y, const TW &w, const Mod &m)
        {
            SEALHIP_NOWRAP(X, m.four_q);
            const uint64_t xn = mul_rem(Y, w, mul_hi_approx(Y, w), X, m);
#if defined(__HIP_DEVICE_COMPILE__)
            Y = gfx::twice_plus(X, m.four_q) - xn;
#else
            Y = (X << 1) + m.four_q - xn;
#endif
            X = xn;
        }
        // X,Y in [0,4q) -> [0,4q)   (Arithmetic<>::guard/add/sub/mul_root, ntt.h:30-61): moduli of 2^60 and above
        static SHL_HD void bfly_fwd_guarded(elem &X, elem &Y, const tw_t &w, const Mod &m)
        {
            uint64_t x = guard(X, m);
            uint64_t t = mul_lazy(Y, w, m);
            X = x + t;
            Y = x + m.two_q - t;
        }
        // x < 64 q (and < 2^64) -> x - k * 2q in [0, 4q) with k in {floor(x / 2q) - 1, floor(x / 2q)}:
        // (x >> sx) 2^sx <= x and rcp <= 2^(31 + bits) / q give k <= x / 2q; the two truncations and the floor lose less than
        // 2^sx / 2q + (x >> sx) / 2^(32 + sh) + 1 < 1 + 2^(2 - sh), and sh >= 2 (x >> sx < 2^32 because x < 2^(bits + 6)).
        // HI32 (moduli of 40 bits and more, decided per kernel body): the high word alone is x >> 32 - no 64-bit shift; its
        // truncation loses 2^32 / 2q <= 2^-8.  Six / five VALU instructions
        // ([v_lshrrev_b64,] v_mul_hi_u32, v_lshrrev_b32, v_mad_u64_u32, v_mul_lo_u32, v_add_u32).
        template <bool HI32 = false>
        static SHL_HD void fix4(elem &x, const Mod &m)
        {
            const uint32_t xs = HI32 ? (uint32_t)(x >> 32) : (uint32_t)(x >> m.sx);
#if defined(__HIP_DEVICE_COMPILE__)
            const uint32_t k = gfx::mul_hi_s(xs, m.rcp) >> (HI32 ? m.sh_hi : m.sh);
            const uint64_t lo = gfx::mad64_s(k, m.m0, x);
            const uint32_t hi = gfx::add32((uint32_t)(lo >> 32), gfx::mul_lo_s(k, m.m1));
#else
            const uint32_t k = (uint32_t)(((uint64_t)xs * m.rcp) >> 32) >> (HI32 ? m.sh_hi : m.sh);
            const uint64_t lo = (uint64_t)k * m.m0 + x;
            const uint32_t hi = (uint32_t)(lo >> 32) + k * m.m1;
#endif
            x = ((uint64_t)hi << 32) | (uint32_t)lo;
        }
        static SHL_HD void fwd_fix(elem &x, const Mod &m)
        {
            fix4<false>(x, m);
        }
        // x < 64 q -> [0, q): the same estimate for floor(x / q) (one bit less of shift), then one conditional subtraction
        template <bool HI32 = false>
        static SHL_HD uint64_t canon_any(elem x, const Mod &m)
        {
            const uint32_t xs = HI32 ? (uint32_t)(x >> 32) : (uint32_t)(x >> m.sx);
#if defined(__HIP_DEVICE_COMPILE__)
            const uint32_t k = gfx::mul_hi_s(xs, m.rcp) >> ((HI32 ? m.sh_hi : m.sh) - 1);
            const uint64_t lo = gfx::mad64_s(k, m.n0, x);
            const uint32_t hi = gfx::add32((uint32_t)(lo >> 32), gfx::mul_lo_s(k, m.n1));
#else
            const uint32_t k = (uint32_t)(((uint64_t)xs * m.rcp) >> 32) >> ((HI32 ? m.sh_hi : m.sh) - 1);
            const uint64_t lo = (uint64_t)k * m.n0 + x;
            const uint32_t hi = (uint32_t)(lo >> 32) + k * m.n1;
#endif
            return csub(((uint64_t)hi << 32) | (uint32_t)lo, m.q);
        }
        // Inverse butterfly without a guard, 16 instructions: X, Y below 2^E q (c = 2^E q) -> X' = X + Y below 2^(E+1) q,
        // Y' = (X + c - Y) w mod q in [0, 4q).  Needs 2^(E+1) q <= 2^64 (IntBounds).
        template <class TW>
        static SHL_HD void bfly_inv_lazy(elem &X, elem &Y, const TW &w, uint64_t c, const Mod &m)
        {
            SEALHIP_NOWRAP(X, Y);
            SEALHIP_NOWRAP(X, c);
            const uint64_t s = X + Y, d = X + c - Y;
            X = s;
            Y = mul_lazy4(d, w, m);
        }
        // X,Y in [0,2q) -> [0,2q)   (dwthandler.h:202-356): moduli of 2^60 and above
        static SHL_HD void bfly_inv(elem &X, elem &Y, const tw_t &w, const Mod &m)
        {
            uint64_t s = X + Y, d = X + m.two_q - Y;
            X = guard(s, m);
            Y = mul_lazy(d, w, m);
        }
        // last inverse stage with N^-1 folded in (dwthandler.h:273-314): ni = N^-1, nw = N^-1 * w; X, Y below c (a multiple of q,
        // 2c <= 2^64); exact quotients, so that the results are in [0, 2q)
        static SHL_HD void bfly_inv_last(elem &X, elem &Y, const tw_t &ni, const tw_t &nw, uint64_t c, const Mod &m)
        {
            SEALHIP_NOWRAP(X, Y);
            SEALHIP_NOWRAP(X, c);
            uint64_t s = X + Y, d = X + c - Y;
            X = mul_lazy(s, ni, m);
            Y = mul_lazy(d, nw, m);
        }
        static SHL_HD void bfly_inv_last(elem &X, elem &Y, const tw_t &ni, const tw_t &nw, const Mod &m)
        {
            bfly_inv_last(X, Y, ni, nw, m.two_q, m);
        }
        static SHL_HD void fix(elem &, const Mod &)
        {}
        // forward result ([0,4q)) -> [0,q)
        static SHL_HD uint64_t fwd_to_canon(elem x, const Mod &m)
        {
            x = x >= m.two_q ? x - m.two_q : x;
            return csub(x, m.q);
        }
        static SHL_HD uint64_t fwd_to_lazy(elem x, const Mod &)
        {
            return x;
        }
        // inverse result ([0,2q)) -> [0,q)
        static SHL_HD uint64_t inv_to_canon(elem x, const Mod &m)
        {
            return csub(x, m.q);
        }
        static SHL_HD uint64_t inv_to_lazy(elem x, const Mod &)
        {
            return x;
        }
        static SHL_HD uint64_t raw(elem x)
        {
            return x;
        }
        static SHL_HD elem unraw(uint64_t x)
        {
            return x;
        }

        // key-switch inner product: 128-bit lazy sum of x*key, x in [0,4q), key canonical
        struct Acc
        {
            uint64_t lo, hi;
        };
        typedef uint64_t key_t;
        static SHL_HD Acc acc_zero()
        {
            return Acc{ 0, 0 };
        }
        static SHL_HD void mac(Acc &a, elem x, key_t k, const Mod &)
        {
            uint64_t pl, ph;
            mul_wide(x, k, pl, ph);
            a.lo += pl;
            a.hi += ph + (a.lo < pl);
        }
        static SHL_HD void acc_fix(Acc &, const Mod &)
        {}
        static SHL_HD uint64_t acc_to_canon(const Acc &a, const Mod &m)
        {
            return barrett128(a.lo, a.hi, m.md);
        }
    };

    // Compile-time range tracking for the integer back end.  ICLS = modulus class of a kernel body (decided per workgroup from
    // the prime, one wave-uniform branch at the top of each kernel):
    //   0 "tight"  2^58 <= q < 2^60: a word holds 16 q;   1 "roomy"  q < 2^58: 64 q (fix4()'s domain);   2 "wide"  q >= 2^60: the
    //   reference's guarded butterflies, every value in the reference's own ranges.
    // Forward: every value of a stage has the same bound B (in units of q), a stage adds 4: fix4() of all values before a stage
    // whenever B + 4 would pass the limit.  Inverse: the sum doubles, the product is back to 4 - tracked per register as an
    // exponent (the two operands of a butterfly always share their history inside a phase), see ntt2_kernels.hip.
    template <int ICLS>
    struct IntBounds
    {
        static constexpr bool wide = ICLS == 2;
        static constexpr bool hi32 = ICLS == 0;           // fix4<true>: the class guarantees q >= 2^40
        static constexpr int lim = ICLS == 0 ? 16 : 64;    // units of q
        static constexpr int lim_exp = ICLS == 0 ? 4 : 6;  // log2(lim)
        static constexpr bool fwd_fix_before(int b)
        {
            return !wide && b + 4 > lim;
        }
        static constexpr int fwd_after_stage(int b)
        {
            return wide ? 4 : (b + 4 > lim ? 4 : b) + 4;
        }
        static constexpr int fwd_after(int b, int stages)
        {
            for (int s = 0; s < stages; s++)
                b = fwd_after_stage(b);
            return b;
        }
        // exponent of register r of a phase before its stage number `idx`; the phase's stages pair register bits first_bit,
        // first_bit + 1, ... (Gentleman-Sande order); every register enters with exponent e_in.  A butterfly whose operands are
        // at the limit fixes them first (-> 2, i.e. 4 q).
        static constexpr int inv_exp(int r, int idx, int first_bit, int e_in)
        {
            int cur = e_in;
            for (int i = 0; i < idx; i++)
            {
                if (cur + 1 > lim_exp)
                    cur = 2;
                cur = ((r >> (first_bit + i)) & 1) ? 2 : cur + 1;
            }
            return cur;
        }
        // the exponent every register is brought under at the end of a phase, so that the next phase (other registers: the
        // history is in the lane index there) starts from one known bound
        static constexpr int inv_phase_out = lim_exp - 1;
    };

    // ---- double-precision back end (q < 2^50)
    template <>
    struct Field<true>
    {
        typedef double elem;
        typedef double tw_t;
        typedef FpDesc Mod;
        static constexpr int tw_words = 1;

        static SHL_HD Mod make_mod(const ModDesc &, const FpDesc &f)
        {
            return f;
        }
        static SHL_HD elem from_canon(uint64_t x, const Mod &)
        {
            return fp_from_u52(x);
        }
        static SHL_HD elem from_any(uint64_t x, const Mod &m)
        {
            return fp_from_u64(x, m);
        }
        static SHL_HD void bfly_fwd(elem &X, elem &Y, const tw_t &w, const Mod &m)
        {
            double t = fp_mulmod(Y, w, m.q, m.qinv);
            Y = X - t;
            X = X + t;
            SEALHIP_BOUND(X);
            SEALHIP_BOUND(Y);
        }
        static SHL_HD void bfly_inv(elem &X, elem &Y, const tw_t &w, const Mod &m)
        {
            double d = X - Y;
            X = X + Y;
            Y = fp_mulmod(d, w, m.q, m.qinv);
        }
        static SHL_HD void bfly_inv_last(elem &X, elem &Y, const tw_t &ni, const tw_t &nw, const Mod &m)
        {
            double s = X + Y, d = X - Y;
            X = fp_mulmod(s, ni, m.q, m.qinv);
            Y = fp_mulmod(d, nw, m.q, m.qinv);
        }
        static SHL_HD void fix(elem &x, const Mod &m)
        {
            x = fp_fix(x, m.q, m.qinv);
        }
        static SHL_HD void fwd_fix(elem &x, const Mod &m)
        {
            x = fp_fix(x, m.q, m.qinv);
        }
        // callers fix() before converting, so |x| <= q/2 + eps
        static SHL_HD uint64_t fwd_to_canon(elem x, const Mod &m)
        {
            return fp_to_canon(x, m);
        }
        static SHL_HD uint64_t fwd_to_lazy(elem x, const Mod &m)
        {
            return fp_to_canon(x, m);
        }
        static SHL_HD uint64_t inv_to_canon(elem x, const Mod &m)
        {
            return fp_to_canon(x, m);
        }
        static SHL_HD uint64_t inv_to_lazy(elem x, const Mod &m)
        {
            return fp_to_canon(x, m);
        }
        static SHL_HD uint64_t raw(elem x)
        {
            return fp_to_bits(x);
        }
        static SHL_HD elem unraw(uint64_t x)
        {
            return fp_from_bits(x);
        }

        // key-switch inner product: each product is reduced to |r| <= q (0.5 + 0.1875 B) - balanced key words, |x| <= B q with
        // B <= 2.64 (lean placement) - i.e. <= 0.995 q, and summed exactly; acc_fix() is called every 7 terms so the sum stays
        // below 0.5 q + 7 * 0.995 q = 7.47 q < 2^53.
        typedef double Acc;
        typedef double key_t; // the key component is stored as doubles for eligible primes
        static SHL_HD Acc acc_zero()
        {
            return 0.0;
        }
        static SHL_HD void mac(Acc &a, elem x, key_t k, const Mod &m)
        {
            a += fp_mulmod(x, k, m.q, m.qinv);
            SEALHIP_BOUND(a);
        }
        static SHL_HD void acc_fix(Acc &a, const Mod &m)
        {
            a = fp_fix(a, m.q, m.qinv);
        }
        static SHL_HD uint64_t acc_to_canon(const Acc &a, const Mod &m)
        {
            return fp_to_canon(fp_fix(a, m.q, m.qinv), m);
        }
    };
} // namespace sealhip
