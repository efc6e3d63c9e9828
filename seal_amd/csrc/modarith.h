// 64-bit modular word arithmetic for gfx950, shared by host table construction and device kernels.
//
// Restates (does not copy) the word-level layer of the reference:
//   MultiplyUIntModOperand / multiply_uint_mod[_lazy]  native/src/seal/util/uintarithsmallmod.h:255-326
//   barrett_reduce_128 / barrett_reduce_64             native/src/seal/util/uintarithsmallmod.h:167-230
//   Modulus::const_ratio (floor(2^128 / q))            native/src/seal/modulus.cpp:66-105
//
// gfx950 has no 64-bit integer multiplier: a 64x64 product is four v_mad_u64_u32 (measured
// ~4 cycles per wave64 instruction per SIMD, profiles/r01_microbench_intmul.txt), so every
// routine here is written to minimise 32x32 products, not to mirror the CPU instruction mix.
// All public results that can reach a ciphertext are canonical residues, which is what makes
// the output bit-identical to the reference whatever lazy ranges are used inside.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SHL_HD __host__ __device__ __forceinline__

// A/B switches of the development builds (tools/quick/ab_*.sh): environment variables that select a losing or superseded
// code path for a measurement.  They exist only in libraries built with -DSEALHIP_AB_SWITCHES; the product library reads the
// five switches listed in include/sealhip.h ("Environment") and nothing else.
#include <cstdlib>
inline const char *shl_ab_getenv(const char *name)
{
#ifdef SEALHIP_AB_SWITCHES
    return std::getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

// Wave-uniform, read-only tables (conversion matrices, per-prime constants, the twiddles of the first stages): read
// through the constant address space the compiler may treat them as invariant and use scalar loads (s_load_dwordx2..16
// into SGPRs) instead of one vector load per lane-uniform element, which also frees the VGPRs that held them.
#if defined(__HIP_DEVICE_COMPILE__)
typedef const uint64_t __attribute__((address_space(4))) *shl_uconst_ptr;
typedef const uint32_t __attribute__((address_space(4))) *shl_uconst32_ptr;
#define SHL_UCONST(p) ((shl_uconst_ptr)(uintptr_t)(p))
#define SHL_UCONST32(p) ((shl_uconst32_ptr)(uintptr_t)(p))
// a value that is the same in every lane of the workgroup (block-level index): move it to an SGPR so that
// everything derived from it (table base addresses, per-prime constants) is provably uniform
#define SHL_UNIFORM(x) ((unsigned)__builtin_amdgcn_readfirstlane((int)(x)))
#else
typedef const uint64_t *shl_uconst_ptr;
typedef const uint32_t *shl_uconst32_ptr;
#define SHL_UCONST(p) (p)
#define SHL_UCONST32(p) (p)
#define SHL_UNIFORM(x) (x)
#endif

namespace sealhip
{
    // One precomputed multiplicand: w and floor(w * 2^64 / q)  ("Shoup pair").
    struct __attribute__((aligned(16))) ShoupOp
    {
        uint64_t w;
        uint64_t wq;
    };

    // Everything a kernel needs to know about one prime.  `ratio` = floor(2^128 / q) (two words).
    struct __attribute__((aligned(16))) ModDesc
    {
        uint64_t q;
        uint64_t two_q;
        uint64_t ratio_lo;
        uint64_t ratio_hi;
    };

    // A read-only window of global memory whose base is the same in every lane (round 3): loads through it are gfx950 buffer
    // loads - address = base (4 SGPRs) + a 32-bit per-lane byte offset + a wave-uniform byte offset (an SGPR or the 12-bit
    // immediate) - so the strided loads of a tile (row e at e * 2 KiB or 4 KiB) cost no vector instruction for their
    // addresses, where flat global loads needed a 64-bit VGPR add (v_add_co + v_addc and a VCC wait state) per row.
    // Offsets are bytes, below 2^31 together.  The emulated build reads the same bytes through the pointer.
    struct UniformView
    {
#if defined(__HIP_DEVICE_COMPILE__)
        __amdgpu_buffer_rsrc_t rsrc;
#else
        const char *base;
#endif
    };
    template <class T>
    __device__ __forceinline__ UniformView uniform_view(const T *wave_uniform_base)
    {
#if defined(__HIP_DEVICE_COMPILE__)
        // raw buffer (stride 0), 2^31 - 1 bytes, word 3 = DATA_FORMAT 32 (the untyped-load setting of gfx90a / gfx94x / gfx950)
        return UniformView{ __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(wave_uniform_base), 0, 0x7fffffff, 0x00020000) };
#else
        return UniformView{ reinterpret_cast<const char *>(wave_uniform_base) };
#endif
    }
    __device__ __forceinline__ uint64_t view_load64(const UniformView &v, unsigned lane_bytes, unsigned uniform_bytes)
    {
#if defined(__HIP_DEVICE_COMPILE__)
        typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
        const u32x2 r = __builtin_amdgcn_raw_buffer_load_b64(v.rsrc, (int)lane_bytes, (int)uniform_bytes, 0);
        return ((uint64_t)r.y << 32) | r.x;
#else
        return *reinterpret_cast<const uint64_t *>(v.base + lane_bytes + uniform_bytes);
#endif
    }
    // the same load with the non-temporal hint (gfx940+ cache-policy bit 1): data that is read once - the key switch's intermediate -
    // is not to displace what the L2 holds for reuse (the key tile of a whole batch)
    __device__ __forceinline__ uint64_t view_load64_nt(const UniformView &v, unsigned lane_bytes, unsigned uniform_bytes)
    {
#if defined(__HIP_DEVICE_COMPILE__)
        typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
        const u32x2 r = __builtin_amdgcn_raw_buffer_load_b64(v.rsrc, (int)lane_bytes, (int)uniform_bytes, 2);
        return ((uint64_t)r.y << 32) | r.x;
#else
        return *reinterpret_cast<const uint64_t *>(v.base + lane_bytes + uniform_bytes);
#endif
    }
    // two consecutive 64-bit words (16-byte aligned)
    __device__ __forceinline__ void view_load128(const UniformView &v, unsigned lane_bytes, unsigned uniform_bytes, uint64_t &w0, uint64_t &w1)
    {
#if defined(__HIP_DEVICE_COMPILE__)
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(v.rsrc, (int)lane_bytes, (int)uniform_bytes, 0);
        w0 = ((uint64_t)r.y << 32) | r.x;
        w1 = ((uint64_t)r.w << 32) | r.z;
#else
        const uint64_t *p = reinterpret_cast<const uint64_t *>(v.base + lane_bytes + uniform_bytes);
        w0 = p[0];
        w1 = p[1];
#endif
    }

    SHL_HD uint64_t mul_hi64(uint64_t a, uint64_t b)
    {
#if defined(__HIP_DEVICE_COMPILE__)
        return __umul64hi(a, b);
#else
        return (uint64_t)(((unsigned __int128)a * b) >> 64);
#endif
    }

    SHL_HD void mul_wide(uint64_t a, uint64_t b, uint64_t &lo, uint64_t &hi)
    {
        lo = a * b;
        hi = mul_hi64(a, b);
    }

    // [0, 2q) -> [0, q)
    SHL_HD uint64_t csub(uint64_t x, uint64_t q)
    {
        return x >= q ? x - q : x;
    }

    // x * w mod q, result in [0, 2q) for any 64-bit x, given wq = floor(w * 2^64 / q), w < q.
    SHL_HD uint64_t mul_shoup_lazy(uint64_t x, uint64_t w, uint64_t wq, uint64_t q)
    {
        uint64_t h = mul_hi64(x, wq);
        return x * w - h * q;
    }
    SHL_HD uint64_t mul_shoup(uint64_t x, uint64_t w, uint64_t wq, uint64_t q)
    {
        return csub(mul_shoup_lazy(x, w, wq, q), q);
    }

    // Barrett reduction of a 128-bit value (hi:lo) to [0, q).  Requires hi:lo < 2^128 and q < 2^62.
    // Quotient estimate floor(x * ratio / 2^128) is low by at most 2; the remainder therefore fits
    // 64 bits and two conditional subtractions finish it.
    SHL_HD uint64_t barrett128(uint64_t lo, uint64_t hi, const ModDesc &m)
    {
        // q_est = floor( (hi*2^64 + lo) * (rhi*2^64 + rlo) / 2^128 ), dropping lo*rlo's low word effects
        uint64_t t1 = mul_hi64(lo, m.ratio_lo);
        uint64_t a_lo, a_hi;
        mul_wide(lo, m.ratio_hi, a_lo, a_hi);
        uint64_t b_lo, b_hi;
        mul_wide(hi, m.ratio_lo, b_lo, b_hi);
        // mid = t1 + a_lo + b_lo  (carry into the quotient word)
        uint64_t mid = t1 + a_lo;
        uint64_t c = mid < t1;
        uint64_t mid2 = mid + b_lo;
        c += mid2 < mid;
        uint64_t qest = hi * m.ratio_hi + a_hi + b_hi + c;
        uint64_t r = lo - qest * m.q;
        r = r >= m.two_q ? r - m.two_q : r;
        return csub(r, m.q);
    }

    // x mod q for a single word x (any value).
    SHL_HD uint64_t barrett64(uint64_t x, const ModDesc &m)
    {
        uint64_t qest = mul_hi64(x, m.ratio_hi);
        uint64_t r = x - qest * m.q;
        return csub(r, m.q);
    }

    // a * b mod q, canonical, for any 64-bit a, b with a*b < 2^128 (always) and q < 2^62.
    SHL_HD uint64_t mul_mod(uint64_t a, uint64_t b, const ModDesc &m)
    {
        uint64_t lo, hi;
        mul_wide(a, b, lo, hi);
        return barrett128(lo, hi, m);
    }

    SHL_HD uint64_t add_mod(uint64_t a, uint64_t b, uint64_t q)
    {
        return csub(a + b, q);
    }
    SHL_HD uint64_t sub_mod(uint64_t a, uint64_t b, uint64_t q)
    {
        return a >= b ? a - b : a + q - b;
    }
    SHL_HD uint64_t neg_mod(uint64_t a, uint64_t q)
    {
        return a ? q - a : 0;
    }
} // namespace sealhip
