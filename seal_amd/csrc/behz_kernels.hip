// BEHZ base conversions for BFV multiplication, one thread per coefficient.
//
// Restates RNSTool::fastbconv_m_tilde (util/rns.cpp:1086-1131), sm_mrq (979-1039), fast_floor
// (1041-1084), fastbconv_sk (903-977) and BaseConverter::fast_convert_array (418-463).  These are
// approximate conversions whose integer formulas must be reproduced literally: the
// conversion result is x + alpha*Q with alpha fixed by exact integer arithmetic, so every dot
// product below is the exact sum mod p (128-bit accumulate + Barrett-128), like dot_product_mod
// (util/uintarithsmallmod.cpp:110-175).
//
// Shape: an O(|ibase| x |obase|) integer mat-vec per coefficient.  It stays on the VALU (64-bit
// modular words, 128-bit accumulators — not an MFMA contraction).  Threads of a wave take
// consecutive coefficients so every global access is a coalesced 512-byte row segment; the
// per-coefficient input vector is parked in LDS as [component][thread] (conflict-free) so the
// output loop can re-read it with the matrix row held in scalar registers.
#include "field.h"
#include "poly_kernels.h"

namespace sealhip
{
    namespace
    {
        constexpr unsigned kBlock = 256;

        __device__ __forceinline__ void mac128(uint64_t &lo, uint64_t &hi, uint64_t a, uint64_t b)
        {
            uint64_t pl, ph;
            mul_wide(a, b, pl, ph);
            lo += pl;
            hi += ph + (lo < pl);
        }

        // y_i = x_i * (Q/q_i)^-1 mod q_i (canonical) for i < count, staged in LDS as y[i*kBlock + tid].
        // Then out_j = sum_i y_i * M[j*count + i] mod p_j.
        __device__ __forceinline__ uint64_t dot_lds(const uint64_t *y, const uint64_t *row, unsigned count, const ModDesc &m)
        {
            uint64_t lo = 0, hi = 0;
            for (unsigned i = 0; i < count; i++)
                mac128(lo, hi, y[i * kBlock + threadIdx.x], row[i]);
            return barrett128(lo, hi, m);
        }

        typedef shl_uconst_ptr uconst_ptr;
        // wave-uniform per-prime constants through the scalar cache
        __device__ __forceinline__ ShoupOp ld_shoup(const ShoupOp *p)
        {
            uconst_ptr u = SHL_UCONST(reinterpret_cast<const uint64_t *>(p));
            return ShoupOp{ u[0], u[1] };
        }
        __device__ __forceinline__ ModDesc ld_mod(const ModDesc *p)
        {
            uconst_ptr u = SHL_UCONST(reinterpret_cast<const uint64_t *>(p));
            return ModDesc{ u[0], u[1], u[2], u[3] };
        }
        __device__ __forceinline__ uint64_t ld_u64(const uint64_t *p)
        {
            return SHL_UCONST(p)[0];
        }
        __device__ __forceinline__ uint32_t ld_u32(const uint32_t *p)
        {
            return SHL_UCONST32(p)[0];
        }

        // 64x64 -> 128 product with the four 32x32 partial products computed once (v_mad_u64_u32 chain)
        __device__ __forceinline__ void mul_wide4(uint64_t a, uint64_t b, uint64_t &lo, uint64_t &hi)
        {
            const uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), b0 = (uint32_t)b, b1 = (uint32_t)(b >> 32);
            const uint64_t p00 = (uint64_t)a0 * b0;
            const uint64_t t = (uint64_t)a0 * b1 + (p00 >> 32);          // < 2^64
            const uint64_t u = (uint64_t)a1 * b0 + (uint32_t)t;          // < 2^64
            hi = (uint64_t)a1 * b1 + (t >> 32) + (u >> 32);
            lo = (u << 32) | (uint32_t)p00;
        }
        // a * s + c with the wave-uniform s in an SGPR: one v_mad_u64_u32 (field.h, gfx::mad64_s: issued through an instruction
        // wrapper so that the compiler does not rebuild register pairs around it)
        __device__ __forceinline__ uint64_t mad_uniform(uint32_t a, uint32_t s, uint64_t c)
        {
#if defined(__HIP_DEVICE_COMPILE__)
            return gfx::mad64_s(a, s, c);
#else
            return (uint64_t)a * s + c;
#endif
        }
        // Exact dot product modulo m of a vector held in REGISTERS with a matrix row read through the scalar cache
        // (dot_product_mod, util/uintarithsmallmod.cpp:110-175): the loop is unrolled to the compile-time bound KM and
        // predicated on the wave-uniform length, so the LDS round trip and the loop-carried latency of dot_lds are gone.
        //
        // Round 3: no carries.  A 64 x 64 -> 128 product accumulated into 128 bits costs four products plus the carry handling
        // (sixteen instructions a term as compiled).  The matrix entry is wave-uniform, so it is cut into three 21-bit limbs with
        // scalar instructions (free for the vector unit); y = y1 2^32 + y0 < 2^61 then gives six products below 2^53 that go to
        // six 64-bit column sums (weights 2^0, 2^21, 2^42 for y0 and 2^32, 2^53, 2^74 for y1) by six v_mad_u64_u32 - no
        // carry, no register pair to build - and 64 terms stay below 2^59.  The columns are added up once per dot product.
        // (hi 2^64 + lo) mod p without the 128 x 128 -> 256 product of Barrett's reduction (79 instructions as compiled): the high word is
        // folded in by a Shoup product with 2^64 mod p (field.h: mul_lazy4, any 64-bit operand, result below 4p), the low word by
        // one with 1 (quotient operand floor(2^64 / p) = the high word of the Barrett ratio), and the sum - below 8p < 2^64 - is made
        // canonical by the reciprocal estimate of canon_any: 12 + 12 + 1 + 12 instructions
        __device__ __forceinline__ uint64_t fold128(uint64_t lo, uint64_t hi, const ModDesc &md, const ShoupOp *two64)
        {
            typedef Field<false> F;
            const F::Mod m = F::make_mod(md, FpDesc{});
            const ShoupOp c = ld_shoup(two64);
            ShoupOpU c64, one;
            c64.w = c.w;
            c64.wq = c.wq;
            one.w = 1;
            one.wq = md.ratio_hi;
            const uint64_t u = F::mul_lazy4(hi, c64, m), v = F::mul_lazy4(lo, one, m);
            return F::canon_any<false>(u + v, m);
        }
        // extra_y * extra_r (both below 2^61) is one more term of the sum when extra_r != 0 (wave-uniform)
        template <unsigned KM>
        __device__ __forceinline__ uint64_t dot_reg(const uint64_t (&y)[KM], uconst_ptr row, unsigned count, const ModDesc &m, const ShoupOp *two64,
                                                    uint64_t extra_y = 0, uint64_t extra_r = 0)
        {
            static_assert(KM <= 128, "six column sums of terms below 2^53: 129 of them stay below 2^60");
            // row: the matrix entries cut into their three 21-bit limbs at context build (context.cpp: split21), two words per entry
            // {limb0 | limb1 << 32, limb2} - cutting them here cost as many scalar instructions as the products cost vector ones
            uint64_t a0 = 0, a1 = 0, a2 = 0, b0 = 0, b1 = 0, b2 = 0;
            if (extra_r)
            {
                const uint32_t ra = (uint32_t)extra_r & 0x1FFFFFu, rb = (uint32_t)(extra_r >> 21) & 0x1FFFFFu, rc = (uint32_t)(extra_r >> 42);
                const uint32_t y0 = (uint32_t)extra_y, y1 = (uint32_t)(extra_y >> 32);
                a0 = mad_uniform(y0, ra, 0);
                a1 = mad_uniform(y0, rb, 0);
                a2 = mad_uniform(y0, rc, 0);
                b0 = mad_uniform(y1, ra, 0);
                b1 = mad_uniform(y1, rb, 0);
                b2 = mad_uniform(y1, rc, 0);
            }
#pragma unroll
            for (unsigned i = 0; i < KM; i++)
                if (i < count)
                {
                    const uint64_t r01 = row[2 * i], r2 = row[2 * i + 1];
                    const uint32_t ra = (uint32_t)r01, rb = (uint32_t)(r01 >> 32), rc = (uint32_t)r2;
                    const uint32_t y0 = (uint32_t)y[i], y1 = (uint32_t)(y[i] >> 32);
                    a0 = mad_uniform(y0, ra, a0);
                    a1 = mad_uniform(y0, rb, a1);
                    a2 = mad_uniform(y0, rc, a2);
                    b0 = mad_uniform(y1, ra, b0);
                    b1 = mad_uniform(y1, rb, b1);
                    b2 = mad_uniform(y1, rc, b2);
                }
            typedef unsigned __int128 u128;
            const u128 t = (u128)a0 + ((u128)a1 << 21) + ((u128)a2 << 42) + ((u128)b0 << 32) + ((u128)b1 << 53) + ((u128)b2 << 74);
            return fold128((uint64_t)t, (uint64_t)(t >> 64), m, two64);
        }

        // ---- stage 0: q -> Bsk U {m~}   (fastbconv_m_tilde)
        __device__ __forceinline__ void stage_fastbconv_m_tilde(
            const ModDesc *mods, const LevelDev &lv, const uint64_t *in, uint64_t *out, size_t N, size_t j, uint64_t *y)
        {
            const unsigned K = lv.K;
            for (unsigned i = 0; i < K; i++)
            {
                const uint64_t q = mods[i].q;
                const ShoupOp mt = lv.m_tilde_mod_q[i], ip = lv.inv_punct_q[i];
                uint64_t v = mul_shoup(in[i * N + j], mt.w, mt.wq, q);  // x * m~ mod q_i   (rns.cpp:1120)
                y[i * kBlock + threadIdx.x] = mul_shoup(v, ip.w, ip.wq, q); // * (Q/q_i)^-1   (rns.cpp:439-456)
            }
            for (unsigned jj = 0; jj < lv.nBsk; jj++)
                out[jj * N + j] = dot_lds(y, lv.q_to_bsk + jj * K, K, mods[lv.bsk_prime[jj]]);
            uint64_t s = 0;
            for (unsigned i = 0; i < K; i++)
                s += y[i * kBlock + threadIdx.x] * lv.q_to_mtilde[i];
            out[lv.nBsk * N + j] = s & (lv.m_tilde - 1); // modulus m~ = 2^32
        }

        // ---- stage 1: Bsk U {m~} -> Bsk   (sm_mrq)
        __device__ __forceinline__ void stage_sm_mrq(
            const ModDesc *mods, const LevelDev &lv, const uint64_t *in, uint64_t *out, size_t N, size_t j)
        {
            const uint64_t mt = lv.m_tilde;
            uint64_t r = (in[lv.nBsk * N + j] * lv.neg_inv_prod_q_mod_mtilde) & (mt - 1);
            for (unsigned jj = 0; jj < lv.nBsk; jj++)
            {
                const ModDesc md = mods[lv.bsk_prime[jj]];
                uint64_t tmp = r;
                if (tmp >= (mt >> 1))
                    tmp += md.q - mt; // centered reduction (rns.cpp:1027-1031)
                uint64_t lo, hi;
                mul_wide(tmp, lv.prod_q_mod_bsk[jj], lo, hi);
                uint64_t c = in[jj * N + j];
                lo += c;
                hi += lo < c;
                uint64_t v = barrett128(lo, hi, md);
                const ShoupOp im = lv.inv_mtilde_mod_bsk[jj];
                out[jj * N + j] = mul_shoup(v, im.w, im.wq, md.q);
            }
        }

        // ---- stage 2: q U Bsk -> Bsk   (fast_floor); input comps [0,K) base q, [K, K+nBsk) base Bsk
        __device__ __forceinline__ void stage_fast_floor(
            const ModDesc *mods, const LevelDev &lv, const uint64_t *in, uint64_t *out, size_t N, size_t j, uint64_t *y)
        {
            const unsigned K = lv.K;
            for (unsigned i = 0; i < K; i++)
            {
                const ShoupOp ip = lv.inv_punct_q[i];
                y[i * kBlock + threadIdx.x] = mul_shoup(in[i * N + j], ip.w, ip.wq, mods[i].q);
            }
            for (unsigned jj = 0; jj < lv.nBsk; jj++)
            {
                const ModDesc md = mods[lv.bsk_prime[jj]];
                uint64_t conv = dot_lds(y, lv.q_to_bsk + jj * K, K, md);
                const ShoupOp iq = lv.inv_prod_q_mod_bsk[jj];
                out[jj * N + j] = mul_shoup(in[(K + jj) * N + j] + (md.q - conv), iq.w, iq.wq, md.q);
            }
        }

        // ---- stage 3: Bsk -> q   (fastbconv_sk)
        __device__ __forceinline__ void stage_fastbconv_sk(
            const ModDesc *mods, const LevelDev &lv, const uint64_t *in, uint64_t *out, size_t N, size_t j, uint64_t *y)
        {
            const unsigned K = lv.K, nB = lv.nB;
            for (unsigned i = 0; i < nB; i++)
            {
                const ShoupOp ip = lv.inv_punct_b[i];
                y[i * kBlock + threadIdx.x] = mul_shoup(in[i * N + j], ip.w, ip.wq, mods[lv.bsk_prime[i]].q);
            }
            const ModDesc msk = mods[lv.msk_prime];
            uint64_t conv_sk = dot_lds(y, lv.b_to_msk, nB, msk);
            uint64_t alpha = mul_shoup(conv_sk + (msk.q - in[nB * N + j]), lv.inv_prod_b_mod_msk.w, lv.inv_prod_b_mod_msk.wq, msk.q);
            const bool negative = alpha > (msk.q >> 1);
            const uint64_t mag = negative ? msk.q - alpha : alpha;
            for (unsigned i = 0; i < K; i++)
            {
                const ModDesc md = mods[i];
                uint64_t g = dot_lds(y, lv.b_to_q + i * nB, nB, md);
                uint64_t pb = lv.prod_b_mod_q[i];
                uint64_t factor = negative ? pb : md.q - pb; // rns.cpp:962-975
                uint64_t lo, hi;
                mul_wide(mag, factor, lo, hi);
                lo += g;
                hi += lo < g;
                out[i * N + j] = barrett128(lo, hi, md);
            }
        }

        __global__ void __launch_bounds__(kBlock) behz_stage_kernel(
            const ModDesc *mods, LevelDev lv, int which, const uint64_t *in, uint64_t *out, unsigned n_log, size_t items,
            unsigned in_comps, unsigned out_comps)
        {
            HIP_DYNAMIC_SHARED(uint64_t, y)
            const size_t N = size_t(1) << n_log;
            const size_t total = items << n_log;
            for (size_t t = blockIdx.x * (size_t)kBlock + threadIdx.x; t < total; t += (size_t)gridDim.x * kBlock)
            {
                const size_t item = t >> n_log, j = t & (N - 1);
                const uint64_t *ip = in + item * in_comps * N;
                uint64_t *op = out + item * out_comps * N;
                if (which == 0)
                    stage_fastbconv_m_tilde(mods, lv, ip, op, N, j, y);
                else if (which == 1)
                    stage_sm_mrq(mods, lv, ip, op, N, j);
                else if (which == 2)
                    stage_fast_floor(mods, lv, ip, op, N, j, y);
                else
                    stage_fastbconv_sk(mods, lv, ip, op, N, j, y);
            }
        }

        // fused lift: fastbconv_m_tilde + sm_mrq without the Bsk U {m~} round trip through HBM.
        // KM >= K: the per-coefficient input vector lives in registers (dot_reg).
        template <unsigned KM>
        __global__ void __launch_bounds__(kBlock) behz_lift_kernel(
            const ModDesc *mods, LevelDev lv, const uint64_t *in, uint64_t *out, unsigned n_log, size_t items)
        {
            const size_t N = size_t(1) << n_log;
            const size_t total = items << n_log;
            const unsigned K = lv.K;
            const uint64_t mt = lv.m_tilde;
            for (size_t t = blockIdx.x * (size_t)kBlock + threadIdx.x; t < total; t += (size_t)gridDim.x * kBlock)
            {
                const size_t item = t >> n_log, j = t & (N - 1);
                const uint64_t *ip = in + item * K * N;
                uint64_t *op = out + item * lv.nBsk * N;
                uint64_t y[KM];
                uint64_t s = 0;
#pragma unroll
                for (unsigned i = 0; i < KM; i++)
                {
                    y[i] = 0;
                    if (i < K)
                    {
                        // x m~ (Q/q_i)^-1 mod q_i: the two constants of rns.cpp:1120 and 439-456 as one (LevelDev)
                        const uint64_t q = ld_u64(&mods[i].q);
                        const ShoupOp c = ld_shoup(&lv.mt_inv_punct_q[i]);
                        y[i] = mul_shoup(ip[i * N + j], c.w, c.wq, q);
                        s += y[i] * ld_u64(&lv.q_to_mtilde[i]);
                    }
                }
                uint64_t r = (((s & (mt - 1)) * lv.neg_inv_prod_q_mod_mtilde)) & (mt - 1);
                for (unsigned jj = 0; jj < lv.nBsk; jj++)
                {
                    // (conv + r_centered Q) m~^-1 mod p_j (sm_mrq, rns.cpp:1027-1037) as ONE exact sum: the matrix row and Q already
                    // carry m~^-1, r_centered Q m~^-1 is one more term
                    const ModDesc md = ld_mod(&mods[ld_u32(&lv.bsk_prime[jj])]);
                    uint64_t tmp = r;
                    if (tmp >= (mt >> 1))
                        tmp += md.q - mt;
                    op[jj * N + j] = dot_reg<KM>(y, SHL_UCONST(lv.q_to_bsk_lift + 2 * jj * K), K, md, &lv.two64_bsk[jj], tmp, ld_u64(&lv.prod_q_lift[jj]));
                }
            }
        }

        // fused tail: (x t) -> fast_floor -> fastbconv_sk   (evaluator.cpp:549-566).  KM >= nB = K or K + 1.
        // The q-side vector and the B-side vector live in registers; only the |Bsk| floor values, which are written
        // under a run-time index, pass through LDS once.
        template <unsigned KM>
        __global__ void __launch_bounds__(kBlock) behz_floor_sk_kernel(
            const ModDesc *mods, LevelDev lv, const uint64_t *dq, const uint64_t *dbsk, uint64_t *out, unsigned n_log,
            size_t items)
        {
            HIP_DYNAMIC_SHARED(uint64_t, f) // [nBsk][kBlock]
            const size_t N = size_t(1) << n_log;
            const size_t total = items << n_log;
            const unsigned K = lv.K, nB = lv.nB, nBsk = lv.nBsk;
            for (size_t t = blockIdx.x * (size_t)kBlock + threadIdx.x; t < total; t += (size_t)gridDim.x * kBlock)
            {
                const size_t item = t >> n_log, j = t & (N - 1);
                const uint64_t *qp = dq + item * K * N;
                const uint64_t *bp = dbsk + item * nBsk * N;
                uint64_t *op = out + item * K * N;
                uint64_t y[KM];
#pragma unroll
                for (unsigned i = 0; i < KM; i++)
                {
                    y[i] = 0;
                    if (i < K)
                    {
                        // step (6), evaluator.cpp:554, and the (Q/q_i)^-1 of the base conversion as one constant
                        const uint64_t q = ld_u64(&mods[i].q);
                        const ShoupOp c = ld_shoup(&lv.t_inv_punct_q[i]);
                        y[i] = mul_shoup(qp[i * N + j], c.w, c.wq, q);
                    }
                }
                for (unsigned jj = 0; jj < nBsk; jj++)
                {
                    // step (7): (x_bsk t - conv) Q^-1 mod p_j, and for the primes of B also the (B/b_j)^-1 that step (8) applies next:
                    // the matrix row and t carry those constants (LevelDev), one Shoup product and one exact sum remain.  f[jj]
                    // is therefore step (8)'s input vector for jj < nB and the floor value itself for m_sk (jj = nB)
                    const ModDesc md = ld_mod(&mods[ld_u32(&lv.bsk_prime[jj])]);
                    const uint64_t conv = dot_reg<KM>(y, SHL_UCONST(lv.q_to_bsk_floor + 2 * jj * K), K, md, &lv.two64_bsk[jj]);
                    const ShoupOp tm = ld_shoup(&lv.t_floor_bsk[jj]);
                    const uint64_t zb = mul_shoup(bp[jj * N + j], tm.w, tm.wq, md.q);
                    f[jj * kBlock + threadIdx.x] = csub(zb + (md.q - conv), md.q);
                }
                // step (8): Shenoy-Kumaresan (each thread reads back only what it wrote: no barrier)
#pragma unroll
                for (unsigned i = 0; i < KM; i++)
                {
                    y[i] = 0;
                    if (i < nB)
                        y[i] = f[i * kBlock + threadIdx.x];
                }
                const ModDesc msk = ld_mod(&mods[lv.msk_prime]);
                uint64_t conv_sk = dot_reg<KM>(y, SHL_UCONST(lv.b_to_msk3), nB, msk, &lv.two64_bsk[nB]);
                uint64_t alpha = mul_shoup(
                    conv_sk + (msk.q - f[nB * kBlock + threadIdx.x]), lv.inv_prod_b_mod_msk.w, lv.inv_prod_b_mod_msk.wq, msk.q);
                const bool negative = alpha > (msk.q >> 1);
                const uint64_t mag = negative ? msk.q - alpha : alpha;
                for (unsigned i = 0; i < K; i++)
                {
                    // conv +/- |alpha| B mod q_i (rns.cpp:962-975) as ONE exact sum: |alpha| B is one more term of the dot product, with
                    // -|alpha| written as the non-negative (multiple of q_i above 2^60) - |alpha|   (|alpha| <= m_sk / 2 < 2^60)
                    const ModDesc md = ld_mod(&mods[i]);
                    const uint64_t ey = negative ? mag : ld_u64(&lv.neg_base_q[i]) - mag;
                    op[i * N + j] = dot_reg<KM>(y, SHL_UCONST(lv.b_to_q3 + 2 * i * nB), nB, md, &lv.two64_q[i], ey, ld_u64(&lv.prod_b_mod_q[i]));
                }
            }
        }

        inline unsigned grid_for(size_t work)
        {
            size_t b = (work + kBlock - 1) / kBlock;
            if (b > 2048)
                b = 2048;
            if (b == 0)
                b = 1;
            return (unsigned)b;
        }
    } // namespace

    hipError_t k_behz_lift(
        const ModDesc *mods, const LevelDev &lv, const uint64_t *in, uint64_t *out, unsigned n_log, size_t items, hipStream_t s)
    {
        size_t work = items << n_log;
        if (!work)
            return hipSuccess;
        const dim3 g(grid_for(work)), b(kBlock);
        if (lv.K <= 4)
            hipLaunchKernelGGL(behz_lift_kernel<4>, g, b, 0, s, mods, lv, in, out, n_log, items);
        else if (lv.K <= 8)
            hipLaunchKernelGGL(behz_lift_kernel<8>, g, b, 0, s, mods, lv, in, out, n_log, items);
        else if (lv.K <= 16)
            hipLaunchKernelGGL(behz_lift_kernel<16>, g, b, 0, s, mods, lv, in, out, n_log, items);
        else if (lv.K <= 32)
            hipLaunchKernelGGL(behz_lift_kernel<32>, g, b, 0, s, mods, lv, in, out, n_log, items);
        else
            hipLaunchKernelGGL(behz_lift_kernel<kMaxComps>, g, b, 0, s, mods, lv, in, out, n_log, items);
        return hipGetLastError();
    }

    hipError_t k_behz_floor_sk(
        const ModDesc *mods, const LevelDev &lv, const uint64_t *dq, const uint64_t *dbsk, uint64_t *out, unsigned n_log,
        size_t items, hipStream_t s)
    {
        size_t work = items << n_log;
        if (!work)
            return hipSuccess;
        const size_t shmem = (size_t)lv.nBsk * kBlock * sizeof(uint64_t);
        const dim3 g(grid_for(work)), b(kBlock);
        const unsigned need = lv.nB > lv.K ? lv.nB : lv.K;
        if (need <= 4)
            hipLaunchKernelGGL(behz_floor_sk_kernel<4>, g, b, shmem, s, mods, lv, dq, dbsk, out, n_log, items);
        else if (need <= 8)
            hipLaunchKernelGGL(behz_floor_sk_kernel<8>, g, b, shmem, s, mods, lv, dq, dbsk, out, n_log, items);
        else if (need <= 16)
            hipLaunchKernelGGL(behz_floor_sk_kernel<16>, g, b, shmem, s, mods, lv, dq, dbsk, out, n_log, items);
        else if (need <= 32)
            hipLaunchKernelGGL(behz_floor_sk_kernel<32>, g, b, shmem, s, mods, lv, dq, dbsk, out, n_log, items);
        else
            hipLaunchKernelGGL(behz_floor_sk_kernel<kMaxComps + 1>, g, b, shmem, s, mods, lv, dq, dbsk, out, n_log, items);
        return hipGetLastError();
    }

    hipError_t k_behz_stage(
        const ModDesc *mods, const LevelDev &lv, int which, const uint64_t *in, uint64_t *out, unsigned n_log, size_t items,
        hipStream_t s)
    {
        size_t work = items << n_log;
        if (!work)
            return hipSuccess;
        unsigned in_comps, out_comps;
        switch (which)
        {
        case 0:
            in_comps = lv.K;
            out_comps = lv.nBsk + 1;
            break;
        case 1:
            in_comps = lv.nBsk + 1;
            out_comps = lv.nBsk;
            break;
        case 2:
            in_comps = lv.K + lv.nBsk;
            out_comps = lv.nBsk;
            break;
        case 3:
            in_comps = lv.nBsk;
            out_comps = lv.K;
            break;
        default:
            return hipErrorInvalidValue;
        }
        size_t shmem = (size_t)(lv.K > lv.nB ? lv.K : lv.nB) * kBlock * sizeof(uint64_t);
        hipLaunchKernelGGL(
            behz_stage_kernel, dim3(grid_for(work)), dim3(kBlock), shmem, s, mods, lv, which, in, out, n_log, items, in_comps,
            out_comps);
        return hipGetLastError();
    }
} // namespace sealhip
