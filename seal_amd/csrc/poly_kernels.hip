// Element-wise RNS polynomial kernels and the key-switching inner product for gfx950.
// All of these are HBM-streaming kernels: one coalesced pass over the operands, 64-bit words,
// grid-stride over a capped grid (256 CUs x 8 workgroups).  See poly_kernels.h for the map to
// the reference functions.
#include "poly_kernels.h"
#include <cstdlib>

namespace sealhip
{
    namespace
    {
        constexpr unsigned kBlock = 256;
#ifndef SEALHIP_EW_GRID_CAP
#define SEALHIP_EW_GRID_CAP 2048
#endif
        inline unsigned grid_for(size_t work)
        {
            size_t b = (work + kBlock - 1) / kBlock;
            if (b > SEALHIP_EW_GRID_CAP)
                b = SEALHIP_EW_GRID_CAP;
            if (b == 0)
                b = 1;
            return (unsigned)b;
        }

        // 128-bit accumulate: (hi:lo) += a*b
        __device__ __forceinline__ void mac128(uint64_t &lo, uint64_t &hi, uint64_t a, uint64_t b)
        {
            uint64_t pl, ph;
            mul_wide(a, b, pl, ph);
            lo += pl;
            hi += ph + (lo < pl);
        }

        // Primes of the double-precision back end (below 2^50, fpd[prime].qi != 0) take the exact-FMA products of field.h: 44 vector
        // instructions per coefficient where the 128-bit Barrett path costs 331 - the kernel sat on the vector ALU and on HBM at
        // once.  Canonical inputs are exact doubles, |a b| < q^2 keeps every residue below 0.875 q (fp_mulmod),
        // the middle sum below 1.75 q takes one fix(); results leave as canonical words: the same bits as the integer path.
        __global__ void __launch_bounds__(kBlock) ckks_multiply_2x2_kernel(
            const ModDesc *mods, const FpDesc *fpd, const uint32_t *comp_prime, const uint64_t *x, const uint64_t *y, uint64_t *out,
            unsigned n_log, unsigned K, size_t plane_words)
        {
            for (size_t i = blockIdx.x * (size_t)kBlock + threadIdx.x; i < plane_words; i += (size_t)gridDim.x * kBlock)
            {
                const unsigned comp = (unsigned)((i >> n_log) % K);
                const unsigned prime = comp_prime ? comp_prime[comp] : comp;
                uint64_t x0 = x[i], x1 = x[plane_words + i];
                uint64_t y0 = y[i], y1 = y[plane_words + i];
                // (the descriptor comes through the scalar cache: the 64 consecutive words of a wave belong to one component whenever
                // N >= 64, and the double-precision back end only exists from N = 2^13 - a per-lane load of it, dependent on the
                // per-lane test of its first word, made the kernel slower than the integer path it replaces: -1.4 % on the step)
                FpDesc f{};
                if (fpd && n_log >= 6)
                    f = ld_uniform_fpd(&fpd[SHL_UNIFORM(prime)]);
                if (f.qi)
                {
                    const double a0 = fp_from_u52(x0), a1 = fp_from_u52(x1), b0 = fp_from_u52(y0), b1 = fp_from_u52(y1);
                    const double r00 = fp_mulmod(a0, b0, f.q, f.qinv), r11 = fp_mulmod(a1, b1, f.q, f.qinv);
                    const double mid = fp_mulmod(a0, b1, f.q, f.qinv) + fp_mulmod(a1, b0, f.q, f.qinv);
                    out[i] = fp_to_canon(r00, f);
                    out[plane_words + i] = fp_to_canon(fp_fix(mid, f.q, f.qinv), f);
                    out[2 * plane_words + i] = fp_to_canon(r11, f);
                    continue;
                }
                const ModDesc md = mods[prime];
                uint64_t lo = 0, hi = 0;
                mac128(lo, hi, x0, y1);
                mac128(lo, hi, x1, y0);
                out[i] = mul_mod(x0, y0, md);
                out[plane_words + i] = barrett128(lo, hi, md);
                out[2 * plane_words + i] = mul_mod(x1, y1, md);
            }
        }

        // The same product in the shape this memory system streams fastest (round 5): ONE 4 KiB chunk per workgroup, two consecutive
        // words (16 bytes) per thread, no grid-stride loop - a float4 copy of that shape moves 6.2 - 6.3 TB/s where the persistent
        // 2048-workgroup loop above moves 4.6 - 5.3 (profiles/r03_microbench_copy_shapes2.txt).  Same arithmetic, same words.
        // words_pairs = plane_words / 2 (plane_words even, N >= 128: the 128 words of a wave belong to one component).
        __global__ void __launch_bounds__(kBlock) ckks_multiply_2x2_wide_kernel(
            const ModDesc *mods, const FpDesc *fpd, const uint32_t *comp_prime, const uint64_t *x, const uint64_t *y, uint64_t *out,
            unsigned n_log, unsigned K, size_t plane_words)
        {
            const size_t p = blockIdx.x * (size_t)kBlock + threadIdx.x; // pair index
            const size_t i = 2 * p;
            if (i >= plane_words)
                return;
            const unsigned comp = (unsigned)((i >> n_log) % K);
            const unsigned prime = comp_prime ? comp_prime[comp] : comp;
            typedef ulonglong2 w2;
            const w2 x0 = *reinterpret_cast<const w2 *>(x + i), x1 = *reinterpret_cast<const w2 *>(x + plane_words + i);
            const w2 y0 = *reinterpret_cast<const w2 *>(y + i), y1 = *reinterpret_cast<const w2 *>(y + plane_words + i);
            uint64_t r0[2], r1[2], r2[2];
            const uint64_t xa[2] = { x0.x, x0.y }, xb[2] = { x1.x, x1.y }, ya[2] = { y0.x, y0.y }, yb[2] = { y1.x, y1.y };
            FpDesc f{};
            if (fpd)
                f = ld_uniform_fpd(&fpd[SHL_UNIFORM(prime)]);
            if (f.qi)
            {
                auto one = [&](uint64_t xa, uint64_t xb, uint64_t ya, uint64_t yb, uint64_t &r0, uint64_t &r1, uint64_t &r2) {
                    const double a0 = fp_from_u52(xa), a1 = fp_from_u52(xb), b0 = fp_from_u52(ya), b1 = fp_from_u52(yb);
                    const double r00 = fp_mulmod(a0, b0, f.q, f.qinv), r11 = fp_mulmod(a1, b1, f.q, f.qinv);
                    const double mid = fp_mulmod(a0, b1, f.q, f.qinv) + fp_mulmod(a1, b0, f.q, f.qinv);
                    r0 = fp_to_canon(r00, f);
                    r1 = fp_to_canon(fp_fix(mid, f.q, f.qinv), f);
                    r2 = fp_to_canon(r11, f);
                };
                one(xa[0], xb[0], ya[0], yb[0], r0[0], r1[0], r2[0]);
                one(xa[1], xb[1], ya[1], yb[1], r0[1], r1[1], r2[1]);
            }
            else
            {
                const ModDesc md = mods[prime];
                auto one = [&](uint64_t xa, uint64_t xb, uint64_t ya, uint64_t yb, uint64_t &r0, uint64_t &r1, uint64_t &r2) {
                    uint64_t lo = 0, hi = 0;
                    mac128(lo, hi, xa, yb);
                    mac128(lo, hi, xb, ya);
                    r0 = mul_mod(xa, ya, md);
                    r1 = barrett128(lo, hi, md);
                    r2 = mul_mod(xb, yb, md);
                };
                one(xa[0], xb[0], ya[0], yb[0], r0[0], r1[0], r2[0]);
                one(xa[1], xb[1], ya[1], yb[1], r0[1], r1[1], r2[1]);
            }
            w2 o0, o1, o2;
            o0.x = r0[0], o0.y = r0[1], o1.x = r1[0], o1.y = r1[1], o2.x = r2[0], o2.y = r2[1];
            *reinterpret_cast<w2 *>(out + i) = o0;
            *reinterpret_cast<w2 *>(out + plane_words + i) = o1;
            *reinterpret_cast<w2 *>(out + 2 * plane_words + i) = o2;
        }

        __global__ void __launch_bounds__(kBlock) multiply_general_kernel(
            const ModDesc *mods, const uint32_t *comp_prime, const uint64_t *x, unsigned sx, const uint64_t *y, unsigned sy,
            uint64_t *out, unsigned n_log, unsigned K, size_t plane_words)
        {
            const unsigned dest = sx + sy - 1;
            for (size_t i = blockIdx.x * (size_t)kBlock + threadIdx.x; i < plane_words; i += (size_t)gridDim.x * kBlock)
            {
                const unsigned comp = (unsigned)((i >> n_log) % K);
                const ModDesc md = mods[comp_prime ? comp_prime[comp] : comp];
                for (unsigned I = 0; I < dest; I++)
                {
                    // a ranges over max(0, I-(sy-1)) .. min(I, sx-1)   (evaluator.cpp:670-681)
                    unsigned a_lo = I > sy - 1 ? I - (sy - 1) : 0;
                    unsigned a_hi = I < sx - 1 ? I : sx - 1;
                    uint64_t acc = 0;
                    for (unsigned a = a_lo; a <= a_hi; a++)
                        acc = add_mod(acc, mul_mod(x[a * plane_words + i], y[(I - a) * plane_words + i], md), md.q);
                    out[I * plane_words + i] = acc;
                }
            }
        }

        __global__ void __launch_bounds__(kBlock) dyadic_kernel(
            const ModDesc *mods, const uint64_t *a, const uint64_t *b, uint64_t *r, unsigned n_log, unsigned comps,
            unsigned first_prime, size_t words)
        {
            for (size_t i = blockIdx.x * (size_t)kBlock + threadIdx.x; i < words; i += (size_t)gridDim.x * kBlock)
            {
                const unsigned comp = (unsigned)((i >> n_log) % comps);
                r[i] = mul_mod(a[i], b[i], mods[first_prime + comp]);
            }
        }

        __global__ void __launch_bounds__(kBlock) addsub_kernel(
            const ModDesc *mods, const uint64_t *a, const uint64_t *b, uint64_t *r, int op, unsigned n_log, unsigned K,
            size_t words)
        {
            for (size_t i = blockIdx.x * (size_t)kBlock + threadIdx.x; i < words; i += (size_t)gridDim.x * kBlock)
            {
                const unsigned comp = (unsigned)((i >> n_log) % K);
                const uint64_t q = mods[comp].q;
                uint64_t v;
                if (op == 0)
                    v = add_mod(a[i], b[i], q);
                else if (op == 1)
                    v = sub_mod(a[i], b[i], q);
                else
                    v = neg_mod(a[i], q);
                r[i] = v;
            }
        }

        // NTT-domain automorphism: result[i] = operand[T[i]],
        // T[i] = bitrev_n(((elt * bitrev_{n+1}(i + N)) >> 1) & (N-1))   (galois.cpp:18-51), computed
        // on the fly (v_bfrev_b32) instead of read from a table.
        __global__ void __launch_bounds__(kBlock) galois_ntt_kernel(
            const uint64_t *in, uint64_t *out, uint32_t elt, unsigned n_log, size_t words)
        {
            const unsigned N = 1u << n_log;
            for (size_t i = blockIdx.x * (size_t)kBlock + threadIdx.x; i < words; i += (size_t)gridDim.x * kBlock)
            {
                const unsigned j = (unsigned)(i & (N - 1));
                const size_t base = i - j;
                unsigned rev = __brev(j + N) >> (32 - (n_log + 1));
                uint64_t raw = ((uint64_t)elt * rev) >> 1;
                unsigned idx = (unsigned)raw & (N - 1);
                unsigned src = n_log ? (__brev(idx) >> (32 - n_log)) : 0;
                out[i] = in[base + src];
            }
        }

        // Coefficient-domain automorphism: result[(i*elt) mod N] = +-operand[i]  (galois.cpp:148-190)
        __global__ void __launch_bounds__(kBlock) galois_coeff_kernel(
            const ModDesc *mods, const uint64_t *in, uint64_t *out, uint32_t elt, unsigned n_log, unsigned K, size_t words)
        {
            const unsigned N = 1u << n_log;
            for (size_t i = blockIdx.x * (size_t)kBlock + threadIdx.x; i < words; i += (size_t)gridDim.x * kBlock)
            {
                const unsigned j = (unsigned)(i & (N - 1));
                const size_t base = i - j;
                const unsigned comp = (unsigned)((i >> n_log) % K);
                const uint64_t q = mods[comp].q;
                uint64_t raw = (uint64_t)j * elt;
                unsigned idx = (unsigned)raw & (N - 1);
                uint64_t v = in[i];
                if ((raw >> n_log) & 1)
                    v = neg_mod(v, q);
                out[base + idx] = v;
            }
        }

        __global__ void __launch_bounds__(kBlock) rescale_combine_kernel(
            const ModDesc *mods, const ShoupOp *inv_q_last, const uint64_t *c, const uint64_t *t, uint64_t *out,
            unsigned n_log, unsigned K, size_t out_words)
        {
            const unsigned Km1 = K - 1;
            for (size_t i = blockIdx.x * (size_t)kBlock + threadIdx.x; i < out_words; i += (size_t)gridDim.x * kBlock)
            {
                const size_t row = i >> n_log; // item*(K-1) + comp
                const unsigned comp = (unsigned)(row % Km1);
                const size_t item = row / Km1;
                const size_t j = i & ((size_t(1) << n_log) - 1);
                const uint64_t q = mods[comp].q;
                const ShoupOp iq = inv_q_last[comp];
                uint64_t cv = c[((item * K + comp) << n_log) + j];
                uint64_t tv = t[i];
                // c in [0,q), t in [0,4q): c + 4q - t in (0, 5q)
                out[i] = mul_shoup(cv + 4 * q - tv, iq.w, iq.wq, q);
            }
        }

        __global__ void __launch_bounds__(kBlock) dyadic_plain_kernel(
            const ModDesc *mods, const uint64_t *a, const uint64_t *p, uint64_t *r, unsigned n_log, unsigned K, size_t words)
        {
            const size_t pw = (size_t)K << n_log;
            for (size_t i = blockIdx.x * (size_t)kBlock + threadIdx.x; i < words; i += (size_t)gridDim.x * kBlock)
            {
                const size_t pi = i % pw;
                r[i] = mul_mod(a[i], p[pi], mods[pi >> n_log]);
            }
        }
        __global__ void __launch_bounds__(kBlock) addsub_plain_kernel(
            const ModDesc *mods, uint64_t *c, const uint64_t *p, int op, unsigned n_log, unsigned K, size_t words)
        {
            const size_t pw = (size_t)K << n_log;
            for (size_t i = blockIdx.x * (size_t)kBlock + threadIdx.x; i < words; i += (size_t)gridDim.x * kBlock)
            {
                const size_t pi = i % pw;
                const uint64_t q = mods[pi >> n_log].q;
                c[i] = op ? sub_mod(c[i], p[pi], q) : add_mod(c[i], p[pi], q);
            }
        }
        __global__ void __launch_bounds__(kBlock) plain_lift_kernel(
            const ModDesc *mods, ModDesc t, const uint64_t *m, size_t coeff_count, uint64_t scale_by, uint64_t threshold,
            const uint64_t *inc, uint64_t *out, unsigned n_log, unsigned K)
        {
            const size_t N = size_t(1) << n_log, words = (size_t)K << n_log;
            for (size_t i = blockIdx.x * (size_t)kBlock + threadIdx.x; i < words; i += (size_t)gridDim.x * kBlock)
            {
                const size_t j = i & (N - 1);
                const unsigned k = (unsigned)(i >> n_log);
                uint64_t v = 0;
                if (j < coeff_count)
                {
                    uint64_t mv = m[j];
                    if (scale_by != 1)
                        mv = mul_mod(mv, scale_by, t);
                    const ModDesc md = mods[k];
                    v = barrett64(mv, md);
                    if (mv >= threshold)
                        v = add_mod(v, inc[k], md.q);
                }
                out[i] = v;
            }
        }
        __global__ void __launch_bounds__(kBlock) plain_stats_kernel(const uint64_t *m, size_t count, uint64_t *stats)
        {
            // one workgroup; stats[0] = number of nonzero coefficients, stats[1] = index of the last one + 1 (0 if none),
            // stats[2] = that coefficient
            __shared__ unsigned long long s_nz[kBlock], s_last[kBlock];
            unsigned long long nz = 0, last = 0;
            for (size_t i = threadIdx.x; i < count; i += kBlock)
                if (m[i])
                {
                    nz++;
                    last = i + 1;
                }
            s_nz[threadIdx.x] = nz;
            s_last[threadIdx.x] = last;
            __syncthreads();
            if (threadIdx.x == 0)
            {
                for (unsigned k = 1; k < kBlock; k++)
                {
                    nz += s_nz[k];
                    last = s_last[k] > last ? s_last[k] : last;
                }
                stats[0] = nz;
                stats[1] = last;
                stats[2] = last ? m[last - 1] : 0;
            }
        }
        __global__ void __launch_bounds__(kBlock) negacyclic_mul_mono_kernel(
            const ModDesc *mods, const uint64_t *in, uint64_t *out, const uint64_t *scalars, size_t e, unsigned n_log, unsigned K,
            size_t words)
        {
            const size_t N = size_t(1) << n_log;
            for (size_t i = blockIdx.x * (size_t)kBlock + threadIdx.x; i < words; i += (size_t)gridDim.x * kBlock)
            {
                const size_t j = i & (N - 1), row = i >> n_log;
                const unsigned k = (unsigned)(row % K);
                const ModDesc md = mods[k];
                uint64_t v = mul_mod(in[i], scalars[k], md);
                const size_t idx = j + e; // e < N
                if (idx >= N)
                    v = neg_mod(v, md.q);
                out[(row << n_log) + (idx & (N - 1))] = v;
            }
        }
        // floor((hi:lo) / t) for a quotient below 2^64, with t's Barrett constant floor(2^128 / t)
        __device__ __forceinline__ uint64_t div128_by(uint64_t lo, uint64_t hi, const ModDesc &t)
        {
            uint64_t t1 = mul_hi64(lo, t.ratio_lo);
            uint64_t a_lo, a_hi, b_lo, b_hi;
            mul_wide(lo, t.ratio_hi, a_lo, a_hi);
            mul_wide(hi, t.ratio_lo, b_lo, b_hi);
            uint64_t mid = t1 + a_lo;
            uint64_t c = mid < t1;
            uint64_t mid2 = mid + b_lo;
            c += mid2 < mid;
            uint64_t qest = hi * t.ratio_hi + a_hi + b_hi + c; // low by at most 2
            uint64_t r = lo - qest * t.q;
            while (r >= t.q)
            {
                r -= t.q;
                qest++;
            }
            return qest;
        }
        __global__ void __launch_bounds__(kBlock) bfv_addsub_plain_kernel(
            const ModDesc *mods, ModDesc t, const uint64_t *m, size_t coeff_count, uint64_t q_mod_t, uint64_t threshold,
            const uint64_t *delta, uint64_t *c0, int op, unsigned n_log, unsigned K, size_t words)
        {
            const size_t N = size_t(1) << n_log;
            for (size_t i = blockIdx.x * (size_t)kBlock + threadIdx.x; i < words; i += (size_t)gridDim.x * kBlock)
            {
                const size_t j = i & (N - 1);
                if (j >= coeff_count)
                    continue;
                const unsigned k = (unsigned)((i >> n_log) % K);
                const ModDesc md = mods[k];
                const uint64_t mv = m[j];
                uint64_t lo, hi;
                mul_wide(mv, q_mod_t, lo, hi);
                lo += threshold;
                hi += lo < threshold;
                const uint64_t fix = div128_by(lo, hi, t);
                const uint64_t scaled = add_mod(mul_mod(mv, delta[k], md), barrett64(fix, md), md.q);
                c0[i] = op ? sub_mod(c0[i], scaled, md.q) : add_mod(c0[i], scaled, md.q);
            }
        }

        __global__ void __launch_bounds__(kBlock) mul_scalar_kernel(
            const ModDesc *mods, const uint64_t *a, uint64_t *r, uint64_t scalar, unsigned n_log, unsigned K, size_t words)
        {
            for (size_t i = blockIdx.x * (size_t)kBlock + threadIdx.x; i < words; i += (size_t)gridDim.x * kBlock)
            {
                const ModDesc md = mods[(i >> n_log) % K];
                r[i] = mul_mod(a[i], barrett64(scalar, md), md);
            }
        }

        __global__ void __launch_bounds__(kBlock) bgv_delta_kernel(
            const ModDesc *mods, ModDesc t, uint64_t inv_t, const uint64_t *q_last_mod_q, const uint64_t *c_last, size_t c_stride,
            uint64_t *delta, unsigned n_log, unsigned ncomp, size_t out_words)
        {
            for (size_t i = blockIdx.x * (size_t)kBlock + threadIdx.x; i < out_words; i += (size_t)gridDim.x * kBlock)
            {
                const size_t row = i >> n_log; // item*ncomp + comp
                const unsigned comp = (unsigned)(row % ncomp);
                const size_t item = row / ncomp;
                const size_t j = i & ((size_t(1) << n_log) - 1);
                const ModDesc md = mods[comp];
                const uint64_t c = c_last[item * c_stride + j];
                uint64_t k = neg_mod(barrett64(c, t), t.q);
                if (inv_t != 1)
                    k = mul_mod(k, inv_t, t);
                const uint64_t d = mul_mod(barrett64(k, md), q_last_mod_q[comp], md);
                delta[i] = add_mod(d, barrett64(c, md), md.q);
            }
        }

        __global__ void __launch_bounds__(kBlock) bfv_modswitch_kernel(
            const ModDesc *mods, const ShoupOp *inv_q_last, const uint64_t *half_mod_q, uint64_t q_last, uint64_t half,
            const uint64_t *c, uint64_t *out, unsigned n_log, unsigned K, size_t out_words)
        {
            const unsigned Km1 = K - 1;
            for (size_t i = blockIdx.x * (size_t)kBlock + threadIdx.x; i < out_words; i += (size_t)gridDim.x * kBlock)
            {
                const size_t row = i >> n_log;
                const unsigned comp = (unsigned)(row % Km1);
                const size_t item = row / Km1;
                const size_t j = i & ((size_t(1) << n_log) - 1);
                const ModDesc md = mods[comp];
                const ShoupOp iq = inv_q_last[comp];
                uint64_t last = c[((item * K + Km1) << n_log) + j];
                uint64_t r = csub(last + half, q_last);                 // (c_last + half) mod q_last
                uint64_t u = sub_mod(barrett64(r, md), half_mod_q[comp], md.q);
                uint64_t cv = c[((item * K + comp) << n_log) + j];
                out[i] = mul_shoup(sub_mod(cv, u, md.q), iq.w, iq.wq, md.q);
            }
        }

        __global__ void __launch_bounds__(kBlock) drop_last_kernel(
            const uint64_t *c, uint64_t *out, unsigned n_log, unsigned K, size_t out_words)
        {
            const unsigned Km1 = K - 1;
            for (size_t i = blockIdx.x * (size_t)kBlock + threadIdx.x; i < out_words; i += (size_t)gridDim.x * kBlock)
            {
                const size_t row = i >> n_log;
                const unsigned comp = (unsigned)(row % Km1);
                const size_t item = row / Km1;
                const size_t j = i & ((size_t(1) << n_log) - 1);
                out[i] = c[((item * K + comp) << n_log) + j];
            }
        }

        // acc[b][k][I][j] = sum_J u[b][I][J][j] * key[J][k][comp(I)][j]  mod q_I, 128-bit lazy sum.
        // One thread = one coefficient j of one target modulus I for KS_BI batch items: the key words
        // are loaded once and reused across the batch items (the key is the dominant HBM stream).
        constexpr unsigned KS_BI = 4;
        __global__ void __launch_bounds__(kBlock) keyswitch_mac_kernel(
            const ModDesc *mods, const uint64_t *u, const uint64_t *key, uint64_t *acc, unsigned n_log, unsigned K,
            unsigned L, unsigned batch, unsigned j0, unsigned j1, unsigned key_digit0)
        {
            const size_t N = size_t(1) << n_log;
            const unsigned I = blockIdx.y;
            const unsigned b0 = blockIdx.z * KS_BI;
            const unsigned kc = (I == K) ? L - 1 : I; // key component / pool prime of target modulus I
            const ModDesc md = mods[kc];
            const unsigned nb = (batch - b0) < KS_BI ? (batch - b0) : KS_BI;
            for (size_t j = blockIdx.x * (size_t)kBlock + threadIdx.x; j < N; j += (size_t)gridDim.x * kBlock)
            {
                uint64_t lo[KS_BI][2], hi[KS_BI][2];
#pragma unroll
                for (unsigned bi = 0; bi < KS_BI; bi++)
                    lo[bi][0] = lo[bi][1] = hi[bi][0] = hi[bi][1] = 0;
                for (unsigned J = j0; J < j1; J++)
                {
                    const uint64_t k0 = key[(((size_t)(J - key_digit0) * 2 + 0) * L + kc) * N + j];
                    const uint64_t k1 = key[(((size_t)(J - key_digit0) * 2 + 1) * L + kc) * N + j];
#pragma unroll
                    for (unsigned bi = 0; bi < KS_BI; bi++)
                    {
                        if (bi < nb)
                        {
                            const uint64_t uv = u[((((size_t)(b0 + bi)) * (K + 1) + I) * K + J) * N + j];
                            mac128(lo[bi][0], hi[bi][0], uv, k0);
                            mac128(lo[bi][1], hi[bi][1], uv, k1);
                        }
                    }
                }
#pragma unroll
                for (unsigned bi = 0; bi < KS_BI; bi++)
                {
                    if (bi < nb)
                    {
                        acc[((((size_t)(b0 + bi)) * 2 + 0) * (K + 1) + I) * N + j] = barrett128(lo[bi][0], hi[bi][0], md);
                        acc[((((size_t)(b0 + bi)) * 2 + 1) * (K + 1) + I) * N + j] = barrett128(lo[bi][1], hi[bi][1], md);
                    }
                }
            }
        }

        // acc[item][I][j] <- acc mod q_I: the sum of `parts` canonical partial sums (digit-parallel key switching)
        // `local_parts` > 1: the summands are still separate buffers of `words` words each (in-launch digit groups of the
        // fused key switch at small batches); they are added here
        // c0 != null: the data-prime components leave as c_k + S_k P^-1 (pm[I] = P^-1 mod q_I), the form the fused key switch writes
        // itself when its digits run as one group (ntt2_kernels.h: KsFusedArgs::fold_c0); c_k = [batch][K][N]
        __global__ void __launch_bounds__(kBlock) keyswitch_reduce_kernel(
            const ModDesc *mods, const uint64_t *acc, unsigned n_log, unsigned K, unsigned L, size_t words, unsigned local_parts, const uint64_t *c0,
            const uint64_t *c1, const ShoupOp *pm, uint64_t *out)
        {
            const size_t N = size_t(1) << n_log;
            for (size_t i = blockIdx.x * (size_t)kBlock + threadIdx.x; i < words; i += (size_t)gridDim.x * kBlock)
            {
                const size_t poly = i >> n_log; // (item * 2 + k) * (K + 1) + I
                const unsigned I = (unsigned)(poly % (K + 1));
                uint64_t v = acc[i];
                for (unsigned g = 1; g < local_parts; g++)
                    v += acc[i + g * words];
                const ModDesc md = mods[I == K ? L - 1 : I];
                v = barrett64(v, md);
                if (c0 && I < K)
                {
                    const size_t pk = poly / (K + 1), item = pk >> 1;
                    const uint64_t *cp = (pk & 1) ? c1 : c0; // c1 == null: the second polynomial's addend is zero (a rotation's pi(c0), 0)
                    const uint64_t c = cp ? cp[((item * K + I) << n_log) + (i & (N - 1))] : 0;
                    const ShoupOp p = pm[I];
                    v = add_mod(mul_shoup(v, p.w, p.wq, md.q), c, md.q);
                }
                out[i] = v;
            }
        }

        // Folded key-switch + rescale tail (Evaluator::switch_key_finish_rescale): coefficient form of the relinearised ciphertext's
        // LAST component from the coefficient forms of its two ingredients.  The component is (c + S P^-1) - NTT(v) P^-1 with
        // v = the mod-down correction (evaluator.cpp:2813-2832); the inverse transform is linear, so its coefficient form is
        // w - v P^-1 with w = INTT(c + S P^-1): no forward transform of v.  In place over w, plus q_last / 2 (the rescale's rounding
        // addend, rns.cpp:858-862).  acc[outer][K+1][N]: component K - 1 = w, component K = t_P + P/2 mod P.
        __global__ void __launch_bounds__(kBlock) ks_last_coeff_kernel(
            const ModDesc *mods, unsigned prime, const ShoupOp *pinv, const uint64_t *fix, uint64_t half_last, uint64_t *acc, unsigned n_log,
            unsigned K, size_t words)
        {
            const ModDesc md = mods[prime];
            const ShoupOp pm = *pinv;
            const uint64_t f = *fix;
            const size_t N = size_t(1) << n_log;
            for (size_t i = blockIdx.x * (size_t)kBlock + threadIdx.x; i < words; i += (size_t)gridDim.x * kBlock)
            {
                const size_t outer = i >> n_log, j = i & (N - 1);
                uint64_t *w = acc + ((outer * (K + 1) + (K - 1)) << n_log) + j;
                const uint64_t v = barrett64(w[N], md) + f; // below 2 q
                const uint64_t y = sub_mod(*w, mul_shoup(v, pm.w, pm.wq, md.q), md.q);
                *w = csub(y + half_last, md.q);
            }
        }

        // ---- reduce-scatter exchange of the digit-parallel key switch.  The K data moduli are owned by the G
        // ranks in contiguous ranges (rank c: [c*base + min(c, extra), ...), sizes differ by at most one; m = ceil(K / G) slots).
        __device__ __forceinline__ unsigned ks_owner_first(unsigned K, unsigned G, unsigned c)
        {
            const unsigned base = K / G, extra = K % G;
            return c * base + (c < extra ? c : extra);
        }
        // send[c][s][b][k][j] = acc[b][k][first(c) + s][j] (zero beyond rank c's range);  sp[b][k][j] = acc[b][k][K][j]
        __global__ void __launch_bounds__(kBlock) ks_pack_targets_kernel(
            const uint64_t *acc, uint64_t *send, uint64_t *sp, unsigned n_log, unsigned K, unsigned G, unsigned m, unsigned batch,
            size_t send_words, size_t sp_words)
        {
            const size_t N = size_t(1) << n_log;
            for (size_t i = blockIdx.x * (size_t)kBlock + threadIdx.x; i < send_words + sp_words; i += (size_t)gridDim.x * kBlock)
            {
                if (i < send_words)
                {
                    const size_t j = i & (N - 1);
                    size_t r = i >> n_log; // ((c*m + s)*batch + b)*2 + k
                    const unsigned k = (unsigned)(r & 1);
                    r >>= 1;
                    const size_t b = r % batch;
                    r /= batch;
                    const unsigned s = (unsigned)(r % m), c = (unsigned)(r / m);
                    const unsigned first = ks_owner_first(K, G, c), count = ks_owner_first(K, G, c + 1) - first;
                    send[i] = s < count ? acc[(((b * 2 + k) * (K + 1) + first + s) << n_log) + j] : 0;
                }
                else
                {
                    const size_t t = i - send_words, j = t & (N - 1), r = t >> n_log; // b*2 + k
                    sp[t] = acc[((r * (K + 1) + K) << n_log) + j];
                }
            }
        }
        // the sums this rank received, in the layout of a key switch over its `count` moduli:
        //   acc3[b][k][s][j] = recv[s][b][k][j] mod q_{first+s} (s < count),  acc3[b][k][count][j] = sp[b][k][j] mod P
        __global__ void __launch_bounds__(kBlock) ks_unpack_owned_kernel(
            const ModDesc *mods, const uint64_t *recv, const uint64_t *sp, uint64_t *acc3, unsigned n_log, unsigned L, unsigned first,
            unsigned count, unsigned batch, size_t own_words, size_t sp_words)
        {
            const size_t N = size_t(1) << n_log;
            for (size_t i = blockIdx.x * (size_t)kBlock + threadIdx.x; i < own_words + sp_words; i += (size_t)gridDim.x * kBlock)
            {
                if (i < own_words)
                {
                    const size_t j = i & (N - 1);
                    size_t r = i >> n_log; // (s*batch + b)*2 + k
                    const unsigned k = (unsigned)(r & 1);
                    r >>= 1;
                    const size_t b = r % batch;
                    const unsigned s = (unsigned)(r / batch); // < count
                    acc3[(((b * 2 + k) * (count + 1) + s) << n_log) + j] = barrett64(recv[i], mods[first + s]);
                }
                else
                {
                    const size_t t = i - own_words, j = t & (N - 1), r = t >> n_log;
                    acc3[((r * (count + 1) + count) << n_log) + j] = barrett64(sp[t], mods[L - 1]);
                }
            }
        }
        // own[s][b][k][j] = inc[k][b][s][j] (zero for the padding slots s >= count)
        __global__ void __launch_bounds__(kBlock) ks_pack_owned_kernel(
            const uint64_t *inc, uint64_t *own, unsigned n_log, unsigned count, unsigned m, unsigned batch, size_t words)
        {
            const size_t N = size_t(1) << n_log;
            for (size_t i = blockIdx.x * (size_t)kBlock + threadIdx.x; i < words; i += (size_t)gridDim.x * kBlock)
            {
                const size_t j = i & (N - 1);
                size_t r = i >> n_log;
                const unsigned k = (unsigned)(r & 1);
                r >>= 1;
                const size_t b = r % batch;
                const unsigned s = (unsigned)(r / batch);
                own[i] = s < count ? inc[(((size_t)k * batch + b) * count + s) * N + j] : 0;
            }
        }
        // ct_k[b][i][j] += all[owner(i)][slot(i)][b][k][j]  (mod q_i)
        __global__ void __launch_bounds__(kBlock) ks_add_gathered_kernel(
            const ModDesc *mods, uint64_t *ct0, uint64_t *ct1, const uint64_t *all, unsigned n_log, unsigned K, unsigned G, unsigned m,
            unsigned batch, size_t words /* batch*K*N */)
        {
            const size_t N = size_t(1) << n_log;
            for (size_t i = blockIdx.x * (size_t)kBlock + threadIdx.x; i < words; i += (size_t)gridDim.x * kBlock)
            {
                const size_t j = i & (N - 1), row = i >> n_log; // b*K + comp
                const unsigned comp = (unsigned)(row % K);
                const size_t b = row / K;
                unsigned c = 0;
                while (ks_owner_first(K, G, c + 1) <= comp)
                    c++;
                const unsigned s = comp - ks_owner_first(K, G, c);
                const uint64_t q = mods[comp].q;
                const size_t base = ((((size_t)c * m + s) * batch + b) * 2) << n_log;
                ct0[i] = add_mod(ct0[i], all[base + j], q);
                ct1[i] = add_mod(ct1[i], all[base + N + j], q);
            }
        }

        __global__ void __launch_bounds__(kBlock) keyswitch_tail_ckks_kernel(
            const ModDesc *mods, const ShoupOp *inv_p, uint64_t *ct0, uint64_t *ct1, const uint64_t *acc,
            const uint64_t *t, unsigned n_log, unsigned K, size_t words /* batch*K*N */)
        {
            for (size_t i = blockIdx.x * (size_t)kBlock + threadIdx.x; i < words; i += (size_t)gridDim.x * kBlock)
            {
                const size_t row = i >> n_log; // b*K + comp
                const unsigned comp = (unsigned)(row % K);
                const size_t b = row / K;
                const size_t j = i & ((size_t(1) << n_log) - 1);
                const uint64_t q = mods[comp].q;
                const ShoupOp ip = inv_p[comp];
#pragma unroll
                for (unsigned k = 0; k < 2; k++)
                {
                    uint64_t a = acc[(((b * 2 + k) * (K + 1) + comp) << n_log) + j];
                    uint64_t tv = t[(((b * 2 + k) * K + comp) << n_log) + j];
                    uint64_t v = mul_shoup(a + 4 * q - tv, ip.w, ip.wq, q);
                    uint64_t *ct = k ? ct1 : ct0;
                    ct[i] = add_mod(ct[i], v, q);
                }
            }
        }

        __global__ void __launch_bounds__(kBlock) keyswitch_tail_bfv_kernel(
            const ModDesc *mods, const ShoupOp *inv_p, const uint64_t *round_fix, uint64_t half_p, uint64_t p,
            uint64_t *ct0, uint64_t *ct1, const uint64_t *acc, unsigned n_log, unsigned K, size_t words)
        {
            for (size_t i = blockIdx.x * (size_t)kBlock + threadIdx.x; i < words; i += (size_t)gridDim.x * kBlock)
            {
                const size_t row = i >> n_log;
                const unsigned comp = (unsigned)(row % K);
                const size_t b = row / K;
                const size_t j = i & ((size_t(1) << n_log) - 1);
                const ModDesc md = mods[comp];
                const ShoupOp ip = inv_p[comp];
#pragma unroll
                for (unsigned k = 0; k < 2; k++)
                {
                    uint64_t a = acc[(((b * 2 + k) * (K + 1) + comp) << n_log) + j];
                    uint64_t r = acc[(((b * 2 + k) * (K + 1) + K) << n_log) + j];
                    uint64_t s = csub(r + half_p, p);
                    uint64_t uu = barrett64(s, md) + round_fix[comp]; // (s mod q_i) - (half mod q_i) + q_i, in [1, 2q)
                    uint64_t v = mul_shoup(a + 2 * md.q - uu, ip.w, ip.wq, md.q);
                    uint64_t *ct = k ? ct1 : ct0;
                    ct[i] = add_mod(ct[i], v, md.q);
                }
            }
        }

        // BFV: the key switch's mod-down by the special prime P and the mod-switch that follows it in one pass (round 4).  A thread
        // owns coefficient j of one polynomial of one item: it completes the LAST component first (the mod-switch divides by it), then
        // every other component - c' = c + (S - v) P^-1 as keyswitch_tail_bfv_kernel, out = (c' - u) q_last^-1 as bfv_modswitch_kernel -
        // so the relinearised ciphertext is never written (evaluator.cpp:2806-2864 then rns.cpp:789-828 on its result: same words).
        __global__ void __launch_bounds__(kBlock) keyswitch_tail_modswitch_bfv_kernel(
            const ModDesc *mods, const ShoupOp *inv_p, const uint64_t *round_fix_p, uint64_t half_p, uint64_t p, const ShoupOp *inv_q_last,
            const uint64_t *half_mod_q, uint64_t q_last, uint64_t half_q_last, const uint64_t *ct0, const uint64_t *ct1, const uint64_t *acc,
            uint64_t *out, unsigned n_log, unsigned K, unsigned batch, size_t threads)
        {
            const unsigned Km1 = K - 1;
            const size_t n_mask = (size_t(1) << n_log) - 1;
            for (size_t i = blockIdx.x * (size_t)kBlock + threadIdx.x; i < threads; i += (size_t)gridDim.x * kBlock)
            {
                const size_t j = i & n_mask, bk = i >> n_log; // bk = k * batch + b: all items of polynomial 0, then of polynomial 1
                const size_t b = bk % batch;
                const unsigned k = (unsigned)(bk / batch);
                const uint64_t *ct = (k ? ct1 : ct0) + ((b * K) << n_log) + j;
                const uint64_t *a = acc + (((b * 2 + k) * (K + 1)) << n_log) + j;
                uint64_t *o = out + (((size_t)k * batch + b) * Km1 << n_log) + j;
                const uint64_t s = csub(a[(size_t)K << n_log] + half_p, p); // (r + P/2) mod P
                auto completed = [&](unsigned comp) {
                    const ModDesc md = mods[comp];
                    const ShoupOp ip = inv_p[comp];
                    const uint64_t uu = barrett64(s, md) + round_fix_p[comp]; // (s mod q_i) - (P/2 mod q_i) + q_i, in [1, 2q)
                    const uint64_t v = mul_shoup(a[(size_t)comp << n_log] + 2 * md.q - uu, ip.w, ip.wq, md.q);
                    return add_mod(ct[(size_t)comp << n_log], v, md.q);
                };
                const uint64_t rl = csub(completed(Km1) + half_q_last, q_last); // (c'_last + q_last/2) mod q_last
                for (unsigned comp = 0; comp < Km1; comp++)
                {
                    const ModDesc md = mods[comp];
                    const ShoupOp iq = inv_q_last[comp];
                    const uint64_t u = sub_mod(barrett64(rl, md), half_mod_q[comp], md.q);
                    o[(size_t)comp << n_log] = mul_shoup(sub_mod(completed(comp), u, md.q), iq.w, iq.wq, md.q);
                }
            }
        }

        __global__ void __launch_bounds__(kBlock) any_nonzero_kernel(const uint64_t *data, size_t words, unsigned *flag)
        {
            unsigned nz = 0;
            for (size_t i = blockIdx.x * (size_t)kBlock + threadIdx.x; i < words; i += (size_t)gridDim.x * kBlock)
                nz |= data[i] != 0;
            if (nz)
                *flag = 1;
        }
    } // namespace

    hipError_t k_ckks_multiply_2x2(
        const ModDesc *mods, const FpDesc *fpd, const uint32_t *comp_prime, const uint64_t *x, const uint64_t *y, uint64_t *out, PlaneGeom g,
        hipStream_t s)
    {
        size_t w = g.words();
#ifndef SEALHIP_TENSOR_WIDE
#define SEALHIP_TENSOR_WIDE 1
#endif
        // large batches: one 4 KiB chunk per workgroup, 16 bytes per thread (ckks_multiply_2x2_wide_kernel); the grid-stride kernel keeps
        // the small ones, where a launch of a few workgroups is all there is
        const size_t pairs = w / 2, wide_blocks = (pairs + kBlock - 1) / kBlock;
        static const char *min_env = shl_ab_getenv("SEALHIP_TENSOR_WIDE_MIN"); // development / emulated builds: take the wide kernel from this many workgroups on
        const size_t wide_min = min_env ? (size_t)std::atol(min_env) : 2048;
        if (SEALHIP_TENSOR_WIDE && g.n_log >= 7 && w % 2 == 0 && wide_blocks > wide_min && wide_blocks < (size_t)0x7fffffff &&
            ((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 16 == 0) && ((uintptr_t)out % 16 == 0))
        {
            hipLaunchKernelGGL(ckks_multiply_2x2_wide_kernel, dim3((unsigned)wide_blocks), dim3(kBlock), 0, s, mods, fpd, comp_prime, x, y, out,
                               g.n_log, g.K, w);
            return hipGetLastError();
        }
        hipLaunchKernelGGL(
            ckks_multiply_2x2_kernel, dim3(grid_for(w)), dim3(kBlock), 0, s, mods, fpd, comp_prime, x, y, out, g.n_log, g.K, w);
        return hipGetLastError();
    }
    hipError_t k_multiply_general(
        const ModDesc *mods, const uint32_t *comp_prime, const uint64_t *x, unsigned sx, const uint64_t *y, unsigned sy,
        uint64_t *out, PlaneGeom g, hipStream_t s)
    {
        size_t w = g.words();
        hipLaunchKernelGGL(
            multiply_general_kernel, dim3(grid_for(w)), dim3(kBlock), 0, s, mods, comp_prime, x, sx, y, sy, out, g.n_log,
            g.K, w);
        return hipGetLastError();
    }
    hipError_t k_dyadic(
        const ModDesc *mods, const uint64_t *a, const uint64_t *b, uint64_t *r, unsigned n_log, unsigned comps,
        unsigned first_prime, size_t polys, hipStream_t s)
    {
        size_t w = (polys * comps) << n_log;
        hipLaunchKernelGGL(dyadic_kernel, dim3(grid_for(w)), dim3(kBlock), 0, s, mods, a, b, r, n_log, comps, first_prime, w);
        return hipGetLastError();
    }
    hipError_t k_addsub(
        const ModDesc *mods, const uint64_t *a, const uint64_t *b, uint64_t *r, int op, PlaneGeom g, unsigned planes,
        hipStream_t s)
    {
        size_t w = g.words() * planes;
        hipLaunchKernelGGL(addsub_kernel, dim3(grid_for(w)), dim3(kBlock), 0, s, mods, a, b, r, op, g.n_log, g.K, w);
        return hipGetLastError();
    }
    hipError_t k_apply_galois(
        const ModDesc *mods, const uint64_t *in, uint64_t *out, uint32_t elt, int ntt_form, PlaneGeom g, unsigned planes,
        hipStream_t s)
    {
        size_t w = g.words() * planes;
        if (ntt_form)
            hipLaunchKernelGGL(galois_ntt_kernel, dim3(grid_for(w)), dim3(kBlock), 0, s, in, out, elt, g.n_log, w);
        else
            hipLaunchKernelGGL(galois_coeff_kernel, dim3(grid_for(w)), dim3(kBlock), 0, s, mods, in, out, elt, g.n_log, g.K, w);
        return hipGetLastError();
    }
    hipError_t k_rescale_combine(
        const ModDesc *mods, const ShoupOp *inv_q_last, const uint64_t *c, const uint64_t *t, uint64_t *out,
        unsigned n_log, unsigned K, size_t items, hipStream_t s)
    {
        size_t w = (items * (K - 1)) << n_log;
        if (!w)
            return hipSuccess;
        hipLaunchKernelGGL(rescale_combine_kernel, dim3(grid_for(w)), dim3(kBlock), 0, s, mods, inv_q_last, c, t, out, n_log, K, w);
        return hipGetLastError();
    }
    hipError_t k_dyadic_plain(const ModDesc *mods, const uint64_t *a, const uint64_t *p, uint64_t *r, unsigned n_log, unsigned K,
                              size_t items, hipStream_t s)
    {
        size_t w = (items * K) << n_log;
        if (!w)
            return hipSuccess;
        hipLaunchKernelGGL(dyadic_plain_kernel, dim3(grid_for(w)), dim3(kBlock), 0, s, mods, a, p, r, n_log, K, w);
        return hipGetLastError();
    }
    hipError_t k_addsub_plain(const ModDesc *mods, uint64_t *c, const uint64_t *p, int op, unsigned n_log, unsigned K, size_t items,
                              hipStream_t s)
    {
        size_t w = (items * K) << n_log;
        if (!w)
            return hipSuccess;
        hipLaunchKernelGGL(addsub_plain_kernel, dim3(grid_for(w)), dim3(kBlock), 0, s, mods, c, p, op, n_log, K, w);
        return hipGetLastError();
    }
    hipError_t k_plain_lift(const ModDesc *mods, ModDesc t, const uint64_t *m, size_t coeff_count, uint64_t scale_by,
                            uint64_t threshold, const uint64_t *upper_half_inc, uint64_t *out, unsigned n_log, unsigned K, hipStream_t s)
    {
        size_t w = (size_t)K << n_log;
        hipLaunchKernelGGL(plain_lift_kernel, dim3(grid_for(w)), dim3(kBlock), 0, s, mods, t, m, coeff_count, scale_by, threshold,
                           upper_half_inc, out, n_log, K);
        return hipGetLastError();
    }
    hipError_t k_plain_stats(const uint64_t *m, size_t coeff_count, uint64_t *stats, hipStream_t s)
    {
        hipError_t e = hipMemsetAsync(stats, 0, 24, s);
        if (e != hipSuccess)
            return e;
        hipLaunchKernelGGL(plain_stats_kernel, dim3(1), dim3(kBlock), 0, s, m, coeff_count, stats);
        return hipGetLastError();
    }
    hipError_t k_negacyclic_mul_mono(const ModDesc *mods, const uint64_t *in, uint64_t *out, const uint64_t *scalars, size_t e,
                                     unsigned n_log, unsigned K, size_t items, hipStream_t s)
    {
        size_t w = (items * K) << n_log;
        if (!w)
            return hipSuccess;
        hipLaunchKernelGGL(negacyclic_mul_mono_kernel, dim3(grid_for(w)), dim3(kBlock), 0, s, mods, in, out, scalars, e, n_log, K, w);
        return hipGetLastError();
    }
    hipError_t k_bfv_addsub_plain(const ModDesc *mods, ModDesc t, const uint64_t *m, size_t coeff_count, uint64_t q_mod_t,
                                  uint64_t threshold, const uint64_t *delta_mod_q, uint64_t *c0, int op, unsigned n_log, unsigned K,
                                  size_t items, hipStream_t s)
    {
        size_t w = (items * K) << n_log;
        if (!w || !coeff_count)
            return hipSuccess;
        hipLaunchKernelGGL(bfv_addsub_plain_kernel, dim3(grid_for(w)), dim3(kBlock), 0, s, mods, t, m, coeff_count, q_mod_t, threshold,
                           delta_mod_q, c0, op, n_log, K, w);
        return hipGetLastError();
    }
    hipError_t k_mul_scalar(
        const ModDesc *mods, const uint64_t *a, uint64_t *r, uint64_t scalar, PlaneGeom g, unsigned planes, hipStream_t s)
    {
        size_t w = g.words() * planes;
        if (!w)
            return hipSuccess;
        hipLaunchKernelGGL(mul_scalar_kernel, dim3(grid_for(w)), dim3(kBlock), 0, s, mods, a, r, scalar, g.n_log, g.K, w);
        return hipGetLastError();
    }
    hipError_t k_bgv_delta(
        const ModDesc *mods, ModDesc t, uint64_t inv_q_last_mod_t, const uint64_t *q_last_mod_q, const uint64_t *c_last,
        size_t c_stride, uint64_t *delta, unsigned n_log, unsigned ncomp, size_t items, hipStream_t s)
    {
        size_t w = (items * ncomp) << n_log;
        if (!w)
            return hipSuccess;
        hipLaunchKernelGGL(
            bgv_delta_kernel, dim3(grid_for(w)), dim3(kBlock), 0, s, mods, t, inv_q_last_mod_t, q_last_mod_q, c_last, c_stride, delta,
            n_log, ncomp, w);
        return hipGetLastError();
    }
    hipError_t k_bfv_modswitch(
        const ModDesc *mods, const LevelDev &lv, const uint64_t *c, uint64_t *out, unsigned n_log, size_t items, hipStream_t s)
    {
        size_t w = (items * (lv.K - 1)) << n_log;
        if (!w)
            return hipSuccess;
        hipLaunchKernelGGL(
            bfv_modswitch_kernel, dim3(grid_for(w)), dim3(kBlock), 0, s, mods, lv.inv_q_last_mod_q, lv.half_mod_q,
            lv.q_last, lv.half_q_last, c, out, n_log, lv.K, w);
        return hipGetLastError();
    }
    hipError_t k_drop_last(const uint64_t *c, uint64_t *out, unsigned n_log, unsigned K, size_t items, hipStream_t s)
    {
        size_t w = (items * (K - 1)) << n_log;
        if (!w)
            return hipSuccess;
        hipLaunchKernelGGL(drop_last_kernel, dim3(grid_for(w)), dim3(kBlock), 0, s, c, out, n_log, K, w);
        return hipGetLastError();
    }
    hipError_t k_keyswitch_mac(
        const ModDesc *mods, const uint64_t *u, const uint64_t *key, uint64_t *acc, unsigned n_log, unsigned K, unsigned L,
        unsigned batch, unsigned j0, unsigned j1, unsigned key_digit0, hipStream_t s)
    {
        size_t N = size_t(1) << n_log;
        unsigned gx = (unsigned)((N + kBlock - 1) / kBlock);
        dim3 grid(gx, K + 1, (batch + KS_BI - 1) / KS_BI);
        hipLaunchKernelGGL(keyswitch_mac_kernel, grid, dim3(kBlock), 0, s, mods, u, key, acc, n_log, K, L, batch, j0, j1, key_digit0);
        return hipGetLastError();
    }
    hipError_t k_keyswitch_reduce(
        const ModDesc *mods, uint64_t *acc, unsigned n_log, unsigned K, unsigned L, unsigned batch, hipStream_t s, unsigned local_parts,
        const uint64_t *c0, const uint64_t *c1, const ShoupOp *pm, uint64_t *out)
    {
        size_t w = ((size_t)batch * 2 * (K + 1)) << n_log;
        if (!w)
            return hipSuccess;
        if (c0 && !pm)
            return hipErrorInvalidValue;
        hipLaunchKernelGGL(
            keyswitch_reduce_kernel, dim3(grid_for(w)), dim3(kBlock), 0, s, mods, acc, n_log, K, L, w, local_parts, c0, c1, pm, out ? out : acc);
        return hipGetLastError();
    }
    hipError_t k_ks_last_coeff(
        const ModDesc *mods, unsigned prime, const ShoupOp *pinv, const uint64_t *fix, uint64_t half_last, uint64_t *acc, unsigned n_log,
        unsigned K, size_t nouter, hipStream_t s)
    {
        const size_t words = nouter << n_log;
        if (!words)
            return hipSuccess;
        hipLaunchKernelGGL(ks_last_coeff_kernel, dim3(grid_for(words)), dim3(kBlock), 0, s, mods, prime, pinv, fix, half_last, acc, n_log, K, words);
        return hipGetLastError();
    }
    hipError_t k_ks_pack_targets(
        const uint64_t *acc, uint64_t *send, uint64_t *sp, unsigned n_log, unsigned K, unsigned G, unsigned m, unsigned batch, hipStream_t s)
    {
        const size_t send_words = ((size_t)G * m * batch * 2) << n_log, sp_words = ((size_t)batch * 2) << n_log;
        hipLaunchKernelGGL(
            ks_pack_targets_kernel, dim3(grid_for(send_words + sp_words)), dim3(kBlock), 0, s, acc, send, sp, n_log, K, G, m, batch,
            send_words, sp_words);
        return hipGetLastError();
    }
    hipError_t k_ks_unpack_owned(
        const ModDesc *mods, const uint64_t *recv, const uint64_t *sp, uint64_t *acc3, unsigned n_log, unsigned L, unsigned first,
        unsigned count, unsigned batch, hipStream_t s)
    {
        const size_t own_words = ((size_t)count * batch * 2) << n_log, sp_words = ((size_t)batch * 2) << n_log;
        hipLaunchKernelGGL(
            ks_unpack_owned_kernel, dim3(grid_for(own_words + sp_words)), dim3(kBlock), 0, s, mods, recv, sp, acc3, n_log, L, first,
            count, batch, own_words, sp_words);
        return hipGetLastError();
    }
    hipError_t k_ks_pack_owned(const uint64_t *inc, uint64_t *own, unsigned n_log, unsigned count, unsigned m, unsigned batch, hipStream_t s)
    {
        const size_t words = ((size_t)m * batch * 2) << n_log;
        if (!words)
            return hipSuccess;
        hipLaunchKernelGGL(ks_pack_owned_kernel, dim3(grid_for(words)), dim3(kBlock), 0, s, inc, own, n_log, count, m, batch, words);
        return hipGetLastError();
    }
    hipError_t k_ks_add_gathered(
        const ModDesc *mods, uint64_t *ct0, uint64_t *ct1, const uint64_t *all, unsigned n_log, unsigned K, unsigned G, unsigned m,
        unsigned batch, hipStream_t s)
    {
        const size_t words = ((size_t)batch * K) << n_log;
        if (!words)
            return hipSuccess;
        hipLaunchKernelGGL(ks_add_gathered_kernel, dim3(grid_for(words)), dim3(kBlock), 0, s, mods, ct0, ct1, all, n_log, K, G, m, batch, words);
        return hipGetLastError();
    }
    hipError_t k_keyswitch_tail_ckks(
        const ModDesc *mods, const ShoupOp *inv_p, uint64_t *ct0, uint64_t *ct1, const uint64_t *acc, const uint64_t *t,
        unsigned n_log, unsigned K, unsigned batch, hipStream_t s)
    {
        size_t w = ((size_t)batch * K) << n_log;
        hipLaunchKernelGGL(
            keyswitch_tail_ckks_kernel, dim3(grid_for(w)), dim3(kBlock), 0, s, mods, inv_p, ct0, ct1, acc, t, n_log, K, w);
        return hipGetLastError();
    }
    hipError_t k_keyswitch_tail_bfv(
        const ModDesc *mods, const ShoupOp *inv_p, const uint64_t *round_fix, uint64_t half_p, uint64_t p, uint64_t *ct0,
        uint64_t *ct1, const uint64_t *acc, unsigned n_log, unsigned K, unsigned batch, hipStream_t s)
    {
        size_t w = ((size_t)batch * K) << n_log;
        hipLaunchKernelGGL(
            keyswitch_tail_bfv_kernel, dim3(grid_for(w)), dim3(kBlock), 0, s, mods, inv_p, round_fix, half_p, p, ct0, ct1,
            acc, n_log, K, w);
        return hipGetLastError();
    }
    hipError_t k_keyswitch_tail_modswitch_bfv(
        const ModDesc *mods, const LevelDev &klv, uint64_t half_p, uint64_t p, const LevelDev &lv, const uint64_t *ct0, const uint64_t *ct1,
        const uint64_t *acc, uint64_t *out, unsigned n_log, unsigned K, unsigned batch, hipStream_t s)
    {
        const size_t threads = ((size_t)batch * 2) << n_log;
        if (!threads || K < 2)
            return hipErrorInvalidValue;
        // one workgroup per 256 coefficients, no grid-stride loop (round 5: the shape this memory system streams fastest,
        // profiles/r05_tensor_wide.txt; this pass moves 21 GB per configs[3] step)
        size_t blocks = (threads + kBlock - 1) / kBlock;
        if (blocks > (size_t)0x7fffffff)
            blocks = 0x7fffffff;
        hipLaunchKernelGGL(
            keyswitch_tail_modswitch_bfv_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, s, mods, klv.inv_q_last_mod_q, klv.round_fix, half_p,
            p, lv.inv_q_last_mod_q, lv.half_mod_q, lv.q_last, lv.half_q_last, ct0, ct1, acc, out, n_log, K, batch, threads);
        return hipGetLastError();
    }
    hipError_t k_any_nonzero(const uint64_t *data, size_t words, unsigned *flag, hipStream_t s)
    {
        if (!words)
            return hipSuccess;
        hipLaunchKernelGGL(any_nonzero_kernel, dim3(grid_for(words)), dim3(kBlock), 0, s, data, words, flag);
        return hipGetLastError();
    }
} // namespace sealhip
