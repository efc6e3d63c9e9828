// See ckks_kernels.h.
#include "ckks_kernels.h"
#include <cmath>

namespace sealhip
{
    namespace
    {
        constexpr unsigned kBlock = 256;
        inline unsigned grid_for(size_t work)
        {
            size_t b = (work + kBlock - 1) / kBlock;
            if (b > 4096)
                b = 4096;
            if (b == 0)
                b = 1;
            return (unsigned)b;
        }
        // std::complex<double> operator* as GCC evaluates it without -ffast-math for finite operands
        __device__ __forceinline__ double2 cmul(double2 a, double2 b)
        {
            return double2{ a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x };
        }
        __device__ __forceinline__ double2 cadd(double2 a, double2 b)
        {
            return double2{ a.x + b.x, a.y + b.y };
        }
        __device__ __forceinline__ double2 csubc(double2 a, double2 b)
        {
            return double2{ a.x - b.x, a.y - b.y };
        }

        __global__ void __launch_bounds__(kBlock) fft_gs_stage_kernel(
            double2 *values, const double2 *roots, unsigned n_log, unsigned gap_log, size_t butterflies, const double *last_scalar)
        {
            const size_t half = size_t(1) << (n_log - 1), gap = size_t(1) << gap_log;
            const size_t m = half >> gap_log; // groups of this stage
            const size_t root_start = (size_t(1) << n_log) - 2 * m + 1;
            for (size_t t = blockIdx.x * (size_t)kBlock + threadIdx.x; t < butterflies; t += (size_t)gridDim.x * kBlock)
            {
                const size_t vec = t >> (n_log - 1), b = t & (half - 1);
                const size_t i = b >> gap_log, j = b & (gap - 1);
                double2 *x = values + (vec << n_log) + (i << (gap_log + 1)) + j, *y = x + gap;
                const double2 u = *x, v = *y, r = roots[root_start + i];
                if (last_scalar)
                {
                    const double sc = *last_scalar;
                    const double2 scaled_r{ r.x * sc, r.y * sc };
                    const double2 sum = cadd(u, v);
                    *x = double2{ sum.x * sc, sum.y * sc };
                    *y = cmul(csubc(u, v), scaled_r);
                }
                else
                {
                    *x = cadd(u, v);
                    *y = cmul(csubc(u, v), r);
                }
            }
        }
        __global__ void __launch_bounds__(kBlock) fft_ct_stage_kernel(
            double2 *values, const double2 *roots, unsigned n_log, unsigned gap_log, size_t butterflies)
        {
            const size_t half = size_t(1) << (n_log - 1), gap = size_t(1) << gap_log;
            const size_t m = half >> gap_log;
            for (size_t t = blockIdx.x * (size_t)kBlock + threadIdx.x; t < butterflies; t += (size_t)gridDim.x * kBlock)
            {
                const size_t vec = t >> (n_log - 1), b = t & (half - 1);
                const size_t i = b >> gap_log, j = b & (gap - 1);
                double2 *x = values + (vec << n_log) + (i << (gap_log + 1)) + j, *y = x + gap;
                const double2 u = *x, v = cmul(*y, roots[m + i]);
                *x = cadd(u, v);
                *y = csubc(u, v);
            }
        }
        __global__ void __launch_bounds__(kBlock) max_abs_real_kernel(const double2 *values, size_t count, unsigned long long *out)
        {
            unsigned long long best = 0;
            for (size_t i = blockIdx.x * (size_t)kBlock + threadIdx.x; i < count; i += (size_t)gridDim.x * kBlock)
            {
                const unsigned long long bits = (unsigned long long)__double_as_longlong(::fabs(values[i].x));
                best = bits > best ? bits : best;
            }
            atomicMax(out, best);
        }
        __global__ void __launch_bounds__(kBlock) ckks_decompose_kernel(
            const ModDesc *mods, const double2 *values, uint64_t *out, unsigned n_log, unsigned K, size_t count, int mode)
        {
            const size_t nmask = (size_t(1) << n_log) - 1;
            const double two_pow_64 = 18446744073709551616.0;
            for (size_t t = blockIdx.x * (size_t)kBlock + threadIdx.x; t < count; t += (size_t)gridDim.x * kBlock)
            {
                const size_t vec = t >> n_log, i = t & nmask;
                double coeffd = ::round(values[t].x);
                const bool is_negative = std::signbit(coeffd);
                coeffd = ::fabs(coeffd);
                uint64_t lo, hi = 0;
                if (mode == 64)
                    lo = (uint64_t)coeffd;
                else
                {
                    lo = (uint64_t)::fmod(coeffd, two_pow_64);
                    hi = (uint64_t)(coeffd / two_pow_64);
                }
                uint64_t *o = out + ((vec * K) << n_log) + i;
                if (mode != 64 && mode != 128)
                {
                    // the "slow case" of encode_internal (ckks.h:624-672): the rounded double is cut into 64-bit words by
                    // repeated fmod / division by 2^64 (both exact: the divisor is a power of two), at most K words since the
                    // coefficient is below the level's modulus, and the K-word integer is reduced modulo every prime
                    // (RNSBase::decompose -> modulo_uint).  Horner over the words from the top: r <- (r 2^64 + word) mod q.
                    uint64_t words[kMaxComps];
                    unsigned nw = 0;
                    double c = coeffd;
                    while (c >= 1 && nw < K)
                    {
                        words[nw++] = (uint64_t)::fmod(c, two_pow_64);
                        c /= two_pow_64;
                    }
                    for (unsigned j = 0; j < K; j++)
                    {
                        const ModDesc md = mods[j];
                        uint64_t r = 0;
                        for (unsigned w = nw; w-- > 0;)
                            r = barrett128(words[w], r, md);
                        o[(size_t)j << n_log] = is_negative ? neg_mod(r, md.q) : r;
                    }
                    continue;
                }
                for (unsigned j = 0; j < K; j++)
                {
                    const ModDesc md = mods[j];
                    const uint64_t r = mode == 64 ? barrett64(lo, md) : barrett128(lo, hi, md);
                    o[(size_t)j << n_log] = is_negative ? neg_mod(r, md.q) : r;
                }
            }
        }
        // x = sum_j [x_j m (Q/q_j)^-1 mod q_j] (Q/q_j) mod Q as K little-endian words (RNSBase::compose_array, rns.cpp:300-360);
        // m = an optional scalar multiplied into every residue first (1 = none)
        __device__ __forceinline__ void crt_compose(
            uint64_t (&acc)[kMaxComps], const uint64_t *in, uint64_t m, const ModDesc *mods, const uint64_t *punct, const ShoupOp *inv_punct,
            const uint64_t *q_words, unsigned n_log, unsigned K)
        {
            for (unsigned w = 0; w < K; w++)
                acc[w] = 0;
            for (unsigned j = 0; j < K; j++)
            {
                const ShoupOp ip = inv_punct[j];
                uint64_t xj = in[(size_t)j << n_log];
                if (m != 1)
                    xj = mul_mod(xj, barrett64(m, mods[j]), mods[j]);
                const uint64_t y = mul_shoup(xj, ip.w, ip.wq, mods[j].q);

                // acc += y * punct_j  (the product is below Q: K words), then one conditional subtraction of Q
                uint64_t carry = 0;
                for (unsigned w = 0; w < K; w++)
                {
                    uint64_t lo, hi;
                    mul_wide(y, punct[(size_t)j * K + w], lo, hi);
                    const uint64_t s1 = acc[w] + lo;
                    const uint64_t c1 = s1 < lo;
                    const uint64_t s2 = s1 + carry;
                    const uint64_t c2 = s2 < carry;
                    acc[w] = s2;
                    carry = hi + c1 + c2; // hi <= 2^64 - 2, so this does not wrap
                }
                // the sum of two values below Q is below 2Q < 2^(64 K + 1): `carry` is its top bit
                bool ge = carry != 0;
                if (!ge)
                {
                    ge = true;
                    for (int w = (int)K - 1; w >= 0; w--)
                        if (acc[w] != q_words[w])
                        {
                            ge = acc[w] > q_words[w];
                            break;
                        }
                }
                if (ge)
                {
                    uint64_t borrow = 0;
                    for (unsigned w = 0; w < K; w++)
                    {
                        const uint64_t d = acc[w] - q_words[w];
                        const uint64_t b1 = acc[w] < q_words[w];
                        const uint64_t d2 = d - borrow;
                        const uint64_t b2 = d < borrow;
                        acc[w] = d2;
                        borrow = b1 | b2;
                    }
                }
            }
        }

        __global__ void __launch_bounds__(kBlock) ckks_compose_scale_kernel(
            const ModDesc *mods, const uint64_t *residues, const uint64_t *punct, const ShoupOp *inv_punct, const uint64_t *q_words,
            const uint64_t *half_words, double inv_scale, double2 *out, unsigned n_log, unsigned K, size_t count)
        {
            const size_t nmask = (size_t(1) << n_log) - 1;
            const double two_pow_64 = 18446744073709551616.0;
            for (size_t t = blockIdx.x * (size_t)kBlock + threadIdx.x; t < count; t += (size_t)gridDim.x * kBlock)
            {
                const size_t vec = t >> n_log, i = t & nmask;
                const uint64_t *in = residues + ((vec * K) << n_log) + i;
                uint64_t acc[kMaxComps];
                crt_compose(acc, in, 1, mods, punct, inv_punct, q_words, n_log, K);
                // ckks.h:746-775: the centred value times inv_scale, word by word
                bool upper = true; // acc >= upper_half_threshold
                for (int w = (int)K - 1; w >= 0; w--)
                    if (acc[w] != half_words[w])
                    {
                        upper = acc[w] > half_words[w];
                        break;
                    }
                double res = 0.0;
                double scaled_two_pow_64 = inv_scale;
                for (unsigned w = 0; w < K; w++, scaled_two_pow_64 *= two_pow_64)
                {
                    if (upper)
                    {
                        if (acc[w] > q_words[w])
                        {
                            const uint64_t diff = acc[w] - q_words[w];
                            res += diff ? (double)diff * scaled_two_pow_64 : 0.0;
                        }
                        else
                        {
                            const uint64_t diff = q_words[w] - acc[w];
                            res -= diff ? (double)diff * scaled_two_pow_64 : 0.0;
                        }
                    }
                    else
                    {
                        const uint64_t c = acc[w];
                        res += c ? (double)c * scaled_two_pow_64 : 0.0;
                    }
                }
                out[t] = double2{ res, 0.0 };
            }
        }
        // significant bits of the centred CRT value of every coefficient, maximum per vector (poly_infty_norm_coeffmod)
        __global__ void __launch_bounds__(kBlock) crt_norm_bits_kernel(
            const ModDesc *mods, const uint64_t *residues, const uint64_t *punct, const ShoupOp *inv_punct, const uint64_t *q_words,
            const uint64_t *half_words, uint64_t m, unsigned *out_bits, unsigned n_log, unsigned K, size_t count)
        {
            const size_t nmask = (size_t(1) << n_log) - 1;
            for (size_t t = blockIdx.x * (size_t)kBlock + threadIdx.x; t < count; t += (size_t)gridDim.x * kBlock)
            {
                const size_t vec = t >> n_log, i = t & nmask;
                uint64_t acc[kMaxComps];
                crt_compose(acc, residues + ((vec * K) << n_log) + i, m, mods, punct, inv_punct, q_words, n_log, K);
                bool upper = true; // acc >= (Q + 1) / 2: the representative is Q - acc
                for (int w = (int)K - 1; w >= 0; w--)
                    if (acc[w] != half_words[w])
                    {
                        upper = acc[w] > half_words[w];
                        break;
                    }
                if (upper)
                {
                    uint64_t borrow = 0;
                    for (unsigned w = 0; w < K; w++)
                    {
                        const uint64_t d = q_words[w] - acc[w];
                        const uint64_t b1 = q_words[w] < acc[w];
                        const uint64_t d2 = d - borrow;
                        const uint64_t b2 = d < borrow;
                        acc[w] = d2;
                        borrow = b1 | b2;
                    }
                }
                unsigned bits = 0;
                for (int w = (int)K - 1; w >= 0; w--)
                    if (acc[w])
                    {
                        bits = (unsigned)w * 64 + (64 - __builtin_clzll(acc[w]));
                        break;
                    }
                atomicMax(out_bits + vec, bits);
            }
        }
        __global__ void __launch_bounds__(kBlock) ckks_place_kernel(const uint32_t *map, const double2 *in, double2 *out, unsigned n_log, unsigned count)
        {
            const unsigned slots = 1u << (n_log - 1);
            for (unsigned i = blockIdx.x * kBlock + threadIdx.x; i < count; i += gridDim.x * kBlock)
            {
                const double2 v = in[i];
                out[map[i]] = v;
                out[map[i + slots]] = double2{ v.x, -v.y };
            }
        }
        __global__ void __launch_bounds__(kBlock) ckks_gather_kernel(const uint32_t *map, const double2 *in, double2 *out, unsigned n_log)
        {
            const unsigned slots = 1u << (n_log - 1);
            for (unsigned i = blockIdx.x * kBlock + threadIdx.x; i < slots; i += gridDim.x * kBlock)
                out[i] = in[map[i]];
        }
    } // namespace

    hipError_t k_fft_gs_stage(double2 *values, const double2 *roots, unsigned n_log, unsigned gap_log, unsigned batch, const double *last_scalar,
                              hipStream_t s)
    {
        const size_t butterflies = (size_t)batch << (n_log - 1);
        hipLaunchKernelGGL(fft_gs_stage_kernel, dim3(grid_for(butterflies)), dim3(kBlock), 0, s, values, roots, n_log, gap_log, butterflies,
                           last_scalar);
        return hipGetLastError();
    }
    hipError_t k_fft_ct_stage(double2 *values, const double2 *roots, unsigned n_log, unsigned gap_log, unsigned batch, hipStream_t s)
    {
        const size_t butterflies = (size_t)batch << (n_log - 1);
        hipLaunchKernelGGL(fft_ct_stage_kernel, dim3(grid_for(butterflies)), dim3(kBlock), 0, s, values, roots, n_log, gap_log, butterflies);
        return hipGetLastError();
    }
    hipError_t k_max_abs_real(const double2 *values, size_t count, unsigned long long *out, hipStream_t s)
    {
        hipLaunchKernelGGL(max_abs_real_kernel, dim3(grid_for(count)), dim3(kBlock), 0, s, values, count, out);
        return hipGetLastError();
    }
    hipError_t k_ckks_decompose(const ModDesc *mods, const double2 *values, uint64_t *out, unsigned n_log, unsigned K, unsigned batch, int mode,
                                hipStream_t s)
    {
        const size_t count = (size_t)batch << n_log;
        hipLaunchKernelGGL(ckks_decompose_kernel, dim3(grid_for(count)), dim3(kBlock), 0, s, mods, values, out, n_log, K, count, mode);
        return hipGetLastError();
    }
    hipError_t k_ckks_compose_scale(const ModDesc *mods, const uint64_t *residues, const uint64_t *punct, const ShoupOp *inv_punct,
                                    const uint64_t *q_words, const uint64_t *half_words, double inv_scale, double2 *out, unsigned n_log, unsigned K,
                                    unsigned batch, hipStream_t s)
    {
        const size_t count = (size_t)batch << n_log;
        hipLaunchKernelGGL(ckks_compose_scale_kernel, dim3(grid_for(count)), dim3(kBlock), 0, s, mods, residues, punct, inv_punct, q_words,
                           half_words, inv_scale, out, n_log, K, count);
        return hipGetLastError();
    }
    hipError_t k_crt_norm_bits(const ModDesc *mods, const uint64_t *residues, const uint64_t *punct, const ShoupOp *inv_punct,
                               const uint64_t *q_words, const uint64_t *half_words, uint64_t m, unsigned *out_bits, unsigned n_log, unsigned K,
                               unsigned batch, hipStream_t s)
    {
        const size_t count = (size_t)batch << n_log;
        hipLaunchKernelGGL(crt_norm_bits_kernel, dim3(grid_for(count)), dim3(kBlock), 0, s, mods, residues, punct, inv_punct, q_words, half_words,
                           m, out_bits, n_log, K, count);
        return hipGetLastError();
    }
    hipError_t k_ckks_place(const uint32_t *map, const double2 *in, double2 *out, unsigned n_log, unsigned count, hipStream_t s)
    {
        if (!count)
            return hipSuccess;
        hipLaunchKernelGGL(ckks_place_kernel, dim3(grid_for(count)), dim3(kBlock), 0, s, map, in, out, n_log, count);
        return hipGetLastError();
    }
    hipError_t k_ckks_gather(const uint32_t *map, const double2 *in, double2 *out, unsigned n_log, hipStream_t s)
    {
        hipLaunchKernelGGL(ckks_gather_kernel, dim3(grid_for(size_t(1) << (n_log - 1))), dim3(kBlock), 0, s, map, in, out, n_log);
        return hipGetLastError();
    }
} // namespace sealhip
