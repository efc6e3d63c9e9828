// See decrypt_kernels.h.  Every result is a canonical residue of an exactly specified integer, except the quotient estimate of
// exact_convert_array, which the reference computes in double precision (sum of x_i / q_i, + 0.5, truncate) and which is
// reproduced operation for operation (IEEE division and addition in the same order; the build uses -ffp-contract=off).
#include "decrypt_kernels.h"

namespace sealhip
{
    namespace
    {
        constexpr unsigned kBlock = 256;
        inline unsigned grid_for(size_t work)
        {
            size_t b = (work + kBlock - 1) / kBlock;
            if (b > 2048)
                b = 2048;
            if (b == 0)
                b = 1;
            return (unsigned)b;
        }
        __device__ __forceinline__ void mac128(uint64_t &lo, uint64_t &hi, uint64_t a, uint64_t b)
        {
            uint64_t pl, ph;
            mul_wide(a, b, pl, ph);
            lo += pl;
            hi += ph + (lo < pl);
        }

        __global__ void __launch_bounds__(kBlock) decrypt_dot_kernel(
            const ModDesc *mods, const uint64_t *plane0, const uint64_t *planes1, size_t plane_words, unsigned size, SkPowers sk,
            uint64_t *out, unsigned n_log, unsigned K)
        {
            const size_t nmask = (size_t(1) << n_log) - 1;
            for (size_t i = blockIdx.x * (size_t)kBlock + threadIdx.x; i < plane_words; i += (size_t)gridDim.x * kBlock)
            {
                const unsigned r = (unsigned)((i >> n_log) % K);
                const ModDesc md = mods[r];
                const size_t kj = ((size_t)r << n_log) + (i & nmask); // position inside a key-level polynomial [L][N]
                uint64_t lo = plane0 ? plane0[i] : 0, hi = 0;
                for (unsigned p = 1; p < size; p++)
                    mac128(lo, hi, planes1[(size_t)(p - 1) * plane_words + i], sk.p[p - 1][kj]);
                out[i] = barrett128(lo, hi, md); // at most 5 products below 2^122 plus one word: no overflow
            }
        }

        __global__ void __launch_bounds__(kBlock) add_inplace_kernel(
            const ModDesc *mods, uint64_t *out, const uint64_t *a, size_t words, unsigned n_log, unsigned K)
        {
            for (size_t i = blockIdx.x * (size_t)kBlock + threadIdx.x; i < words; i += (size_t)gridDim.x * kBlock)
            {
                const unsigned r = (unsigned)((i >> n_log) % K);
                out[i] = add_mod(out[i], a[i], mods[r].q);
            }
        }

        __global__ void __launch_bounds__(kBlock) neg_add_noise_kernel(
            const ModDesc *mods, uint64_t *c0, const uint64_t *e, uint64_t m, size_t words, unsigned n_log, unsigned K, bool negate)
        {
            for (size_t i = blockIdx.x * (size_t)kBlock + threadIdx.x; i < words; i += (size_t)gridDim.x * kBlock)
            {
                const ModDesc md = mods[(unsigned)((i >> n_log) % K)];
                uint64_t noise = e[i];
                if (m != 1)
                    noise = mul_mod(noise, barrett64(m, md), md);
                const uint64_t sum = add_mod(c0[i], noise, md.q);
                c0[i] = negate ? neg_mod(sum, md.q) : sum;
            }
        }

        __global__ void __launch_bounds__(kBlock) expand_small_kernel(
            const ModDesc *mods, const int8_t *small, uint64_t *out, size_t words, unsigned n_log, unsigned K)
        {
            const size_t nmask = (size_t(1) << n_log) - 1;
            for (size_t i = blockIdx.x * (size_t)kBlock + threadIdx.x; i < words; i += (size_t)gridDim.x * kBlock)
            {
                const size_t row = i >> n_log; // p * K + r
                const int v = small[((row / K) << n_log) + (i & nmask)];
                out[i] = v < 0 ? mods[(unsigned)(row % K)].q - (uint64_t)(-v) : (uint64_t)v;
            }
        }

        __global__ void __launch_bounds__(kBlock) slot_scatter_kernel(
            const uint32_t *map, const uint64_t *in, uint64_t *out, unsigned n_log, size_t words, uint64_t signed_mod)
        {
            const size_t nmask = (size_t(1) << n_log) - 1;
            for (size_t i = blockIdx.x * (size_t)kBlock + threadIdx.x; i < words; i += (size_t)gridDim.x * kBlock)
            {
                uint64_t v = in[i];
                if (signed_mod && (int64_t)v < 0)
                    v += signed_mod;
                out[(i & ~nmask) + map[i & nmask]] = v;
            }
        }
        __global__ void __launch_bounds__(kBlock) slot_gather_kernel(
            const uint32_t *map, const uint64_t *in, uint64_t *out, unsigned n_log, size_t words, uint64_t signed_mod)
        {
            const size_t nmask = (size_t(1) << n_log) - 1;
            for (size_t i = blockIdx.x * (size_t)kBlock + threadIdx.x; i < words; i += (size_t)gridDim.x * kBlock)
            {
                uint64_t v = in[(i & ~nmask) + map[i & nmask]];
                if (signed_mod && v > (signed_mod >> 1))
                    v -= signed_mod;
                out[i] = v;
            }
        }

        // rns.cpp:1133-1191
        __global__ void __launch_bounds__(kBlock) decrypt_scale_and_round_kernel(
            const ModDesc *mods, LevelDev lvl, ModDesc t, const uint64_t *phase, uint64_t *out, unsigned n_log, size_t coeffs)
        {
            const unsigned K = lvl.K;
            const size_t nmask = (size_t(1) << n_log) - 1;
            const ModDesc g = mods[lvl.gamma_prime];
            for (size_t i = blockIdx.x * (size_t)kBlock + threadIdx.x; i < coeffs; i += (size_t)gridDim.x * kBlock)
            {
                const size_t b = i >> n_log, j = i & nmask;
                const uint64_t *in = phase + (((size_t)b * K) << n_log) + j;
                uint64_t tlo = 0, thi = 0, glo = 0, ghi = 0;
                for (unsigned r = 0; r < K; r++)
                {
                    const uint64_t q = mods[r].q;
                    // |gamma t|_{q_r} x, then the fast base conversion's y_r = . (Q/q_r)^-1 mod q_r
                    const ShoupOp ptg = lvl.dec_prod_t_gamma_mod_q[r], ipq = lvl.dec_inv_punct_q[r];
                    uint64_t y = mul_shoup(in[(size_t)r << n_log], ptg.w, ptg.wq, q);
                    y = mul_shoup(y, ipq.w, ipq.wq, q);
                    mac128(tlo, thi, y, lvl.dec_q_to_t[r]);
                    mac128(glo, ghi, y, lvl.dec_q_to_gamma[r]);
                }
                // times -Q^-1 modulo t and gamma
                const uint64_t vt = mul_mod(barrett128(tlo, thi, t), lvl.dec_neg_inv_q_mod_t, t);
                const uint64_t vg = mul_mod(barrett128(glo, ghi, g), lvl.dec_neg_inv_q_mod_gamma, g);
                // remove the error term: centred gamma component, then gamma^-1 mod t
                const uint64_t add_result = add_mod(vt, barrett64(g.q - vg, t), t.q);
                const uint64_t sub_result = sub_mod(vt, barrett64(vg, t), t.q);
                const uint64_t v = vg > (g.q >> 1) ? add_result : sub_result;
                out[i] = mul_mod(v, lvl.dec_inv_gamma_mod_t, t);
            }
        }

        // exact_convert_array (rns.cpp:465-540) with the single output modulus t, then the correction-factor fix
        __global__ void __launch_bounds__(kBlock) decrypt_modt_kernel(
            const ModDesc *mods, LevelDev lvl, ModDesc t, uint64_t fix, const uint64_t *phase, uint64_t *out, unsigned n_log, size_t coeffs)
        {
            const unsigned K = lvl.K;
            const size_t nmask = (size_t(1) << n_log) - 1;
            for (size_t i = blockIdx.x * (size_t)kBlock + threadIdx.x; i < coeffs; i += (size_t)gridDim.x * kBlock)
            {
                const size_t b = i >> n_log, j = i & nmask;
                const uint64_t *in = phase + (((size_t)b * K) << n_log) + j;
                uint64_t lo = 0, hi = 0;
                double aggregated_v = 0.0;
                for (unsigned r = 0; r < K; r++)
                {
                    const uint64_t q = mods[r].q;
                    const ShoupOp ipq = lvl.dec_inv_punct_q[r];
                    const uint64_t y = mul_shoup(in[(size_t)r << n_log], ipq.w, ipq.wq, q);
                    aggregated_v += (double)y / (double)q;
                    mac128(lo, hi, y, lvl.dec_q_to_t[r]);
                }
                aggregated_v += 0.5;
                const uint64_t rounded_v = (uint64_t)aggregated_v;
                const uint64_t sum_mod_t = barrett128(lo, hi, t);
                const uint64_t v_q_mod_t = mul_mod(barrett64(rounded_v, t), lvl.q_mod_t, t);
                uint64_t v = sub_mod(sum_mod_t, v_q_mod_t, t.q);
                if (fix != 1)
                    v = mul_mod(v, fix, t);
                out[i] = v;
            }
        }
    } // namespace

    hipError_t k_decrypt_dot(const ModDesc *mods, const uint64_t *plane0, const uint64_t *planes1, size_t plane_words, unsigned size,
                             SkPowers sk, uint64_t *out, unsigned n_log, unsigned K, hipStream_t s)
    {
        if (!plane_words)
            return hipSuccess;
        hipLaunchKernelGGL(decrypt_dot_kernel, dim3(grid_for(plane_words)), dim3(kBlock), 0, s, mods, plane0, planes1, plane_words, size, sk,
                           out, n_log, K);
        return hipGetLastError();
    }
    hipError_t k_add_inplace(const ModDesc *mods, uint64_t *out, const uint64_t *a, size_t words, unsigned n_log, unsigned K, hipStream_t s)
    {
        if (!words)
            return hipSuccess;
        hipLaunchKernelGGL(add_inplace_kernel, dim3(grid_for(words)), dim3(kBlock), 0, s, mods, out, a, words, n_log, K);
        return hipGetLastError();
    }
    hipError_t k_neg_add_noise(const ModDesc *mods, uint64_t *c0, const uint64_t *e, uint64_t m, size_t words, unsigned n_log, unsigned K,
                               hipStream_t s, bool negate)
    {
        if (!words)
            return hipSuccess;
        hipLaunchKernelGGL(neg_add_noise_kernel, dim3(grid_for(words)), dim3(kBlock), 0, s, mods, c0, e, m, words, n_log, K, negate);
        return hipGetLastError();
    }
    hipError_t k_expand_small(const ModDesc *mods, const int8_t *small, uint64_t *out, unsigned n_log, unsigned K, unsigned polys, hipStream_t s)
    {
        const size_t words = ((size_t)polys * K) << n_log;
        if (!words)
            return hipSuccess;
        hipLaunchKernelGGL(expand_small_kernel, dim3(grid_for(words)), dim3(kBlock), 0, s, mods, small, out, words, n_log, K);
        return hipGetLastError();
    }
    hipError_t k_slot_scatter(const uint32_t *map, const uint64_t *in, uint64_t *out, unsigned n_log, unsigned batch, uint64_t signed_mod,
                              hipStream_t s)
    {
        const size_t words = (size_t)batch << n_log;
        if (!words)
            return hipSuccess;
        hipLaunchKernelGGL(slot_scatter_kernel, dim3(grid_for(words)), dim3(kBlock), 0, s, map, in, out, n_log, words, signed_mod);
        return hipGetLastError();
    }
    hipError_t k_slot_gather(const uint32_t *map, const uint64_t *in, uint64_t *out, unsigned n_log, unsigned batch, uint64_t signed_mod,
                             hipStream_t s)
    {
        const size_t words = (size_t)batch << n_log;
        if (!words)
            return hipSuccess;
        hipLaunchKernelGGL(slot_gather_kernel, dim3(grid_for(words)), dim3(kBlock), 0, s, map, in, out, n_log, words, signed_mod);
        return hipGetLastError();
    }
    hipError_t k_decrypt_scale_and_round(const ModDesc *mods, const LevelDev &lvl, ModDesc t, const uint64_t *phase, uint64_t *out,
                                         unsigned n_log, unsigned batch, hipStream_t s)
    {
        const size_t coeffs = (size_t)batch << n_log;
        hipLaunchKernelGGL(decrypt_scale_and_round_kernel, dim3(grid_for(coeffs)), dim3(kBlock), 0, s, mods, lvl, t, phase, out, n_log, coeffs);
        return hipGetLastError();
    }
    hipError_t k_decrypt_modt(const ModDesc *mods, const LevelDev &lvl, ModDesc t, uint64_t fix, const uint64_t *phase, uint64_t *out,
                              unsigned n_log, unsigned batch, hipStream_t s)
    {
        const size_t coeffs = (size_t)batch << n_log;
        hipLaunchKernelGGL(decrypt_modt_kernel, dim3(grid_for(coeffs)), dim3(kBlock), 0, s, mods, lvl, t, fix, phase, out, n_log, coeffs);
        return hipGetLastError();
    }
} // namespace sealhip
