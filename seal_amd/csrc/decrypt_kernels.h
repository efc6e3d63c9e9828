// Kernels of the decryption path: the phase c_0 + c_1 s + ... + c_{k-1} s^{k-1} (Decryptor::
// dot_product_ct_sk_array, native/src/seal/decryptor.cpp) and the two RNS read-outs of the plaintext
// (RNSTool::decrypt_scale_and_round, rns.cpp:1133-1191: BFV; RNSTool::decrypt_modt = BaseConverter::exact_convert_array,
// rns.cpp:465-540: BGV).  Element-wise, HBM-streaming, one pass each; batch = independent ciphertexts.
#pragma once
#include "context.h"

namespace sealhip
{
    struct SkPowers
    {
        const uint64_t *p[5]; // s^1 .. s^5 in NTT form at the key level, [L][N] each (SEAL_CIPHERTEXT_SIZE_MAX = 6)
    };
    // out[b][r][j] = (with_c0 ? ct_0 : 0) + sum_{p=1..size-1} ct_p[b][r][j] * s^p[r][j]  mod q_r;  ct planes are [batch][K][N]
    // (src may hold only the planes 1.. when with_c0 == 0: pass plane0 = nullptr)
    hipError_t k_decrypt_dot(const ModDesc *mods, const uint64_t *plane0, const uint64_t *planes1, size_t plane_words, unsigned size,
                             SkPowers sk, uint64_t *out, unsigned n_log, unsigned K, hipStream_t s);
    // out = out + a mod q_r over [batch][K][N]
    hipError_t k_add_inplace(const ModDesc *mods, uint64_t *out, const uint64_t *a, size_t words, unsigned n_log, unsigned K, hipStream_t s);
    // symmetric encryption tail (encrypt_zero_symmetric, util/rlwe.cpp:357-381): c0 <- -(c0 + e * m) mod q_r over [K][N] words,
    // m = t (BGV: the noise is p*e) or 1
    hipError_t k_neg_add_noise(const ModDesc *mods, uint64_t *c0, const uint64_t *e, uint64_t m, size_t words, unsigned n_log, unsigned K,
                               hipStream_t s, bool negate = true); // negate == false: c0 <- c0 + e * m (public-key encryption)
    // the Encryptor's small polynomials (u, e_0, e_1: N signed bytes each, sampled on the host) replicated into the RNS components:
    // out[p][r][i] = small[p][i] < 0 ? q_r + small[p][i] : small[p][i] over `polys` polynomials (util/rlwe.cpp:24-43, 120-150)
    hipError_t k_expand_small(const ModDesc *mods, const int8_t *small, uint64_t *out, unsigned n_log, unsigned K, unsigned polys, hipStream_t s);
    // BatchEncoder index map (batchencoder.cpp:97-123): scatter out[b][map[i]] = in[b][i] (encode), gather out[b][i] = in[b][map[i]]
    // (decode) over `batch` vectors of N words; signed_mod != 0 converts between the balanced signed representation and [0, t):
    // encode: negative int64 v -> t + v; decode: value > t/2 -> value - t (as int64)
    hipError_t k_slot_scatter(const uint32_t *map, const uint64_t *in, uint64_t *out, unsigned n_log, unsigned batch, uint64_t signed_mod,
                              hipStream_t s);
    hipError_t k_slot_gather(const uint32_t *map, const uint64_t *in, uint64_t *out, unsigned n_log, unsigned batch, uint64_t signed_mod,
                             hipStream_t s);
    // BFV: phase [batch][K][N] (coefficient form) -> plaintext coefficients mod t, [batch][N]
    hipError_t k_decrypt_scale_and_round(const ModDesc *mods, const LevelDev &lvl, ModDesc t, const uint64_t *phase, uint64_t *out,
                                         unsigned n_log, unsigned batch, hipStream_t s);
    // BGV: phase [batch][K][N] (coefficient form) -> (phase mod t) * fix mod t, [batch][N]; fix = correction_factor^-1 mod t
    hipError_t k_decrypt_modt(const ModDesc *mods, const LevelDev &lvl, ModDesc t, uint64_t fix, const uint64_t *phase, uint64_t *out,
                              unsigned n_log, unsigned batch, hipStream_t s);
} // namespace sealhip
