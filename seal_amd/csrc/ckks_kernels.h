// Kernels of the CKKS encoder (seal::CKKSEncoder, native/src/seal/ckks.h:458-789): the complex FFT of
// util::DWTHandler (dwthandler.h:94-356) in double precision, the rounding / RNS decomposition of encode_internal and the CRT
// composition / scaling of decode_internal.  Floating point here is the reference's own: every operation of every butterfly is
// the same IEEE-754 double operation in the same order (complex product = (ac - bd, ad + bc), no contraction), so the results
// are the reference's bit for bit; the order in which independent butterflies run is irrelevant.
#pragma once
#include "context.h"

namespace sealhip
{
    // one Gentleman-Sande stage of transform_from_rev (dwthandler.h:202-356) over `batch` vectors of 2^n_log complex values:
    // butterflies (x, y) <- (x + y, (x - y) * r), group i of the stage uses roots[root_start + i].  last_scalar != nullptr marks
    // the final stage with the scalar folded in: x <- (x + y) * s, y <- (x - y) * (r * s).
    hipError_t k_fft_gs_stage(double2 *values, const double2 *roots, unsigned n_log, unsigned gap_log, unsigned batch, const double *last_scalar,
                              hipStream_t s);
    // one Cooley-Tukey stage of transform_to_rev (dwthandler.h:94-191): (x, y) <- (x + y r, x - y r), roots[m + i]
    hipError_t k_fft_ct_stage(double2 *values, const double2 *roots, unsigned n_log, unsigned gap_log, unsigned batch, hipStream_t s);
    // max |Re v| over the vectors, as the bit pattern of a non-negative double (a NaN compares largest); *out must be zeroed
    hipError_t k_max_abs_real(const double2 *values, size_t count, unsigned long long *out, hipStream_t s);
    // encode_internal's rounding and decomposition (ckks.h:559-672): mode 64: |coefficient| < 2^64; mode 128: < 2^128; any other
    // mode: the multi-precision branch (coefficients up to the level's modulus: K 64-bit words)
    hipError_t k_ckks_decompose(const ModDesc *mods, const double2 *values, uint64_t *out, unsigned n_log, unsigned K, unsigned batch, int mode,
                                hipStream_t s);
    // decode_internal's CRT composition and scaling (ckks.h:741-781; RNSBase::compose_array, rns.cpp:300-360): coefficient-form
    // residues [batch][K][N] -> complex values (imaginary part 0).  punct = [K][K] words (Q / q_j), inv_punct = [K] Shoup pairs,
    // q_words / half_words = Q and (Q + 1) / 2 as K words
    hipError_t k_ckks_compose_scale(const ModDesc *mods, const uint64_t *residues, const uint64_t *punct, const ShoupOp *inv_punct,
                                    const uint64_t *q_words, const uint64_t *half_words, double inv_scale, double2 *out, unsigned n_log, unsigned K,
                                    unsigned batch, hipStream_t s);
    // max over the coefficients of vector b of the bit length of the centred CRT value of (m * residue): out_bits[b] (zeroed
    // beforehand) - the norm of Decryptor::invariant_noise_budget (decryptor.cpp:222-241; poly_infty_norm_coeffmod)
    hipError_t k_crt_norm_bits(const ModDesc *mods, const uint64_t *residues, const uint64_t *punct, const ShoupOp *inv_punct,
                               const uint64_t *q_words, const uint64_t *half_words, uint64_t m, unsigned *out_bits, unsigned n_log, unsigned K,
                               unsigned batch, hipStream_t s);
    // slot <-> coefficient index map: scatter out[map[i]] = in[i] and out[map[i + slots]] = conj(in[i]) for i < count (rest 0
    // beforehand), gather out[i] = in[map[i]] for i < slots
    hipError_t k_ckks_place(const uint32_t *map, const double2 *in, double2 *out, unsigned n_log, unsigned count, hipStream_t s);
    hipError_t k_ckks_gather(const uint32_t *map, const double2 *in, double2 *out, unsigned n_log, hipStream_t s);
} // namespace sealhip
