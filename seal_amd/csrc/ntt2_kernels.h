// Two-pass NTT engine (2^13 <= N <= 2^16) and fused key-switching kernels: launch interface.
// See ntt2_kernels.hip for the decomposition and field.h for the two arithmetic back ends.
#pragma once
#include "ntt_kernels.h"

namespace sealhip
{
    bool ntt2_supports(int log_n);

    // Same contract as ntt_forward (ntt_kernels.h); `mid` is a scratch buffer of
    // nouter * ncomp * N words that holds the transforms between the two passes (tile order).
    // `ring` (optional): scratch of ntt2_ring_words(t, b) words for the one-launch kernel of N = 2^16 (plain in-place transforms of
    // double-precision components, batches large enough to loop: ntt2_kernels.hip, ntt2_fwd_ring).  ntt2_ring_words() is 0 when
    // the batch does not qualify; without a ring the two-launch engine runs.
    size_t ntt2_ring_words(const NttTables &t, const NttBatch &b);
    hipError_t ntt2_forward(const NttTables &t, const NttBatch &b, int out_lazy, uint64_t *mid, hipStream_t stream, uint64_t *ring = nullptr, size_t ring_words = 0);

    // The one-launch kernel's own interface (ntt2_ring.hip; used by ntt2_forward): a run of `nc` consecutive double-precision components
    // [c0, c0 + nc) of a plain in-place batch.  ntt2_ring_teams() = teams per component the device holds for such a run (0: the run does
    // not qualify - small batch, kernel switched off by SEALHIP_NTT_RING=0, emulated build), ntt2_ring_run_words() = the scratch it needs.
    struct NttRingRun
    {
        uint64_t *data;
        size_t outer_stride;
        const uint32_t *comp_prime;
        unsigned prime_first, ncomp, c0, nc, nouter;
        int lazy;
        unsigned teams_per_comp;
        uint64_t *ring;
        size_t ring_words;
    };
    unsigned ntt2_ring_teams(unsigned nc, unsigned nouter);
    size_t ntt2_ring_run_words(unsigned nc, unsigned teams_per_comp);
    bool ntt2_ring_stream_ok(hipStream_t stream); // false while `stream` is being captured into a graph
    hipError_t ntt2_ring_launch(const NttTables &t, const NttRingRun &run, hipStream_t stream);

    // Same contract as ntt_inverse; when b.src is set the input is read from src (natural order,
    // same component layout as data, stride src_outer_stride) and data is only written.
    hipError_t ntt2_inverse(const NttTables &t, const NttBatch &b, int out_lazy, uint64_t *mid, hipStream_t stream);

    // switch_key_inplace inner part (evaluator.cpp:2663-2755):
    //   acc[b][k][I] = sum_J NTT_I(t[b][J] mod q_I) (.) key[J][k][comp(I)]   canonical, natural order
    // t: [batch][K][N] coefficient form.  target_ntt (CKKS) = the same digits in NTT form, used for
    // I == J instead of transforming (null for BFV).  key: register order (key_to_register_order).
    // targets*: device arrays, one entry per target modulus, the integer-back-end (60-bit) moduli first:
    //   targets1: pairs (I, pool prime); targets2: triples (I, pool prime, key component)
    // Both passes run one launch per back end; the integer-back-end launches go to a side stream so that their
    // latency-bound workgroups share the CUs with the issue-bound double-precision ones.
    struct KsFusedArgs
    {
        const uint64_t *t;
        const uint64_t *target_ntt;
        const uint64_t *key;
        uint64_t *mid; // [batch][K+1][K][N] scratch
        uint64_t *acc; // [batch][2][K+1][N]
        const uint32_t *targets1, *targets2;
        unsigned ntargets, n_int; // all targets; how many of them (the leading ones) are integer-back-end moduli
        unsigned K, L, batch;
        // digit-parallel key switching: only the digits [j0, j1) contribute to acc;
        // `key` holds digits [key_digit0, key_digit0 + resident) of the full key
        unsigned j0, j1, key_digit0;
        // in-launch digit groups (0 or 1 = none): with `parts` > 1 the digits [j0, j1) are cut into `parts` slices handled by
        // separate workgroups, slice g writing its partial sums to acc + g * batch*2*(K+1)*N (acc holds `parts` such buffers,
        // added by k_keyswitch_reduce).  Multiplies the number of workgroups when batch * targets * tiles does not fill the chip.
        unsigned parts;
        // Optional (parts <= 1, the full digit range only): the data-prime components leave as c_k[item][I] + S_k[item][I] P^-1 mod q_I
        // instead of the bare sums - the first step of the key-switch tail (evaluator.cpp:2845-2863), taken while S is in
        // registers.  fold_c0 / fold_c1 = the ciphertext's two planes [batch][K][N] (fold_c1 null: the second addend is zero), fold_pm[I] = P^-1 mod q_I (device).  The
        // special-prime component is the plain sum either way.
        const uint64_t *fold_c0 = nullptr, *fold_c1 = nullptr;
        const ShoupOp *fold_pm = nullptr;
        // Round 6, instead of fold_c0 / fold_c1: the addend is a 2 x 2 tensor product that was never stored (Evaluator::multiply deferred
        // it, evaluator.h: LazyProduct) - c0 = x0 y0, c1 = x0 y1 + x1 y0 are formed in ks2's epilogue from the operands' planes
        // (fold_x / fold_y = plane 0 of x / y, [batch][K][N]; plane 1 is fold_plane words further).  Same restrictions as fold_c0.
        // With target_ntt null the diagonal terms (I == J) are x1 y1 formed in ks2 as well: the product is then never stored at all.
        const uint64_t *fold_x = nullptr, *fold_y = nullptr;
        size_t fold_plane = 0;
        // Round 6, rotations without the permutation kernels: target_ntt and fold_c0 are the UNPERMUTED polynomials c1 and c0, read
        // through the NTT-domain automorphism of this Galois element (ntt2_device.h: galois_src_index); fold_c1 must be null (zero)
        uint32_t galois_elt = 0;
        // Callers that cut a batch into chunks (Evaluator::switch_key_partial, round 5) decide these for the WHOLE batch:
        // order1 = which pass-1 kernel runs (-1: decided from this call's grid, 0: target-resident ks1_kernel, 1: digit-resident
        // ks1t_kernel); no_class_fork = the two arithmetic classes run one after the other on `stream` (the caller's chunk
        // streams provide the concurrency) instead of forking the integer class to the launcher's side stream.
        int order1 = -1;
        bool no_class_fork = false;
    };
    hipError_t ks_fused(const NttTables &t, const KsFusedArgs &k, hipStream_t stream);

    // Register order of a key-switching key (what ks2 reads).  A decomposition digit is `key_digit_units(t, L)` units of N words,
    // its components one after the other: a prime of the double-precision back end takes 2 units - N pairs of balanced doubles,
    // (first, second key polynomial) of a coefficient side by side - a prime of the integer back end 4 units - N pairs
    // (word, floor(word 2^64 / q)) of the first polynomial, then N pairs of the second: the key is the precomputed operand of a Shoup
    // product (round 3: ks2's sums fit 64-bit words) and a pair is one 16-byte load.  Inside a component the position of
    // coefficient (tile hg, row-in-tile, column) is hg*4096 + e*256 + tid  <->  natural hg*4096 + (tid>>4)*256 + (tid&15)*16 + e.
    // (Round 3 gave every (polynomial, component) 2 N words and left the odd polynomial's slot of a double-precision prime unused:
    // 504 MB per C5 key instead of 284 - ADVICE r3.)
    inline unsigned key_comp_units(const NttTables &t, unsigned comp)
    {
        return t.fp_host && t.fp_host[comp] ? 2u : 4u;
    }
    inline unsigned key_comp_offset_units(const NttTables &t, unsigned comp) // of component `comp` inside a digit
    {
        unsigned u = 0;
        for (unsigned c = 0; c < comp; c++)
            u += key_comp_units(t, c);
        return u;
    }
    inline unsigned key_digit_units(const NttTables &t, unsigned L)
    {
        return key_comp_offset_units(t, L);
    }
    inline size_t key_register_order_words(const NttTables &t, unsigned L, size_t digits)
    {
        return (digits * key_digit_units(t, L)) << t.log_n;
    }
    // [digits][2][L][N] natural-order canonical key words -> register order (`out` holds key_register_order_words(t, L, digits) words)
    hipError_t key_to_register_order(const NttTables &t, const uint64_t *in, uint64_t *out, unsigned L, size_t digits, hipStream_t stream);
    // the way back: what a saved key holds.  Digits are self-contained: `in` may point at any digit of a resident key.
    hipError_t key_from_register_order(const NttTables &t, const uint64_t *in, uint64_t *out, unsigned L, size_t digits, hipStream_t stream);
} // namespace sealhip
