// Element-wise RNS polynomial kernels (HBM-bound glue between the transforms) and the
// key-switching inner product.  Launch wrappers only; kernels are in poly_kernels.hip.
//
// Reference functions covered (results canonical, hence bit-identical):
//   dyadic_product_coeffmod / add_/sub_/negate_poly_coeffmod   util/polyarithsmallmod.cpp:43-284
//   ckks_multiply tile loop (x0y0, x0y1+x1y0, x1y1)            evaluator.cpp:604-663
//   GaloisTool::apply_galois / apply_galois_ntt                 util/galois.cpp:148-218
//   divide_and_round_q_last[_ntt]_inplace tail                  util/rns.cpp:789-901
//   switch_key_inplace inner product and mod-down tail          evaluator.cpp:2663-2864
//
// Plane layout everywhere: a "plane" is one polynomial of every batch item, [batch][K][N] words.
#pragma once
#include "context.h"

namespace sealhip
{
    struct PlaneGeom
    {
        unsigned n_log;   // log2 N
        unsigned K;       // components per item in this plane
        unsigned batch;   // items
        size_t words() const { return ((size_t)batch * K) << n_log; }
    };

    // out (size 3) = x (size 2) * y (size 2); out may be x (in place).  Planes are plane_words apart.
    // comp_prime (device, may be null = identity) maps a component to its pool prime.
    // fpd (device, may be null = integer arithmetic for every prime): the double-precision back end's descriptors (ntt_kernels.h).
    hipError_t k_ckks_multiply_2x2(
        const ModDesc *mods, const FpDesc *fpd, const uint32_t *comp_prime, const uint64_t *x, const uint64_t *y, uint64_t *out, PlaneGeom g,
        hipStream_t s);
    // out[I] = sum_a x[a] * y[I-a] for general sizes; out must not alias x or y.
    hipError_t k_multiply_general(
        const ModDesc *mods, const uint32_t *comp_prime, const uint64_t *x, unsigned size_x, const uint64_t *y,
        unsigned size_y, uint64_t *out, PlaneGeom g, hipStream_t s);
    // r = a .* b over `count` polys of `comps` comps (first_prime..): the raw dyadic seam
    hipError_t k_dyadic(
        const ModDesc *mods, const uint64_t *a, const uint64_t *b, uint64_t *r, unsigned n_log, unsigned comps,
        unsigned first_prime, size_t polys, hipStream_t s);
    // op: 0 add, 1 sub, 2 negate (b ignored).  `planes` planes of geometry g.
    hipError_t k_addsub(
        const ModDesc *mods, const uint64_t *a, const uint64_t *b, uint64_t *r, int op, PlaneGeom g, unsigned planes,
        hipStream_t s);
    // r = a * (scalar mod q_i) mod q_i over `planes` planes (multiply_poly_scalar_coeffmod,
    // util/polyarithsmallmod.h:440-448 / .cpp:197-224); r may be a.
    hipError_t k_mul_scalar(
        const ModDesc *mods, const uint64_t *a, uint64_t *r, uint64_t scalar, PlaneGeom g, unsigned planes, hipStream_t s);
    // ---- plaintext operands (one polynomial applied to every batch item)
    // r[item][k] = a[item][k] .* p[k]: multiply_plain_ntt (evaluator.cpp:2157-2194); items = size * batch
    hipError_t k_dyadic_plain(const ModDesc *mods, const uint64_t *a, const uint64_t *p, uint64_t *r, unsigned n_log, unsigned K,
                              size_t items, hipStream_t s);
    // c[item][k] (+/-)= p[k] for the items of one plane (op 0 add, 1 sub): CKKS / BGV add_plain, sub_plain
    hipError_t k_addsub_plain(const ModDesc *mods, uint64_t *c, const uint64_t *p, int op, unsigned n_log, unsigned K, size_t items,
                              hipStream_t s);
    // plaintext coefficients modulo t -> RNS form [K][N] (the lift of transform_to_ntt_inplace / multiply_plain_normal,
    // evaluator.cpp:2098-2125, 2243-2282): out[i][j] = m_j mod q_i, plus upper_half_inc[i] (mod q_i) when
    // m_j >= threshold; zero beyond coeff_count.  scale_by != 1 first multiplies m_j by it modulo t (BGV add_plain).
    hipError_t k_plain_lift(const ModDesc *mods, ModDesc t, const uint64_t *m, size_t coeff_count, uint64_t scale_by,
                            uint64_t threshold, const uint64_t *upper_half_inc, uint64_t *out, unsigned n_log, unsigned K, hipStream_t s);
    // nonzero_coeff_count, significant_coeff_count - 1 and the coefficient there (plaintext.h:371-399), written to
    // stats[0..2] (device): decides the monomial shortcut of multiply_plain_normal (evaluator.cpp:2051-2095)
    hipError_t k_plain_stats(const uint64_t *m, size_t coeff_count, uint64_t *stats, hipStream_t s);
    // negacyclic_multiply_poly_mono_coeffmod (util/polyarithsmallmod.cpp:286-334): out = in * (scalar_k x^e) per component
    // k, scalar_k canonical modulo q_k (device array [K]); out != in
    hipError_t k_negacyclic_mul_mono(const ModDesc *mods, const uint64_t *in, uint64_t *out, const uint64_t *scalars, size_t e,
                                     unsigned n_log, unsigned K, size_t items, hipStream_t s);
    // BFV add_plain / sub_plain: multiply_add/sub_plain_with_scaling_variant (util/scalingvariant.cpp:70-175):
    //   fix = floor((m (Q mod t) + (t+1)/2) / t);  c0[item][i][j] (+/-)= (m delta_i + fix) mod q_i
    hipError_t k_bfv_addsub_plain(const ModDesc *mods, ModDesc t, const uint64_t *m, size_t coeff_count, uint64_t q_mod_t,
                                  uint64_t threshold, const uint64_t *delta_mod_q, uint64_t *c0, int op, unsigned n_log, unsigned K,
                                  size_t items, hipStream_t s);
    // Galois automorphism on `planes` planes; ntt_form selects the NTT-domain gather or the
    // coefficient-domain signed scatter.  in != out.
    hipError_t k_apply_galois(
        const ModDesc *mods, const uint64_t *in, uint64_t *out, uint32_t galois_elt, int ntt_form, PlaneGeom g,
        unsigned planes, hipStream_t s);

    // CKKS rescale tail: out[item][i] = (c[item][i] - t[item][i]) * q_last^-1 mod q_i, i < K-1;
    // c has K comps per item, t and out K-1; t is lazy (< 4 q_i).
    hipError_t k_rescale_combine(
        const ModDesc *mods, const ShoupOp *inv_q_last, const uint64_t *c, const uint64_t *t, uint64_t *out,
        unsigned n_log, unsigned K, size_t items, hipStream_t s);
    // BGV correction polynomial of mod_t_and_divide_q_last_ntt_inplace (rns.cpp:1203-1229) and of the
    // BGV key-switch mod-down (evaluator.cpp:2762-2791), coefficient form:
    //   k = -(c mod t) * q_last^-1 mod t;   delta[item][i] = ((k mod q_i) * (q_last mod q_i) + (c mod q_i)) mod q_i
    // c = c_last + item*c_stride, canonical mod q_last; delta [items][ncomp][N] canonical.
    hipError_t k_bgv_delta(
        const ModDesc *mods, ModDesc t, uint64_t inv_q_last_mod_t, const uint64_t *q_last_mod_q, const uint64_t *c_last,
        size_t c_stride, uint64_t *delta, unsigned n_log, unsigned ncomp, size_t items, hipStream_t s);
    // BFV mod-switch (coefficient domain), whole formula in one kernel.
    hipError_t k_bfv_modswitch(
        const ModDesc *mods, const LevelDev &lv, const uint64_t *c, uint64_t *out, unsigned n_log, size_t items,
        hipStream_t s);
    // drop the last component: out[item][i] = c[item][i], i < K-1
    hipError_t k_drop_last(const uint64_t *c, uint64_t *out, unsigned n_log, unsigned K, size_t items, hipStream_t s);

    // Key-switch inner product.  u: [batch][K+1][K][N] (target digit J raised to modulus I, NTT form,
    // canonical); key: [digits][2][L][N]; acc: [batch][2][K+1][N] canonical.  Modulus index I == K
    // means the special prime (pool/key component L-1).
    // Only the digits [j0, j1) are summed; key holds the digits from key_digit0 on.
    hipError_t k_keyswitch_mac(
        const ModDesc *mods, const uint64_t *u, const uint64_t *key, uint64_t *acc, unsigned n_log, unsigned K,
        unsigned L, unsigned batch, unsigned j0, unsigned j1, unsigned key_digit0, hipStream_t s);
    // acc <- acc mod q_I in place after partial sums of several ranks were added (each canonical, at most 8 of them)
    // c0 / c1 / pm (optional): the data-prime components leave as c_k[item][I] + (the sum) pm[I] mod q_I - KsFusedArgs::fold_c0 for a
    // key switch whose digits ran as several in-launch groups or on several ranks.  out (optional): the result goes there, acc is only read
    hipError_t k_keyswitch_reduce(
        const ModDesc *mods, uint64_t *acc, unsigned n_log, unsigned K, unsigned L, unsigned batch, hipStream_t s, unsigned local_parts = 1,
        const uint64_t *c0 = nullptr, const uint64_t *c1 = nullptr, const ShoupOp *pm = nullptr, uint64_t *out = nullptr);
    // acc[outer][K+1][N], component K - 1 <- (it - (component K mod q + *fix) * *pinv + half_last) mod q, q = mods[prime]: the
    // coefficient form of the key-switched ciphertext's last component, ready for the rescale (poly_kernels.hip)
    hipError_t k_ks_last_coeff(
        const ModDesc *mods, unsigned prime, const ShoupOp *pinv, const uint64_t *fix, uint64_t half_last, uint64_t *acc, unsigned n_log,
        unsigned K, size_t nouter, hipStream_t s);
    // Reduce-scatter exchange of the digit-parallel key switch (evaluator_keyswitch.cpp: switch_key_exchange_*).  The K data
    // moduli are owned by the G ranks in contiguous ranges whose sizes differ by at most one (rank c: K/G (+1 for c < K%G)
    // moduli); m = ceil(K / G) slots per rank.
    //   pack_targets : send[c][s][b][k][N] = acc[b][k][first(c)+s] (zero padding), sp[b][k][N] = acc[b][k][K] (special prime)
    //   unpack_owned : acc3[b][k][s] = recv[s][b][k] mod q_{first+s},  acc3[b][k][count] = sp[b][k] mod P  (Barrett, sums of <= 8
    //                  parts): the layout [batch][2][count+1][N] of a key switch over the rank's own `count` moduli
    //   pack_owned   : own[s][b][k][N] = inc[k][b][s][N]  (inc: this rank's increments, compact [2][batch][count][N])
    //   add_gathered : ct_k[b][i] += all[owner(i)][slot(i)][b][k]  mod q_i
    hipError_t k_ks_pack_targets(
        const uint64_t *acc, uint64_t *send, uint64_t *sp, unsigned n_log, unsigned K, unsigned G, unsigned m, unsigned batch, hipStream_t s);
    hipError_t k_ks_unpack_owned(
        const ModDesc *mods, const uint64_t *recv, const uint64_t *sp, uint64_t *acc3, unsigned n_log, unsigned L, unsigned first,
        unsigned count, unsigned batch, hipStream_t s);
    hipError_t k_ks_pack_owned(const uint64_t *inc, uint64_t *own, unsigned n_log, unsigned count, unsigned m, unsigned batch, hipStream_t s);
    hipError_t k_ks_add_gathered(
        const ModDesc *mods, uint64_t *ct0, uint64_t *ct1, const uint64_t *all, unsigned n_log, unsigned K, unsigned G, unsigned m,
        unsigned batch, hipStream_t s);
    // Key-switch tail (CKKS): ct_k[b][i] += (acc[b][k][i] - t[b][k][i]) * P^-1 mod q_i.
    // ct planes: ct0 and ct1, each [batch][K][N]; acc [batch][2][K+1][N]; t [batch][2][K][N] lazy.
    hipError_t k_keyswitch_tail_ckks(
        const ModDesc *mods, const ShoupOp *inv_p, uint64_t *ct0, uint64_t *ct1, const uint64_t *acc,
        const uint64_t *t, unsigned n_log, unsigned K, unsigned batch, hipStream_t s);
    // Key-switch tail (BFV): acc comps 0..K-1 already in coefficient form (canonical); r = acc comp K
    // in coefficient form canonical; ct_k[b][i] += (acc_i - ((r+half mod P) mod q_i - half mod q_i)) * P^-1.
    hipError_t k_keyswitch_tail_bfv(
        const ModDesc *mods, const ShoupOp *inv_p, const uint64_t *round_fix, uint64_t half_p, uint64_t p,
        uint64_t *ct0, uint64_t *ct1, const uint64_t *acc, unsigned n_log, unsigned K, unsigned batch, hipStream_t s);
    // The same tail followed by the BFV mod-switch to the next level in one pass: out [2][batch][K-1][N] = divide_and_round_q_last of
    // the completed ciphertext (rns.cpp:789-828), which is never written.  klv = the key level's constants (P^-1, P/2 fixes), lv = the
    // ciphertext's level (q_last^-1, q_last/2).
    hipError_t k_keyswitch_tail_modswitch_bfv(
        const ModDesc *mods, const LevelDev &klv, uint64_t half_p, uint64_t p, const LevelDev &lv, const uint64_t *ct0, const uint64_t *ct1,
        const uint64_t *acc, uint64_t *out, unsigned n_log, unsigned K, unsigned batch, hipStream_t s);
    // BEHZ (BFV multiply) per-coefficient base conversions, util/rns.cpp:903-1131.
    // lift: fastbconv_m_tilde + sm_mrq.  in [items][K][N] canonical -> out [items][nBsk][N] canonical.
    hipError_t k_behz_lift(const ModDesc *mods, const LevelDev &lv, const uint64_t *in, uint64_t *out, unsigned n_log,
                           size_t items, hipStream_t s);
    // floor_sk: (x t) -> fast_floor -> fastbconv_sk.  dq [items][K][N] (< 2q), dbsk [items][nBsk][N] (< 2p)
    // -> out [items][K][N] canonical.
    hipError_t k_behz_floor_sk(const ModDesc *mods, const LevelDev &lv, const uint64_t *dq, const uint64_t *dbsk,
                               uint64_t *out, unsigned n_log, size_t items, hipStream_t s);
    // the four RNSTool stages separately (per-kernel parity seam): see shl_rns_stage in sealhip.h
    hipError_t k_behz_stage(const ModDesc *mods, const LevelDev &lv, int which, const uint64_t *in, uint64_t *out,
                            unsigned n_log, size_t items, hipStream_t s);
    // any nonzero word in [data, data+words)?  *flag (device) |= 1
    hipError_t k_any_nonzero(const uint64_t *data, size_t words, unsigned *flag, hipStream_t s);
} // namespace sealhip
