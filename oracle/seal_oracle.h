/*
 * seal_oracle.h — TEST INFRASTRUCTURE ONLY (see seal_oracle.c).  Plain-C restatement of the
 * reference's hot path used as the parity checker ("port" oracle).  Never linked into, loaded by or
 * shipped with seal_amd/.
 */
#ifndef SEAL_ORACLE_H
#define SEAL_ORACLE_H
#include <stdint.h>

#define SO_MAX_PRIMES 64

typedef struct so_ctx so_ctx;

/* scheme: 1 bfv, 2 ckks, 3 bgv.  primes = full coeff_modulus (last one is the special key-switching prime). */
so_ctx *so_ctx_create(int scheme, uint64_t n, const uint64_t *primes, int count, uint64_t plain_modulus);
void so_ctx_destroy(so_ctx *c);
uint64_t so_ntt_root(const so_ctx *c, int prime_index);
/* BEHZ base Bsk = (B..., m_sk) of the level with K data primes; returns |Bsk| */
int so_base_bsk(const so_ctx *c, int K, uint64_t *out);
/* helpers restating CoeffModulus::Create / get_primes / PlainModulus::Batching */
int so_get_primes(uint64_t factor, int bit_size, int count, uint64_t *out);
int so_coeff_modulus_create(uint64_t n, const int *bit_sizes, int count, uint64_t *out);

/* One polynomial component, in place, canonical output.  prime_index indexes the context's primes;
 * aux != 0 selects the BEHZ prime pool instead (index into so_base_bsk order of the top level). */
void so_ntt_forward(const so_ctx *c, int prime_index, uint64_t *a);
void so_ntt_inverse(const so_ctx *c, int prime_index, uint64_t *a);
/* O(N^2) definition of the same transform (SURVEY §8(a')), for pinning so_ntt_forward */
void so_ntt_forward_naive(const so_ctx *c, int prime_index, const uint64_t *a, uint64_t *out);

/* Ciphertext-level operations on one ciphertext with K data primes; slabs are [poly][K][N]. */
void so_dyadic(const so_ctx *c, int prime_index, const uint64_t *a, const uint64_t *b, uint64_t *r);
void so_ckks_multiply(const so_ctx *c, int K, const uint64_t *x, int sx, const uint64_t *y, int sy, uint64_t *out);
int so_bfv_multiply(const so_ctx *c, int K, const uint64_t *x, int sx, const uint64_t *y, int sy, uint64_t *out);
/* ct (2 polys, in/out) += KS(target); key = [digits][2][L][N] */
void so_switch_key(const so_ctx *c, int K, uint64_t *ct, const uint64_t *target, const uint64_t *key);
/* CKKS rescale / BFV mod_switch_to_next: in [size][K][N] -> out [size][K-1][N] */
void so_rescale(const so_ctx *c, int K, const uint64_t *in, int size, uint64_t *out);
void so_bfv_mod_switch(const so_ctx *c, int K, const uint64_t *in, int size, uint64_t *out);
void so_drop_last(const so_ctx *c, int K, const uint64_t *in, int size, uint64_t *out);
/* BGV mod_switch_to_next (rns.cpp:1193-1236) and the correction factor it leaves (evaluator.cpp:1286-1292) */
void so_bgv_mod_switch(const so_ctx *c, int K, const uint64_t *in, int size, uint64_t *out);
uint64_t so_bgv_mod_switch_correction(const so_ctx *c, int K, uint64_t correction_factor);
/* plaintext operands: centred lift of `count` coefficients mod t into [K][N] residues (times scale_by mod t);
 * BFV add_plain / sub_plain on c0 = [K][N] (util/scalingvariant.cpp:70-175) */
void so_plain_lift(const so_ctx *c, int K, const uint64_t *m, uint64_t count, uint64_t scale_by, uint64_t *out);
void so_bfv_addsub_plain(const so_ctx *c, int K, uint64_t *c0, const uint64_t *m, uint64_t count, int sub);
/* one polynomial [K][N] */
void so_apply_galois(const so_ctx *c, int K, int ntt_form, uint32_t elt, const uint64_t *in, uint64_t *out);
uint32_t so_galois_elt_from_step(const so_ctx *c, int step);
/* BEHZ stages on one polynomial (see shl_rns_stage numbering 0..3) */
int so_rns_stage(const so_ctx *c, int K, int which, const uint64_t *in, uint64_t *out);

/* CPU baseline ("port"): seconds for `reps` x (multiply + relinearize + rescale) on one thread */
double so_time_ckks_pipeline(const so_ctx *c, int K, const uint64_t *a, const uint64_t *b, const uint64_t *rlk, int reps,
                             uint64_t *out);
#endif
