// Device mirror of a SEALContext: the modulus-switching chain and every per-level constant
// the hot path reads, built on the host with the reference's own definitions and uploaded once.
//
// Mirrors (by behaviour, not by structure):
//   SEALContext / ContextData chain            native/src/seal/context.cpp:495-575, context.h:322-346
//   NTTTables::initialize                      native/src/seal/util/ntt.cpp:241-300
//   RNSTool::initialize (BEHZ bases, inverses) native/src/seal/util/rns.cpp:578-787
//   RNSBase::initialize (CRT data)             native/src/seal/util/rns.cpp:212-257
//   BaseConverter::initialize                  native/src/seal/util/rns.cpp:541-562
// HBM layout: one "prime pool" (coefficient primes first, then the BEHZ auxiliary primes) with
// ModDesc[], forward/inverse twiddle tables [prime][N] of 16-byte Shoup pairs, and one flat
// constant block per level.
#pragma once
#include "hostmath.h"
#include "ntt_kernels.h"
#include <memory>
#include <string>
#include <vector>

namespace sealhip
{
    enum class Scheme : uint8_t
    {
        none = 0,
        bfv = 1,
        ckks = 2,
        bgv = 3
    };

    typedef uint64_t parms_id_type[4];

    constexpr unsigned kMaxComps = 64; // SEAL_COEFF_MOD_COUNT_MAX (defines.h:52)

    // Constants of one level, as the kernels see them (device pointers into one allocation).
    struct LevelDev
    {
        unsigned K = 0;     // coeff_modulus_size at this level
        unsigned nB = 0;    // |B|
        unsigned nBsk = 0;  // |B| + 1
        // rescale / mod-switch (RNSTool::inv_q_last_mod_q_, rns.cpp:769-776)
        const ShoupOp *inv_q_last_mod_q = nullptr; // [K-1]
        // q_i - (floor(q_last/2) mod q_i): the rounding correction of rns.cpp:874-877 /
        // evaluator.cpp:2831, added on load by the NTT (NttBatch::src_mode 2)
        const uint64_t *round_fix = nullptr;       // [K-1]
        // floor(q_last/2) mod q_i (coefficient-domain variant, rns.cpp:816-817)
        const uint64_t *half_mod_q = nullptr;      // [K-1]
        uint64_t q_last = 0, half_q_last = 0;
        // BGV (mod_t_and_divide_q_last_ntt_inplace, rns.cpp:1193-1236): q_last mod q_i and q_last^-1 mod t
        // (RNSTool::inv_q_last_mod_t_, rns.cpp:778-786); unused unless the scheme is BGV
        const uint64_t *q_last_mod_q = nullptr;    // [K-1]
        uint64_t inv_q_last_mod_t = 0;
        // plaintext lifting and scaling (BFV/BGV; context.cpp:332-376 of the reference):
        //   delta_mod_q[i] = floor(Q/t) mod q_i ("coeff_div_plain_modulus"), upper_half_inc[i] = (Q - t) mod q_i,
        //   q_mod_t = Q mod t, plain_upper_half_threshold = (t+1)/2
        const uint64_t *delta_mod_q = nullptr;     // [K]
        const uint64_t *upper_half_inc = nullptr;  // [K]
        uint64_t q_mod_t = 0;
        uint64_t plain_upper_half_threshold = 0;
        // decryption (RNSTool::decrypt_scale_and_round, rns.cpp:1133-1191: BFV; RNSTool::decrypt_modt = BaseConverter::
        // exact_convert_array, rns.cpp:1193-1198 / 465-540: BGV); null for CKKS
        const ShoupOp *dec_inv_punct_q = nullptr;     // [K] (Q/q_i)^-1 mod q_i
        const uint64_t *dec_q_to_t = nullptr;         // [K] (Q/q_i) mod t
        const ShoupOp *dec_prod_t_gamma_mod_q = nullptr; // [K] t*gamma mod q_i                      (BFV)
        const uint64_t *dec_q_to_gamma = nullptr;     // [K] (Q/q_i) mod gamma                       (BFV)
        uint64_t dec_neg_inv_q_mod_t = 0, dec_neg_inv_q_mod_gamma = 0, dec_inv_gamma_mod_t = 0; // (BFV)
        uint32_t gamma_prime = 0;                     // pool index of gamma                           (BFV)
        // BEHZ (BFV multiply); all null when the scheme is CKKS
        const uint32_t *bsk_prime = nullptr;       // [nBsk] pool index of each Bsk prime (B..., m_sk)
        const ShoupOp *inv_punct_q = nullptr;      // [K]     (Q/q_i)^-1 mod q_i
        const ShoupOp *m_tilde_mod_q = nullptr;    // [K]     m~ mod q_i as multiplier
        const uint64_t *q_to_bsk = nullptr;        // [nBsk][K]   (Q/q_i) mod p_j
        const uint64_t *q_to_mtilde = nullptr;     // [K]         (Q/q_i) mod m~
        const uint64_t *prod_q_mod_bsk = nullptr;  // [nBsk]
        const ShoupOp *inv_mtilde_mod_bsk = nullptr; // [nBsk]
        const ShoupOp *inv_prod_q_mod_bsk = nullptr; // [nBsk]
        const ShoupOp *inv_punct_b = nullptr;      // [nB]    (B/b_i)^-1 mod b_i
        const uint64_t *b_to_q = nullptr;          // [K][nB]     (B/b_i) mod q_j
        const uint64_t *b_to_msk = nullptr;        // [nB]        (B/b_i) mod m_sk
        const uint64_t *prod_b_mod_q = nullptr;    // [K]
        const ShoupOp *t_mod_q = nullptr;          // [K]     plain modulus as multiplier
        const ShoupOp *t_mod_bsk = nullptr;        // [nBsk]
        // the same constants multiplied together where the reference applies them one after the other (round 3: every product by
        // a constant that is followed by another product by a constant is one product by their product - the residues are the
        // same, the kernels do 30 Shoup products per coefficient instead of 74)
        const ShoupOp *mt_inv_punct_q = nullptr;   // [K]     m~ (Q/q_i)^-1 mod q_i                         (lift)
        const uint64_t *q_to_bsk_lift = nullptr;   // [nBsk][K][2] (Q/q_i) m~^-1 mod p_j, cut into 21-bit limbs      (lift)
        const uint64_t *prod_q_lift = nullptr;     // [nBsk]      Q m~^-1 mod p_j                           (lift)
        const ShoupOp *t_inv_punct_q = nullptr;    // [K]     t (Q/q_i)^-1 mod q_i                          (floor)
        const uint64_t *q_to_bsk_floor = nullptr;  // [nBsk][K][2] (Q/q_i) Q^-1 [(B/b_j)^-1, j < nB] mod p_j, limbs (floor)
        const ShoupOp *t_floor_bsk = nullptr;      // [nBsk]      t Q^-1 [(B/b_j)^-1, j < nB] mod p_j       (floor)
        // 2^64 mod p as a Shoup operand: folds the high word of a 128-bit dot product into the low one (behz_kernels.hip)
        const ShoupOp *two64_bsk = nullptr;        // [nBsk]
        const ShoupOp *two64_q = nullptr;          // [K]
        const uint64_t *b_to_q3 = nullptr, *b_to_msk3 = nullptr; // b_to_q / b_to_msk cut into 21-bit limbs: two words per entry {limb0 | limb1 << 32, limb2}
        const uint64_t *neg_base_q = nullptr;      // [K] the multiple of q_i just above 2^60: (neg_base - x) is -x mod q_i, non-negative, for x <= 2^60
        ShoupOp inv_prod_b_mod_msk{ 0, 0 };
        uint64_t neg_inv_prod_q_mod_mtilde = 0;
        uint64_t m_tilde = 0;
        uint32_t msk_prime = 0;                    // pool index of m_sk
    };

    struct Level
    {
        size_t chain_index = 0;
        unsigned K = 0;
        parms_id_type parms_id{ 0, 0, 0, 0 };
        int total_coeff_modulus_bit_count = 0;
        std::vector<uint64_t> bsk;  // host copy of Bsk primes (B..., m_sk)
        LevelDev dev;
        void *dev_block = nullptr;  // owning allocation behind dev's pointers
    };

    class Context
    {
    public:
        // Throws std::invalid_argument / std::logic_error with the reference's conditions
        // (context.cpp:142-460) when the parameters are not usable.
        // what SEALContext::validate checks before its security verdict; throws std::invalid_argument with the reference's error names
        static void check_basic_parameters(Scheme scheme, size_t poly_modulus_degree, const std::vector<uint64_t> &coeff_modulus);
        Context(Scheme scheme, size_t poly_modulus_degree, const std::vector<uint64_t> &coeff_modulus,
                uint64_t plain_modulus, bool expand_mod_chain);
        ~Context();
        Context(const Context &) = delete;
        Context &operator=(const Context &) = delete;

        Scheme scheme() const { return scheme_; }
        size_t n() const { return n_; }
        int log_n() const { return log_n_; }
        uint64_t plain_modulus() const { return plain_modulus_; }
        const std::vector<uint64_t> &coeff_modulus() const { return primes_; }
        bool using_keyswitching() const { return using_keyswitching_; }
        bool using_batching() const { return using_batching_; }
        // sec_level_type of SEALContext's constructor (0 none, 128, 192, 256): recorded for EncryptionParameterQualifiers; the
        // C ABI checks it against CoeffModulus::MaxBitCount before it builds the context (context.cpp:219-231 of the reference)
        int sec_level() const { return sec_level_; }
        void set_sec_level(int s) { sec_level_ = s; }

        // chain: levels_[0] is the key level (chain_index = size-1) ... back() has chain_index 0
        const std::vector<Level> &levels() const { return levels_; }
        const Level &key_level() const { return levels_.front(); }
        const Level &first_level() const { return levels_[using_keyswitching_ ? 1 : 0]; }
        const Level &last_level() const { return levels_.back(); }
        const Level *level_by_chain_index(size_t chain_index) const;
        const Level *level_by_parms_id(const uint64_t *parms_id) const;
        const Level *next_level(const Level &l) const;
        void set_parms_id(size_t chain_index, const uint64_t *parms_id);

        // prime pool
        const std::vector<uint64_t> &pool_primes() const { return pool_; }
        unsigned aux_first() const { return (unsigned)primes_.size(); } // pool index of m_sk
        // pool index of the plain modulus t when it supports batching (BFV / BGV, t prime, t = 1 mod 2N): its NTT tables are
        // SEALContext::ContextData::plain_ntt_tables() (context.cpp:415-425 of the reference); -1 otherwise
        int plain_prime_index() const { return plain_prime_; }
        const NttTables &ntt_tables() const { return tables_; }
        const ModDesc *dev_mods() const { return d_mods_; }
        // host copies (tests / introspection)
        uint64_t ntt_root(unsigned pool_index) const { return roots_[pool_index]; }
        const std::vector<ModDesc> &host_mods() const { return h_mods_; }
        // true when pool prime p runs on the double-precision back end (field.h)
        bool fp_prime(unsigned pool_index) const { return h_fpd_[pool_index].qi != 0; }

        // special prime P^-1 mod q_i for key switching (key level's inv_q_last_mod_q)
        const ShoupOp *dev_inv_special_mod_q() const { return key_level().dev.inv_q_last_mod_q; }

    private:
        void build_pool_and_tables();
        void build_level(Level &lvl);

        Scheme scheme_;
        size_t n_;
        int log_n_;
        uint64_t plain_modulus_;
        std::vector<uint64_t> primes_;  // coefficient primes (key level order)
        std::vector<uint64_t> pool_;    // primes_ + [m_sk, gamma, B_0, B_1, ...]
        std::vector<uint64_t> roots_;   // minimal primitive 2N-th root per pool prime (0 = none)
        std::vector<ModDesc> h_mods_;
        std::vector<Level> levels_;
        bool using_keyswitching_ = false;
        bool using_batching_ = false;
        int sec_level_ = 0;
        int plain_prime_ = -1;

        ModDesc *d_mods_ = nullptr;
        ShoupOp *d_fwd_ = nullptr;
        ShoupOp *d_inv_ = nullptr;
        ShoupOp *d_ninv_ = nullptr;
        FpDesc *d_fpd_ = nullptr;
        double *d_fwd_d_ = nullptr, *d_inv_d_ = nullptr, *d_ninv_d_ = nullptr;
        std::vector<FpDesc> h_fpd_;
        std::vector<unsigned char> h_fp_flag_;
        NttTables tables_{};
    };
} // namespace sealhip
