// seal::KeyGenerator on the device (native/src/seal/keygenerator.{h,cpp}): the secret key, the public key and the
// key-switching keys (RelinKeys / GaloisKeys) the hot path consumes, generated in HBM with the reference's algorithm and the
// reference's randomness, so that under the reference's seeded factory (Blake2xbPRNGFactory(seed): every create() restarts the
// stream) every key is word-for-word the reference's.  Built from the Encryptor's encrypt_zero_symmetric (decryptor.h) and the
// device samplers (xof_kernels.h); a C5 RelinKeys set (15 digits x 2 x 16 x 65536 words = 252 MB) is 15 symmetric encryptions.
#pragma once
#include "decryptor.h"
#include <memory>

namespace sealhip
{
    class KeyGenerator
    {
    public:
        // KeyGenerator(context) (keygenerator.cpp:24-37): samples a fresh secret key; seed8 != nullptr = the seeded factory
        KeyGenerator(const Context &context, const uint64_t *seed8);
        // KeyGenerator(context, secret_key) (keygenerator.cpp:39-54)
        KeyGenerator(const Context &context, const SecretKey &secret_key, const uint64_t *seed8);
        KeyGenerator(const KeyGenerator &) = delete;
        KeyGenerator &operator=(const KeyGenerator &) = delete;

        const SecretKey &secret_key() const { return sk_; }
        // create_public_key (keygenerator.cpp:93-121): encrypt_zero_symmetric at the key level, NTT form
        void create_public_key(PublicKey &destination);
        // create_relin_keys(count = 1) (keygenerator.cpp:123-157): the key for s^2
        void create_relin_keys(KSwitchKeys &destination);
        // create_galois_keys(galois_elts) (keygenerator.cpp:159-209): one key per element, at GaloisKeys::get_index(elt)
        void create_galois_keys(const uint32_t *galois_elts, size_t count, KSwitchKeys &destination);
        // create_galois_keys(steps) (keygenerator.h:207-232: needs batching; GaloisTool::get_elts_from_steps) and create_galois_keys()
        // (all the elements GaloisTool::get_elts_all lists: 3^(2^i), 3^-(2^i), 2N - 1)
        void create_galois_keys_from_steps(const int *steps, size_t count, KSwitchKeys &destination);
        void create_galois_keys_all(KSwitchKeys &destination);
        std::vector<uint32_t> galois_elts_all() const;
        // The Serializable<RelinKeys> / Serializable<GaloisKeys> forms, saved (create_relin_keys() / create_galois_keys(elts) with
        // save_seed = true, then save(compr_mode none); kswitchkeys.cpp:47-90): every digit as its SEEDED ciphertext - c_0 and the
        // 64-byte seed c_1 re-expands from - i.e. half the bytes of the full key set.  galois = false: the relinearization key.
        // Returns the bytes written; seeded_save_size = the capacity that suffices.
        size_t seeded_save_size(bool galois, size_t key_count) const;
        size_t save_seeded(bool galois, const uint32_t *galois_elts, size_t count, uint8_t *out, size_t capacity);
        // one key in the reference's layout [digit][2][L][N] (KSwitchKeys::data()[index][digit].data()) copied to host memory:
        // galois_elt == 0 -> the relinearization key.  For parity tests and for saving keys; regenerates the key.
        size_t key_words() const; // words per key
        void key_to_host(uint32_t galois_elt, uint64_t *host_words);

    private:
        void sample_secret_key();
        void configure(const uint64_t *seed8);
        // generate_one_kswitch_key (keygenerator.cpp:322-357): out_dev = [digits][2][L][N]
        void one_kswitch_key(const uint64_t *new_key_dev, uint64_t *out_dev, uint64_t *public_seeds = nullptr); // [digits][8]
        void relin_key(uint64_t *out_dev, uint64_t *public_seeds = nullptr);
        void galois_key(uint32_t galois_elt, uint64_t *out_dev, uint64_t *public_seeds = nullptr);
        const Context &context_;
        SecretKey sk_;
        std::unique_ptr<Encryptor> encryptor_;
        bool seeded_ = false;
        uint64_t seed_[8];
    };
} // namespace sealhip
