// Device-resident counterparts of seal::SecretKey and seal::Decryptor (the caller on the far side of the hot
// path).  Same method names, argument checks and exception classes as the reference (native/src/seal/decryptor.h:56-186,
// decryptor.cpp); the arithmetic reuses the NTT engine and three element-wise kernels (decrypt_kernels.h).
#pragma once
#include "decrypt_kernels.h"
#include "evaluator.h"
#include <mutex>
#include <vector>

namespace sealhip
{
    // seal::SecretKey (secretkey.h): one polynomial at the key level in NTT form, [L][N] words, resident in HBM
    class SecretKey
    {
    public:
        explicit SecretKey(const Context &ctx) : ctx_(&ctx) {}
        ~SecretKey();
        SecretKey(const SecretKey &) = delete;
        SecretKey &operator=(const SecretKey &) = delete;
        const Context &context() const { return *ctx_; }
        // words = SecretKey::data().data(): L*N residues, host memory (unaligned pointers into a stream are fine)
        void set(const void *host_words, size_t word_count);
        const uint64_t *data() const { return dev_; }

    private:
        const Context *ctx_;
        uint64_t *dev_ = nullptr;
    };

    class Decryptor
    {
    public:
        // Decryptor::Decryptor (decryptor.cpp:45-77): keeps its own copy of s; powers s^2.. are computed on demand
        Decryptor(const Context &context, const SecretKey &secret_key);
        ~Decryptor();
        Decryptor(const Decryptor &) = delete;
        Decryptor &operator=(const Decryptor &) = delete;

        // Decryptor::decrypt (decryptor.cpp:79-113) for a batch of one: CKKS -> NTT-form plaintext at the ciphertext's level
        // with its scale; BFV / BGV -> coefficients modulo t, trimmed to the significant ones.  Synchronises.
        void decrypt(const Ciphertext &encrypted, Plaintext &destination);
        // the same for every item of a device-resident batch, untrimmed, into caller-owned device memory:
        // [batch][K][N] words (CKKS) or [batch][N] words (BFV / BGV).  Stream-ordered on the null stream.
        size_t decrypt_batch_words(const Ciphertext &encrypted) const;
        void decrypt_batch(const Ciphertext &encrypted, uint64_t *device_out);

    private:
        void check(const Ciphertext &encrypted) const;
        void compute_secret_key_array(size_t max_power);                 // decryptor.cpp:243-316
        void dot_product_ct_sk_array(const Ciphertext &encrypted, uint64_t *phase, bool to_coeff_form); // decryptor.cpp:318-412
        const Context &context_;
        std::mutex mu_;
        std::vector<uint64_t *> powers_; // s^1, s^2, ... at the key level, NTT form
    };
} // namespace sealhip
