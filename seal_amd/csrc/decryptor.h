// Device-resident counterparts of seal::SecretKey and seal::Decryptor (the caller on the far side of the hot
// path).  Same method names, argument checks and exception classes as the reference (native/src/seal/decryptor.h:56-186,
// decryptor.cpp); the arithmetic reuses the NTT engine and three element-wise kernels (decrypt_kernels.h).
#pragma once
#include "ckks_encoder.h"
#include "decrypt_kernels.h"
#include "evaluator.h"
#include "serial.h"
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
        uint64_t *allocate(); // the [L][N] device words, for a producer on the device (keygen.h)
        void get(uint64_t *host_words) const; // SecretKey::data() copied to the host

    private:
        const Context *ctx_;
        uint64_t *dev_ = nullptr;
    };

    // seal::PublicKey (publickey.h): a size-2 key-level ciphertext in NTT form, [2][L][N] words, resident in HBM
    class PublicKey
    {
    public:
        explicit PublicKey(const Context &ctx) : ctx_(&ctx) {}
        ~PublicKey();
        PublicKey(const PublicKey &) = delete;
        PublicKey &operator=(const PublicKey &) = delete;
        const Context &context() const { return *ctx_; }
        void set(const void *host_words, size_t word_count); // 2*L*N words: PublicKey::data().data()
        void set_parts(const void *stored, size_t stored_words, const uint64_t *expanded, size_t expanded_words);
        const uint64_t *data() const { return dev_; }
        uint64_t *allocate(); // the [2][L][N] device words, for a producer on the device (keygen.h)
        void get(uint64_t *host_words) const;

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
        // Decryptor::invariant_noise_budget (decryptor.cpp:188-241), BFV / BGV: bits of noise room left, per batch item
        int invariant_noise_budget(const Ciphertext &encrypted);
        std::vector<int> invariant_noise_budgets(const Ciphertext &encrypted);
        size_t decrypt_batch_words(const Ciphertext &encrypted) const;
        void decrypt_batch(const Ciphertext &encrypted, uint64_t *device_out);

    private:
        void check(const Ciphertext &encrypted) const;
        void compute_secret_key_array(size_t max_power);                 // decryptor.cpp:243-316
        void dot_product_ct_sk_array(const Ciphertext &encrypted, uint64_t *phase, bool to_coeff_form); // decryptor.cpp:318-412
        const Context &context_;
        std::mutex mu_;
        std::vector<uint64_t *> powers_; // s^1, s^2, ... at the key level, NTT form
        std::map<size_t, uint64_t *> crt_;  // per level: ckks_encoder.h build_crt_constants
    };
    // seal::BatchEncoder (native/src/seal/batchencoder.h, batchencoder.cpp): N integers modulo t <-> one plaintext polynomial, through
    // the negacyclic NTT modulo t (the plain modulus has its own tables in the context's prime pool) and the 2 x N/2 matrix index
    // map.  encode / decode move host vectors like the reference; the *_device forms work on `batch` vectors already in HBM
    // (e.g. the output of Decryptor::decrypt_batch) without leaving it.
    class BatchEncoder
    {
    public:
        explicit BatchEncoder(const Context &context);
        ~BatchEncoder();
        BatchEncoder(const BatchEncoder &) = delete;
        BatchEncoder &operator=(const BatchEncoder &) = delete;
        size_t slot_count() const { return context_.n(); }
        // values: count <= N unsigned values below t, or signed values of magnitude <= t/2 (is_signed)
        void encode(const uint64_t *values, size_t count, bool is_signed, Plaintext &destination) const;
        // N values out (signed: the balanced representatives as int64 bit patterns)
        void decode(const Plaintext &plain, uint64_t *values, bool is_signed) const;
        // device vectors [batch][N]; in and out may not alias
        void encode_device(const uint64_t *values, unsigned batch, bool is_signed, uint64_t *coefficients) const;
        void decode_device(const uint64_t *coefficients, unsigned batch, bool is_signed, uint64_t *values) const;

    private:
        const Context &context_;
        uint32_t *map_ = nullptr; // matrix_reps_index_map_, device
    };

    // seal::Encryptor, the secret-key half (native/src/seal/encryptor.h: encrypt_symmetric / encrypt_zero_symmetric and their
    // Serializable<> forms; encryptor.cpp:116-330, util/rlwe.cpp:270-395).  The randomness is the reference's: a bootstrap
    // Blake2xb PRNG yields the public seed of c_1 = a (expanded by sample_poly_uniform) and the centred-binomial noise e
    // (sample_poly_cbd); both are sampled on the device from the reference's byte streams (xof_kernels.h; the host keeps the
    // same samplers, serial.h, for the cases the kernels leave out) and c_0 = -(a s + e) [+ the plaintext] is computed there too.
    // Public-key encryption (encrypt / encrypt_zero; util::encrypt_zero_asymmetric, rlwe.cpp:196-268) follows the same pattern with
    // u <- ternary (serial.h: sample_poly_ternary, tied to libstdc++'s uniform_int_distribution), c_j = pk_j u + e_j at the level
    // above and one modulus switch down (encryptor.cpp:139-186).
    class Encryptor
    {
    public:
        // either key may be null (Encryptor(context, public_key) / (context, secret_key) / both)
        Encryptor(const Context &context, const PublicKey *public_key, const SecretKey *secret_key);
        Encryptor(const Context &context, const SecretKey &secret_key) : Encryptor(context, nullptr, &secret_key) {}
        const Context &context() const { return context_; }
        // Encryptor::encrypt_zero(parms_id, destination) / encrypt(plain, destination): public-key encryption, batch of one
        void encrypt_zero(const uint64_t *parms_id, Ciphertext &destination);
        void encrypt(const Plaintext &plain, Ciphertext &destination);
        ~Encryptor();
        Encryptor(const Encryptor &) = delete;
        Encryptor &operator=(const Encryptor &) = delete;

        // the reference's seeded factory (Blake2xbPRNGFactory(seed): every encryption restarts from this seed) for reproducible
        // runs and parity tests; without it every encryption draws a fresh 64-byte seed from the operating system
        void set_seed(const uint64_t *seed8);
        void clear_seed() { seeded_ = false; }

        // Encryptor::encrypt_zero_symmetric(parms_id, destination) / encrypt_symmetric(plain, destination): batch of one
        void encrypt_zero_symmetric(const uint64_t *parms_id, Ciphertext &destination);
        void encrypt_symmetric(const Plaintext &plain, Ciphertext &destination);
        // the Serializable<Ciphertext> forms, saved: a SEEDED stream (c_0 and the seed of c_1); returns the bytes written
        size_t symmetric_save_size(const uint64_t *parms_id) const;
        size_t encrypt_zero_symmetric_save(const uint64_t *parms_id, uint8_t *out, size_t capacity);
        size_t encrypt_symmetric_save(const Plaintext &plain, uint8_t *out, size_t capacity);

    private:
        friend class KeyGenerator; // keys are encryptions of zero under s (keygenerator.cpp:93-121, 322-357)
        const Level *level_for(const uint64_t *parms_id) const;
        const Level *level_for(const Plaintext &plain) const; // + the checks of Encryptor::encrypt_internal
        // key_form: NTT form whatever the scheme - how KeyGenerator calls encrypt_zero_symmetric (is_ntt_form = true)
        void zero(const Level &lvl, bool save_seed, Ciphertext &destination, uint64_t *public_seed, bool host_sampling = false,
                  bool key_form = false);
        void zero_asymmetric(const Level &lvl, Ciphertext &destination);
        void zero_asymmetric_at(const Level &lvl, Ciphertext &destination, bool host_sampling = false); // util::encrypt_zero_asymmetric
        void bootstrap_seed(uint64_t *seed8) const;
        uint64_t *pk_ = nullptr; // [2][L][N], NTT form
        void add_plain(const Plaintext &plain, Ciphertext &destination);
        size_t save(const Ciphertext &ct, const uint64_t *public_seed, uint8_t *out, size_t capacity) const;
        const Context &context_;
        Evaluator evaluator_;
        uint64_t *sk_ = nullptr; // [L][N], NTT form
        bool seeded_ = false;
        uint64_t seed_[8];
    };
} // namespace sealhip
