// The reference's wire format for the two objects that enter and leave the hot path — seal::Ciphertext and
// seal::KSwitchKeys (RelinKeys / GaloisKeys) — parsed straight into the word layout the device slabs use
// (SURVEY 8(f) N4).  Host side only; the C ABI (capi_core.cpp: Ciphertext_Load / _UnsafeLoad / _Save / _SaveSize,
// KSwitchKeys_Load / _UnsafeLoad) uploads the images.
//
// Format (all little-endian; every object is framed by a 16-byte SEALHeader whose `size` counts the header):
//   SEALHeader  { u16 magic = 0xA15E; u8 header_size = 16; u8 version_major, version_minor; u8 compr_mode; u16 reserved; u64 size }
//                                                   native/src/seal/serialization.h (struct SEALHeader)
//   Ciphertext  = SEALHeader, parms_id (4 x u64), is_ntt_form (u8), size, poly_modulus_degree, coeff_modulus_size (u64 each),
//                 scale (f64), correction_factor (u64), DynArray, [UniformRandomGeneratorInfo when seeded]
//                                                   native/src/seal/ciphertext.cpp:153-403
//   DynArray    = SEALHeader, count (u64), count x u64                       native/src/seal/dynarray.h:662-735
//   seeded      : the DynArray holds c_0 only (N*K words) and is followed by
//                 SEALHeader, prng_type (u8: 1 = blake2xb, 2 = shake256), seed (64 bytes); c_1 = sample_poly_uniform(prng(seed))
//                                                   native/src/seal/ciphertext.cpp:118-151, randomgen.cpp:87-110, util/rlwe.cpp
//   KSwitchKeys = SEALHeader, parms_id, keys_dim1 (u64), keys_dim1 x { keys_dim2 (u64), keys_dim2 x PublicKey }, a PublicKey
//                 being serialized as its Ciphertext              native/src/seal/kswitchkeys.cpp:47-180, publickey.h:106-131
// compr_mode: none (0), zlib (1: the system zlib, the reference's util/ztools.cpp:200-480) and zstd (2: libzstd.so.1 loaded
// at run time when present, ztools.cpp:560-900) — the reference's default builds write zstd.  Only the OUTERMOST object is
// compressed: its SEALHeader stays in the clear and the payload after it is one deflate / zstd stream of the member bytes
// (nested objects are always saved with compr_mode none: ciphertext.cpp:179-193, kswitchkeys.cpp:70-75).
// Exceptions are the reference's: std::invalid_argument / std::logic_error / std::runtime_error("I/O error") at the same
// conditions (Serialization::Load, serialization.cpp:341-553; Ciphertext::load_members; valcheck.cpp).
#pragma once
#include "context.h"
#include <cstdint>
#include <list>
#include <vector>

namespace sealhip
{
    namespace serial
    {
        constexpr uint16_t kMagic = 0xA15E;
        constexpr uint8_t kHeaderSize = 0x10;
        constexpr uint8_t kVersionMajor = 4, kVersionMinor = 4; // the reference release this format follows (4.4.x)

        // host image of one seal::Ciphertext: metadata + [size][K][N] words (seed already expanded)
        struct CiphertextImage
        {
            const Level *level = nullptr;
            bool is_ntt_form = false;
            uint64_t size = 0;
            double scale = 1.0;
            uint64_t correction_factor = 1;
            bool was_seeded = false;
            // The words [size][K][N] in two pieces so that a stream is never copied on the host: `stored` points INTO the
            // input buffer (unaligned; valid while the caller keeps that buffer) and covers everything for a full object,
            // c_0 for a seeded one, whose c_1 is expanded into `expanded`.
            const uint8_t *stored = nullptr;
            size_t stored_words = 0;
            std::vector<uint64_t> expanded;
            std::list<std::vector<uint8_t>> inflated; // decompressed payloads `stored` may point into (compressed streams)
            // device_expand loads (Blake2xb seed, whole PRNG buffers): c_1 is left to the device (xof.h) - `expanded` stays empty
            // and the caller expands `pending_seed` into the pending_words words after the stored ones
            size_t pending_words = 0;
            uint64_t pending_seed[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
            uint8_t pending_type = 1; // 1 = blake2xb, 2 = shake256
            size_t word_count() const { return stored_words + expanded.size() + pending_words; }
            void copy_words(uint64_t *dst) const; // gather both pieces into one host array
        };
        struct KSwitchKeysImage
        {
            // keys[index] = decomposition digits of key `index`, each a size-2 key-level ciphertext in NTT form; empty = no key
            std::vector<std::vector<CiphertextImage>> keys;
            std::list<std::vector<uint8_t>> inflated; // as CiphertextImage::inflated, for all the digits
        };

        // Ciphertext::unsafe_load (check_data = false) / Ciphertext::load (true: is_valid_for, valcheck.cpp).  Returns the
        // bytes consumed.  A BGV ciphertext stored in coefficient form is returned as stored (is_ntt_form false): the caller
        // transforms it on the device, as the end of Ciphertext::load_members does on the host.
        // device_expand: leave the expansion of a Blake2xb-seeded c_1 to the caller (CiphertextImage::pending_words) when the
        // device kernel can do it; other seeded objects are expanded here on the host as before.
        size_t load_ciphertext(const Context &ctx, const uint8_t *in, size_t size, bool check_data, CiphertextImage &out,
                               bool device_expand = false);
        // KSwitchKeys::unsafe_load / load
        size_t load_kswitchkeys(const Context &ctx, const uint8_t *in, size_t size, bool check_data, KSwitchKeysImage &out,
                                bool device_expand = false);

        // compression of a saved object: `raw` = the uncompressed stream (header + members) as the save_* functions write it;
        // returns the bytes written to out (header with compr_mode and the compressed size, then the compressed members).
        // compress_bound = the capacity that always suffices (Serialization::ComprSizeEstimate plays this role).
        bool compr_mode_supported(uint8_t compr_mode);
        size_t compress_bound(size_t raw_bytes, uint8_t compr_mode);
        size_t compress_stream(const uint8_t *raw, size_t raw_bytes, uint8_t compr_mode, uint8_t *out, size_t capacity);

        // Ciphertext::save_size(compr_mode_type::none) / Ciphertext::save for a full (unseeded) ciphertext
        size_t ciphertext_save_size(uint64_t size, uint64_t poly_modulus_degree, uint64_t coeff_modulus_size);
        size_t save_ciphertext(const uint64_t *parms_id, bool is_ntt_form, uint64_t size, uint64_t poly_modulus_degree,
                               uint64_t coeff_modulus_size, double scale, uint64_t correction_factor, const uint64_t *words,
                               uint8_t *out, size_t capacity, size_t *data_offset = nullptr);
        // (words == nullptr: everything but the coefficient words is written and *data_offset tells the caller where they go,
        //  so that a device slab can be copied straight into the stream)

        // seal::Plaintext (plaintext.cpp:save_members / load_members): SEALHeader, parms_id (zero = coefficient form), coeff_count (u64),
        // scale (f64), DynArray.  level == nullptr: coefficient form.  `stored` points into the input buffer (unaligned).
        struct PlaintextImage
        {
            const Level *level = nullptr;
            uint64_t coeff_count = 0;
            double scale = 1.0;
            const uint8_t *stored = nullptr;
            std::list<std::vector<uint8_t>> inflated;
        };
        // Plaintext::unsafe_load (check_data = false) / Plaintext::load (true: data level, every coefficient below its modulus)
        size_t load_plaintext(const Context &ctx, const uint8_t *in, size_t size, bool check_data, PlaintextImage &out);
        // the coefficient range part of is_data_valid_for(const Plaintext &) (valcheck.cpp:348-396)
        bool plaintext_in_range(const Context &ctx, const PlaintextImage &img);
        size_t plaintext_save_size(uint64_t coeff_count);
        // as save_ciphertext: words == nullptr leaves the coefficient words to the caller (*data_offset)
        size_t save_plaintext(const uint64_t *parms_id, uint64_t coeff_count, double scale, const uint64_t *words, uint8_t *out,
                              size_t capacity, size_t *data_offset = nullptr);

        // EncryptionParameters::load (encryptionparams.cpp:51-122): scheme (u8), poly_modulus_degree, coeff_modulus_size (u64 each), every
        // Modulus as its own SEALHeader + u64, then the plain modulus the same way.  Returns the bytes consumed; throws the reference's
        // classes (logic_error for an invalid scheme / degree / modulus count, runtime_error("I/O error") for a short stream).
        size_t load_encryption_parameters(const uint8_t *in, size_t size, uint8_t &scheme, uint64_t &poly_modulus_degree,
                                          std::vector<uint64_t> &coeff_modulus, uint64_t &plain_modulus);

        // The reference's buffered PRNG (UniformRandomGenerator, randomgen.cpp:179-227): type 1 = Blake2xbPRNG, 2 = Shake256PRNG
        struct Prng
        {
            uint8_t type = 1;
            uint64_t seed[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
            uint64_t counter = 0;
            uint8_t buf[4096];
            size_t head = 4096;
            bool parallel = true; // bulk draws may use several host threads (switched off where the caller already does)
            Prng() = default;
            Prng(uint8_t t, const uint64_t *s) : type(t)
            {
                for (int i = 0; i < 8; i++)
                    seed[i] = s[i];
            }
            void refill();
            void generate(size_t bytes, uint8_t *dst);
        };
        void sample_poly_uniform(Prng &prng, const uint64_t *primes, size_t K, size_t N, uint64_t *dst);
        void sample_poly_cbd(Prng &prng, const uint64_t *primes, size_t K, size_t N, uint64_t *dst);
        // sample_poly_ternary (util/rlwe.cpp:24-43).  The reference draws each coefficient with std::uniform_int_distribution
        // <uint64_t>(0, 2) over a 32-bit generator, whose algorithm belongs to the C++ library: this restates libstdc++'s
        // (GCC >= 11: Lemire's multiply-shift, bits/uniform_int_dist.h _S_nd) - value = (g * 3) >> 32 with g = 0 redrawn - which is
        // what the reference build in this image uses (checked byte for byte in tests/decrypt_cases.py).
        void sample_poly_ternary(Prng &prng, const uint64_t *primes, size_t K, size_t N, uint64_t *dst);
        // the same two distributions as N signed bytes (ternary: -1, 0, 1; cbd: -21 .. 21) - what the Encryptor uploads; the device
        // replicates them into the RNS components (decrypt_kernels.h: k_expand_small)
        void sample_small_ternary(Prng &prng, size_t N, int8_t *dst);
        void sample_small_cbd(Prng &prng, size_t N, int8_t *dst);
        // Serializable<Ciphertext>::save of a seeded ciphertext (ciphertext.cpp:171-196): members, DynArray with c_0 only, then
        // the UniformRandomGeneratorInfo (type, seed) c_1 is re-expanded from.  words == nullptr: as save_ciphertext.
        size_t seeded_ciphertext_save_size(uint64_t poly_modulus_degree, uint64_t coeff_modulus_size);
        size_t save_seeded_ciphertext(const uint64_t *parms_id, bool is_ntt_form, uint64_t poly_modulus_degree, uint64_t coeff_modulus_size,
                                      double scale, uint64_t correction_factor, const uint64_t *c0_words, uint8_t prng_type,
                                      const uint64_t *seed, uint8_t *out, size_t capacity, size_t *data_offset = nullptr);

        // sample_poly_uniform (util/rlwe.cpp) with the Blake2xb PRNG of randomgen.cpp seeded by `seed` (8 words):
        // K*N words, component r uniform in [0, primes[r])
        void expand_seed_blake2xb(const uint64_t *seed, const uint64_t *primes, size_t K, size_t N, uint64_t *destination);
    } // namespace serial
} // namespace sealhip
