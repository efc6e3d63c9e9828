// See decryptor.h.  Reference: native/src/seal/decryptor.cpp.
#include "decryptor.h"
#include "xof.h"
#include "hostmath.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace sealhip
{
    namespace
    {
        void ck(hipError_t e, const char *what)
        {
            if (e != hipSuccess)
                throw std::runtime_error(std::string("HIP failure in ") + what + ": " + hipGetErrorString(e));
        }
        NttBatch polys(uint64_t *data, size_t K, size_t n, size_t count)
        {
            NttBatch b{};
            b.data = data;
            b.outer_stride = K * n;
            b.ncomp = (unsigned)K;
            b.nouter = (unsigned)count;
            b.prime_first = 0;
            return b;
        }
    } // namespace

    SecretKey::~SecretKey()
    {
        if (dev_)
            (void)hipFree(dev_);
    }
    uint64_t *SecretKey::allocate()
    {
        if (!dev_)
            ck(hipMalloc(reinterpret_cast<void **>(&dev_), ctx_->key_level().K * ctx_->n() * 8), "hipMalloc secret key");
        return dev_;
    }
    void SecretKey::get(uint64_t *host_words) const
    {
        if (!dev_ || !host_words)
            throw std::invalid_argument("secret key is not set");
        ck(hipMemcpy(host_words, dev_, ctx_->key_level().K * ctx_->n() * 8, hipMemcpyDeviceToHost), "download secret key");
    }
    uint64_t *PublicKey::allocate()
    {
        if (!dev_)
            ck(hipMalloc(reinterpret_cast<void **>(&dev_), 2 * ctx_->key_level().K * ctx_->n() * 8), "hipMalloc public key");
        return dev_;
    }
    void PublicKey::get(uint64_t *host_words) const
    {
        if (!dev_ || !host_words)
            throw std::invalid_argument("public key is not set");
        ck(hipMemcpy(host_words, dev_, 2 * ctx_->key_level().K * ctx_->n() * 8, hipMemcpyDeviceToHost), "download public key");
    }
    void SecretKey::set(const void *host_words, size_t word_count)
    {
        const size_t want = ctx_->key_level().K * ctx_->n();
        if (!host_words || word_count != want)
            throw std::invalid_argument("secret_key is not valid for encryption parameters");
        if (!dev_)
            ck(hipMalloc(reinterpret_cast<void **>(&dev_), want * 8), "hipMalloc secret key");
        ck(hipMemcpy(dev_, host_words, want * 8, hipMemcpyHostToDevice), "upload secret key");
    }

    PublicKey::~PublicKey()
    {
        if (dev_)
            (void)hipFree(dev_);
    }
    void PublicKey::set_parts(const void *stored, size_t stored_words, const uint64_t *expanded, size_t expanded_words)
    {
        const size_t want = 2 * ctx_->key_level().K * ctx_->n();
        if (stored_words + expanded_words != want || (stored_words && !stored))
            throw std::invalid_argument("public key is not valid for encryption parameters");
        if (!dev_)
            ck(hipMalloc(reinterpret_cast<void **>(&dev_), want * 8), "hipMalloc public key");
        if (stored_words)
            ck(hipMemcpy(dev_, stored, stored_words * 8, hipMemcpyHostToDevice), "upload public key");
        if (expanded_words)
            ck(hipMemcpy(dev_ + stored_words, expanded, expanded_words * 8, hipMemcpyHostToDevice), "upload public key");
    }
    void PublicKey::set(const void *host_words, size_t word_count)
    {
        set_parts(host_words, word_count, nullptr, 0);
    }

    Decryptor::Decryptor(const Context &context, const SecretKey &secret_key) : context_(context)
    {
        if (&secret_key.context() != &context || !secret_key.data())
            throw std::invalid_argument("secret key is not valid for encryption parameters");
        const size_t words = context.key_level().K * context.n();
        uint64_t *p = nullptr;
        ck(hipMalloc(reinterpret_cast<void **>(&p), words * 8), "hipMalloc secret key array");
        powers_.push_back(p);
        ck(hipMemcpy(p, secret_key.data(), words * 8, hipMemcpyDeviceToDevice), "copy secret key");
    }
    Decryptor::~Decryptor()
    {
        for (auto &kv : crt_)
            (void)hipFree(kv.second);
        for (uint64_t *p : powers_)
        {
            // the reference wipes key material before releasing it (decryptor.cpp: seal_memzero)
            (void)hipMemset(p, 0, context_.key_level().K * context_.n() * 8);
            (void)hipFree(p);
        }
    }

    void Decryptor::compute_secret_key_array(size_t max_power)
    {
        std::lock_guard<std::mutex> lock(mu_);
        const size_t L = context_.key_level().K, words = L * context_.n();
        while (powers_.size() < max_power)
        {
            // s^(k+1) = s^k (.) s in NTT form at the key level
            uint64_t *p = nullptr;
            ck(hipMalloc(reinterpret_cast<void **>(&p), words * 8), "hipMalloc secret key power");
            powers_.push_back(p);
            ck(k_dyadic(context_.dev_mods(), powers_[powers_.size() - 2], powers_[0], p, (unsigned)context_.log_n(), (unsigned)L, 0, 1, nullptr),
               "secret key power");
        }
    }

    void Decryptor::check(const Ciphertext &e) const
    {
        // is_valid_for metadata (valcheck.cpp; the coefficient range is not re-read from HBM) + decryptor.cpp:82-92
        if (&e.context() != &context_ || !e.level() || e.level()->chain_index > context_.first_level().chain_index)
            throw std::invalid_argument("encrypted is not valid for encryption parameters");
        if (e.size() < 2)
            throw std::invalid_argument("encrypted is empty");
        if (e.size() > 6)
            throw std::invalid_argument("encrypted is not valid for encryption parameters");
        const Scheme s = context_.scheme();
        if (s == Scheme::bfv && e.is_ntt_form())
            throw std::invalid_argument("encrypted cannot be in NTT form");
        if (s != Scheme::bfv && !e.is_ntt_form())
            throw std::invalid_argument("encrypted must be in NTT form");
    }

    void Decryptor::dot_product_ct_sk_array(const Ciphertext &e, uint64_t *phase, bool to_coeff_form)
    {
        const size_t n = context_.n(), K = e.level()->K, B = e.batch();
        const unsigned n_log = (unsigned)context_.log_n();
        const size_t plane_words = e.plane_words();
        compute_secret_key_array(e.size() - 1);
        SkPowers sk{};
        {
            std::lock_guard<std::mutex> lock(mu_);
            for (size_t p = 0; p + 1 < e.size(); p++)
                sk.p[p] = powers_[p];
        }
        const NttTables &tb = context_.ntt_tables();
        if (e.is_ntt_form())
        {
            ck(k_decrypt_dot(context_.dev_mods(), e.plane(0), e.plane(1), plane_words, (unsigned)e.size(), sk, phase, n_log, (unsigned)K, nullptr),
               "decrypt dot product");
            if (to_coeff_form)
                ck(ntt_inverse(tb, polys(phase, K, n, B), 0, nullptr), "decrypt intt");
        }
        else
        {
            // coefficient-form input (BFV): c_1.. are transformed, multiplied, summed, transformed back, then c_0 is added
            Scratch tmp((e.size() - 1) * plane_words);
            ck(hipMemcpyAsync(tmp.p, e.plane(1), (e.size() - 1) * plane_words * 8, hipMemcpyDeviceToDevice, nullptr), "decrypt copy");
            ck(ntt_forward(tb, polys(tmp.p, K, n, (e.size() - 1) * B), 0, nullptr), "decrypt ntt");
            ck(k_decrypt_dot(context_.dev_mods(), nullptr, tmp.p, plane_words, (unsigned)e.size(), sk, phase, n_log, (unsigned)K, nullptr),
               "decrypt dot product");
            ck(ntt_inverse(tb, polys(phase, K, n, B), 0, nullptr), "decrypt intt");
            ck(k_add_inplace(context_.dev_mods(), phase, e.plane(0), plane_words, n_log, (unsigned)K, nullptr), "decrypt add c0");
            ck(hipStreamSynchronize(nullptr), "decrypt sync"); // tmp goes back to the pool
        }
    }

    std::vector<int> Decryptor::invariant_noise_budgets(const Ciphertext &e)
    {
        check(e);
        const Scheme s = context_.scheme();
        if (s != Scheme::bfv && s != Scheme::bgv)
            throw std::logic_error("unsupported scheme");
        const Level &lvl = *e.level();
        const unsigned n_log = (unsigned)context_.log_n(), K = lvl.K, B = (unsigned)e.batch();
        uint64_t *crt;
        {
            std::lock_guard<std::mutex> lock(mu_);
            auto it = crt_.find(lvl.chain_index);
            if (it == crt_.end())
                it = crt_.emplace(lvl.chain_index, build_crt_constants(context_, lvl)).first;
            crt = it->second;
        }
        Scratch noise(e.plane_words()), bits((B + 1) / 2 + 1);
        dot_product_ct_sk_array(e, noise.p, true); // coefficient form (BGV: INTT of the NTT-form phase)
        unsigned *d_bits = reinterpret_cast<unsigned *>(bits.p);
        ck(hipMemsetAsync(d_bits, 0, B * sizeof(unsigned), nullptr), "zero norms");
        // BFV multiplies the phase by t first (the invariant noise is t * phase / Q); BGV takes the phase as it is
        ck(k_crt_norm_bits(context_.dev_mods(), noise.p, crt, reinterpret_cast<const ShoupOp *>(crt + (size_t)K * K + 2 * K), crt + (size_t)K * K,
                           crt + (size_t)K * K + K, s == Scheme::bfv ? context_.plain_modulus() : 1, d_bits, n_log, K, B, nullptr),
           "noise norm");
        std::vector<unsigned> host(B);
        ck(hipMemcpy(host.data(), d_bits, B * sizeof(unsigned), hipMemcpyDeviceToHost), "download norms");
        std::vector<int> out(B);
        for (unsigned b = 0; b < B; b++)
            out[b] = std::max(0, lvl.total_coeff_modulus_bit_count - (int)host[b] - 1);
        return out;
    }
    int Decryptor::invariant_noise_budget(const Ciphertext &e)
    {
        if (e.batch() != 1)
            throw std::invalid_argument("Decryptor::invariant_noise_budget takes a batch of one: use invariant_noise_budgets");
        return invariant_noise_budgets(e)[0];
    }

    size_t Decryptor::decrypt_batch_words(const Ciphertext &e) const
    {
        check(e);
        return context_.scheme() == Scheme::ckks ? e.plane_words() : e.batch() * context_.n();
    }

    void Decryptor::decrypt_batch(const Ciphertext &e, uint64_t *out)
    {
        check(e);
        if (!out)
            throw std::invalid_argument("destination");
        const Scheme s = context_.scheme();
        const unsigned n_log = (unsigned)context_.log_n();
        if (s == Scheme::ckks)
        {
            dot_product_ct_sk_array(e, out, false); // decryptor.cpp:153-186
            return;
        }
        Scratch phase(e.plane_words());
        dot_product_ct_sk_array(e, phase.p, true);
        const ModDesc t = host::make_mod(context_.plain_modulus());
        if (s == Scheme::bfv) // decryptor.cpp:115-151
            ck(k_decrypt_scale_and_round(context_.dev_mods(), e.level()->dev, t, phase.p, out, n_log, (unsigned)e.batch(), nullptr),
               "decrypt_scale_and_round");
        else
        {
            // decryptor.cpp:188-233
            uint64_t fix = 1;
            if (e.correction_factor() != 1)
            {
                try
                {
                    fix = host::invmod(e.correction_factor() % context_.plain_modulus(), context_.plain_modulus());
                }
                catch (const std::invalid_argument &)
                {
                    throw std::logic_error("invalid correction factor");
                }
            }
            ck(k_decrypt_modt(context_.dev_mods(), e.level()->dev, t, fix, phase.p, out, n_log, (unsigned)e.batch(), nullptr), "decrypt_modt");
        }
        ck(hipStreamSynchronize(nullptr), "decrypt sync"); // phase goes back to the pool
    }

    void Decryptor::decrypt(const Ciphertext &e, Plaintext &destination)
    {
        check(e);
        if (e.batch() != 1)
            throw std::invalid_argument("Decryptor::decrypt takes a batch of one: use decrypt_batch");
        if (&destination.context() != &context_)
            throw std::invalid_argument("destination belongs to another context");
        const size_t words = decrypt_batch_words(e);
        uint64_t *slab = DevicePool::global().alloc_words(words);
        try
        {
            decrypt_batch(e, slab);
            ck(hipStreamSynchronize(nullptr), "decrypt sync");
        }
        catch (...)
        {
            DevicePool::global().free_words(slab);
            throw;
        }
        if (context_.scheme() == Scheme::ckks)
        {
            destination.adopt(slab, words, words);
            destination.set_level(e.level());
            destination.scale() = e.scale();
            return;
        }
        // trim to the significant coefficients (get_significant_uint64_count_uint), at least one
        std::vector<uint64_t> host(words);
        hipError_t err = hipMemcpy(host.data(), slab, words * 8, hipMemcpyDeviceToHost);
        if (err != hipSuccess)
        {
            DevicePool::global().free_words(slab);
            ck(err, "decrypt read-back");
        }
        size_t count = words;
        while (count > 1 && host[count - 1] == 0)
            count--;
        destination.adopt(slab, count, words);
        destination.set_level(nullptr);
    }
    // ---------------------------------------------------------------- BatchEncoder
    BatchEncoder::BatchEncoder(const Context &context) : context_(context)
    {
        // batchencoder.cpp:17-48
        if (context.scheme() != Scheme::bfv && context.scheme() != Scheme::bgv)
            throw std::invalid_argument("unsupported scheme");
        if (context.plain_prime_index() < 0)
            throw std::invalid_argument("encryption parameters are not valid for batching");
        // populate_matrix_reps_index_map (batchencoder.cpp:97-123)
        const size_t n = context.n(), row = n >> 1, m = n << 1;
        const int logn = context.log_n();
        std::vector<uint32_t> map(n);
        uint64_t pos = 1;
        auto rev = [&](uint64_t v) {
            uint64_t r = 0;
            for (int b = 0; b < logn; b++)
                r |= ((v >> b) & 1) << (logn - 1 - b);
            return (uint32_t)r;
        };
        for (size_t i = 0; i < row; i++)
        {
            map[i] = rev((pos - 1) >> 1);
            map[row | i] = rev((m - pos - 1) >> 1);
            pos = (pos * 3) & (m - 1);
        }
        ck(hipMalloc(reinterpret_cast<void **>(&map_), n * 4), "hipMalloc index map");
        ck(hipMemcpy(map_, map.data(), n * 4, hipMemcpyHostToDevice), "upload index map");
    }
    BatchEncoder::~BatchEncoder()
    {
        if (map_)
            (void)hipFree(map_);
    }
    void BatchEncoder::encode_device(const uint64_t *values, unsigned batch, bool is_signed, uint64_t *coefficients) const
    {
        if (!values || !coefficients || values == coefficients)
            throw std::invalid_argument("values / coefficients");
        const unsigned n_log = (unsigned)context_.log_n();
        ck(k_slot_scatter(map_, values, coefficients, n_log, batch, is_signed ? context_.plain_modulus() : 0, nullptr), "slot scatter");
        NttBatch b{};
        b.data = coefficients;
        b.outer_stride = context_.n();
        b.ncomp = 1;
        b.nouter = batch;
        b.prime_first = (unsigned)context_.plain_prime_index();
        ck(ntt_inverse(context_.ntt_tables(), b, 0, nullptr), "intt mod t");
    }
    void BatchEncoder::decode_device(const uint64_t *coefficients, unsigned batch, bool is_signed, uint64_t *values) const
    {
        if (!values || !coefficients || values == coefficients)
            throw std::invalid_argument("values / coefficients");
        const size_t words = (size_t)batch * context_.n();
        Scratch tmp(words);
        ck(hipMemcpyAsync(tmp.p, coefficients, words * 8, hipMemcpyDeviceToDevice, nullptr), "copy coefficients");
        NttBatch b{};
        b.data = tmp.p;
        b.outer_stride = context_.n();
        b.ncomp = 1;
        b.nouter = batch;
        b.prime_first = (unsigned)context_.plain_prime_index();
        ck(ntt_forward(context_.ntt_tables(), b, 0, nullptr), "ntt mod t");
        ck(k_slot_gather(map_, tmp.p, values, (unsigned)context_.log_n(), batch, is_signed ? context_.plain_modulus() : 0, nullptr), "slot gather");
        ck(hipStreamSynchronize(nullptr), "decode sync"); // tmp goes back to the pool
    }
    void BatchEncoder::encode(const uint64_t *values, size_t count, bool is_signed, Plaintext &destination) const
    {
        // batchencoder.cpp:125-165 (unsigned) / 167-215 (signed)
        const size_t n = context_.n();
        const uint64_t t = context_.plain_modulus();
        if (&destination.context() != &context_)
            throw std::invalid_argument("destination belongs to another context");
        if (count > n)
            throw std::invalid_argument("values_matrix size is too large");
        if (count && !values)
            throw std::invalid_argument("values_matrix");
        std::vector<uint64_t> padded(n, 0);
        for (size_t i = 0; i < count; i++)
        {
            const uint64_t v = values[i];
            if (is_signed)
            {
                const int64_t sv = (int64_t)v;
                const uint64_t mag = sv < 0 ? (uint64_t)0 - v : v;
                if (mag > (t >> 1))
                    throw std::invalid_argument("input value is larger than plain_modulus");
            }
            else if (v >= t)
                throw std::invalid_argument("input value is larger than plain_modulus");
            padded[i] = v;
        }
        Scratch in(n);
        uint64_t *slab = DevicePool::global().alloc_words(n);
        try
        {
            ck(hipStreamSynchronize(nullptr), "encode sync");
            ck(hipMemcpy(in.p, padded.data(), n * 8, hipMemcpyHostToDevice), "upload values");
            encode_device(in.p, 1, is_signed, slab);
            ck(hipStreamSynchronize(nullptr), "encode sync");
        }
        catch (...)
        {
            DevicePool::global().free_words(slab);
            throw;
        }
        destination.adopt(slab, n, n);
        destination.set_level(nullptr);
    }
    void BatchEncoder::decode(const Plaintext &plain, uint64_t *values, bool is_signed) const
    {
        // batchencoder.cpp:357-397 / 399-447
        const size_t n = context_.n();
        if (&plain.context() != &context_ || plain.coeff_count() > n)
            throw std::invalid_argument("plain is not valid for encryption parameters");
        if (plain.is_ntt_form())
            throw std::invalid_argument("plain cannot be in NTT form");
        if (!values)
            throw std::invalid_argument("destination");
        Scratch in(n), out(n);
        ck(hipStreamSynchronize(nullptr), "decode sync");
        ck(hipMemsetAsync(in.p, 0, n * 8, nullptr), "zero pad");
        if (plain.coeff_count())
            ck(hipMemcpyAsync(in.p, plain.data(), plain.coeff_count() * 8, hipMemcpyDeviceToDevice, nullptr), "copy plain");
        decode_device(in.p, 1, is_signed, out.p);
        ck(hipMemcpy(values, out.p, n * 8, hipMemcpyDeviceToHost), "download values");
    }

    // ---------------------------------------------------------------- Encryptor (secret-key encryption)
    Encryptor::Encryptor(const Context &context, const PublicKey *public_key, const SecretKey *secret_key)
        : context_(context), evaluator_(context)
    {
        const size_t words = context.key_level().K * context.n();
        if (secret_key)
        {
            if (&secret_key->context() != &context || !secret_key->data())
                throw std::invalid_argument("secret key is not valid for encryption parameters");
            ck(hipMalloc(reinterpret_cast<void **>(&sk_), words * 8), "hipMalloc secret key");
            ck(hipMemcpy(sk_, secret_key->data(), words * 8, hipMemcpyDeviceToDevice), "copy secret key");
        }
        if (public_key)
        {
            if (&public_key->context() != &context || !public_key->data())
                throw std::invalid_argument("public key is not valid for encryption parameters");
            ck(hipMalloc(reinterpret_cast<void **>(&pk_), 2 * words * 8), "hipMalloc public key");
            ck(hipMemcpy(pk_, public_key->data(), 2 * words * 8, hipMemcpyDeviceToDevice), "copy public key");
        }
    }
    void Encryptor::bootstrap_seed(uint64_t *seed8) const
    {
        if (seeded_)
            std::memcpy(seed8, seed_, 64);
        else
            host::random_bytes(seed8, 64);
    }

    // util::encrypt_zero_asymmetric (util/rlwe.cpp:196-268) at `lvl`
    void Encryptor::zero_asymmetric_at(const Level &lvl, Ciphertext &d, bool host_sampling)
    {
        const size_t n = context_.n(), K = lvl.K, L = context_.key_level().K, words = K * n;
        const unsigned n_log = (unsigned)context_.log_n();
        const Scheme scheme = context_.scheme();
        const bool ntt_form = scheme != Scheme::bfv;
        // u <- R_3, then e_0, e_1 <- chi, all from one PRNG, in this order: bytes [0, 4n), [4n, 10n), [10n, 16n) of its stream, which
        // the device produces itself (xof_kernels.h) - unless a ternary draw has to be redrawn (probability n / 2^32), which shifts
        // everything after it: then the sampling is repeated here on the host, as it is for rings too small for whole 64-byte pieces
        uint64_t boot[8];
        bootstrap_seed(boot);
        if (n < 4 || std::getenv("SEALHIP_ENCRYPT_HOST_SAMPLING")) // (the switch: tests run the host branch, otherwise one call in 2^16)
            host_sampling = true;

        ck(hipStreamSynchronize(nullptr), "encrypt sync");
        d.resize(&lvl, 2, nullptr);
        d.is_ntt_form() = ntt_form;
        d.scale() = 1.0;
        d.correction_factor() = 1;
        const NttTables &tb = context_.ntt_tables();
        const ModDesc *mods = context_.dev_mods();
        const size_t small_words = (3 * n + 7) / 8;
        Scratch du(words), de(2 * words), ds(small_words + 1), stream(host_sampling ? 1 : 2 * n);
        int8_t *dsmall = reinterpret_cast<int8_t *>(ds.p);
        unsigned *redraw = reinterpret_cast<unsigned *>(ds.p + small_words);
        if (host_sampling)
        {
            // (N signed bytes each; the device replicates them into the RNS components)
            serial::Prng prng(1, boot);
            std::vector<int8_t> small(3 * n);
            serial::sample_small_ternary(prng, n, small.data());
            serial::sample_small_cbd(prng, n, small.data() + n);
            serial::sample_small_cbd(prng, n, small.data() + 2 * n);
            ck(hipMemcpy(ds.p, small.data(), 3 * n, hipMemcpyHostToDevice), "upload u, e");
        }
        else
        {
            XofSeed seed;
            std::memcpy(seed.w, boot, sizeof(seed.w));
            ck(hipMemsetAsync(redraw, 0, 8, nullptr), "clear flag");
            ck(k_blake2xb_stream(seed, 0, 16 * n / 64, stream.p, nullptr), "bootstrap stream");
            ck(k_small_from_stream(reinterpret_cast<const uint8_t *>(stream.p), n, 4 * n, 2 * n, dsmall, redraw, nullptr), "sample u, e");
        }
        ck(k_expand_small(mods, dsmall, du.p, n_log, (unsigned)K, 1, nullptr), "expand u");
        ck(k_expand_small(mods, dsmall + n, de.p, n_log, (unsigned)K, 2, nullptr), "expand e");
        ck(ntt_forward(tb, polys(du.p, K, n, 1), 0, nullptr), "ntt u");
        for (size_t j = 0; j < 2; j++)
            ck(k_dyadic(mods, du.p, pk_ + j * L * n, d.plane(j), n_log, (unsigned)K, 0, 1, nullptr), "pk u");
        if (ntt_form)
            ck(ntt_forward(tb, polys(de.p, K, n, 2), 0, nullptr), "ntt noise");
        else
            ck(ntt_inverse(tb, polys(d.data(), K, n, 2), 0, nullptr), "intt pk u");
        ck(k_neg_add_noise(mods, d.data(), de.p, scheme == Scheme::bgv ? context_.plain_modulus() : 1, 2 * words, n_log, (unsigned)K, nullptr,
                           false),
           "c + e");
        unsigned flag = 0;
        if (!host_sampling)
            ck(hipMemcpy(&flag, redraw, sizeof(flag), hipMemcpyDeviceToHost), "read flag");
        ck(hipStreamSynchronize(nullptr), "encrypt sync");
        if (flag)
            zero_asymmetric_at(lvl, d, true);
    }
    // Encryptor::encrypt_zero_internal, asymmetric branch (encryptor.cpp:139-186): encrypt one level up, switch down
    void Encryptor::zero_asymmetric(const Level &lvl, Ciphertext &d)
    {
        if (!pk_)
            throw std::logic_error("public key is not set");
        if (&d.context() != &context_)
            throw std::invalid_argument("destination belongs to another context");
        if (d.batch() != 1)
            throw std::invalid_argument("Encryptor encrypts one ciphertext at a time: destination must be a batch of one");
        const Level *prev = context_.level_by_chain_index(lvl.chain_index + 1);
        if (!prev)
        {
            zero_asymmetric_at(lvl, d);
            return;
        }
        zero_asymmetric_at(*prev, d);
        evaluator_.mod_switch_scale_to_next(d);
        evaluator_.synchronize();
        d.scale() = 1.0;             // destination.scale() = temp.scale(), .correction_factor() = temp.correction_factor()
        d.correction_factor() = 1;
    }
    void Encryptor::encrypt_zero(const uint64_t *parms_id, Ciphertext &destination)
    {
        zero_asymmetric(*level_for(parms_id), destination);
    }
    void Encryptor::encrypt(const Plaintext &plain, Ciphertext &destination)
    {
        if (!pk_)
            throw std::logic_error("public key is not set");
        zero_asymmetric(*level_for(plain), destination);
        add_plain(plain, destination);
    }

    Encryptor::~Encryptor()
    {
        if (pk_)
            (void)hipFree(pk_);
        if (sk_)
        {
            (void)hipMemset(sk_, 0, context_.key_level().K * context_.n() * 8);
            (void)hipFree(sk_);
        }
    }
    void Encryptor::set_seed(const uint64_t *seed8)
    {
        if (!seed8)
            throw std::invalid_argument("seed");
        std::memcpy(seed_, seed8, sizeof(seed_));
        seeded_ = true;
    }

    const Level *Encryptor::level_for(const uint64_t *parms_id) const
    {
        const Level *l = parms_id ? context_.level_by_parms_id(parms_id) : nullptr;
        if (!l)
            throw std::invalid_argument("parms_id is not valid for encryption parameters"); // encryptor.cpp:130-134
        return l;
    }
    const Level *Encryptor::level_for(const Plaintext &plain) const
    {
        // Encryptor::encrypt_internal (encryptor.cpp:213-330): where each scheme encrypts and what it accepts
        if (&plain.context() != &context_)
            throw std::invalid_argument("plain is not valid for encryption parameters");
        const Scheme s = context_.scheme();
        if (s == Scheme::ckks)
        {
            if (!plain.is_ntt_form())
                throw std::invalid_argument("plain must be in NTT form");
            if (plain.level()->chain_index > context_.first_level().chain_index ||
                plain.coeff_count() != plain.level()->K * context_.n())
                throw std::invalid_argument("plain is not valid for encryption parameters");
            return plain.level();
        }
        if (plain.is_ntt_form())
            throw std::invalid_argument("plain cannot be in NTT form");
        if (plain.coeff_count() > context_.n())
            throw std::invalid_argument("plain is not valid for encryption parameters");
        return &context_.first_level();
    }

    // util::encrypt_zero_symmetric (util/rlwe.cpp:270-395)
    void Encryptor::zero(const Level &lvl, bool save_seed, Ciphertext &d, uint64_t *public_seed, bool host_sampling, bool key_form)
    {
        if (!sk_)
            throw std::logic_error("secret key is not set");
        if (&d.context() != &context_)
            throw std::invalid_argument("destination belongs to another context");
        if (d.batch() != 1)
            throw std::invalid_argument("Encryptor encrypts one ciphertext at a time: destination must be a batch of one");
        const size_t n = context_.n(), K = lvl.K, words = K * n;
        const unsigned n_log = (unsigned)context_.log_n();
        const Scheme scheme = context_.scheme();
        const bool ntt_form = key_form || scheme != Scheme::bfv;
        // a polynomial too small to hold the seed is saved in full (rlwe.cpp:298-306): 16 + 1 + 64 bytes -> 11 words, plus a marker
        if (save_seed && words < 12)
            save_seed = false;

        // host: the reference's randomness, in the reference's order
        uint64_t boot_seed[8];
        bootstrap_seed(boot_seed);
        serial::Prng bootstrap(1, boot_seed);
        uint64_t pub[8];
        bootstrap.generate(sizeof(pub), reinterpret_cast<uint8_t *>(pub));
        serial::Prng cprng(1, pub);
        // a = sample_poly_uniform(cprng): on the device when the stream is whole PRNG buffers (xof.h), else here
        const bool device_a = xof_device_ok(1, K, n);
        std::vector<uint64_t> a(device_a ? 0 : words);
        // the noise: bytes [64, 64 + 6n) of the bootstrap stream, sampled on the device when they are whole 64-byte pieces
        const bool device_e = !host_sampling && (6 * n) % 64 == 0 && !std::getenv("SEALHIP_ENCRYPT_HOST_SAMPLING");
        std::vector<int8_t> noise(device_e ? 0 : n);
        if (!device_a)
            serial::sample_poly_uniform(cprng, context_.coeff_modulus().data(), K, n, a.data());
        if (!device_e)
            serial::sample_small_cbd(bootstrap, n, noise.data());
        if (public_seed)
            std::memcpy(public_seed, pub, sizeof(pub));

        // device: c1 = a, c0 = -(a s + e) (BGV: e -> t e)
        ck(hipStreamSynchronize(nullptr), "encrypt sync");
        d.resize(&lvl, 2, nullptr);
        d.is_ntt_form() = ntt_form;
        d.scale() = 1.0;
        d.correction_factor() = 1;
        uint64_t *c0 = d.plane(0), *c1 = d.plane(1);
        const NttTables &tb = context_.ntt_tables();
        const ModDesc *mods = context_.dev_mods();
        Scratch e(words), ds((n + 7) / 8 + 1), stream(device_e ? 6 * n / 8 : 1);
        if (device_a)
        {
            XofJob job;
            std::memcpy(job.seed, pub, sizeof(job.seed));
            job.dst = c1;
            sample_uniform_device(context_, K, { job });
        }
        else
            ck(hipMemcpy(c1, a.data(), words * 8, hipMemcpyHostToDevice), "upload a");
        if (device_e)
        {
            XofSeed seed;
            std::memcpy(seed.w, boot_seed, sizeof(seed.w));
            ck(k_blake2xb_stream(seed, 1, 6 * n / 64, stream.p, nullptr), "bootstrap stream");
            ck(k_small_from_stream(reinterpret_cast<const uint8_t *>(stream.p), 0, 0, n, reinterpret_cast<int8_t *>(ds.p),
                                   reinterpret_cast<unsigned *>(ds.p + (n + 7) / 8), nullptr),
               "sample noise");
        }
        else
            ck(hipMemcpy(ds.p, noise.data(), n, hipMemcpyHostToDevice), "upload noise");
        ck(k_expand_small(mods, reinterpret_cast<const int8_t *>(ds.p), e.p, n_log, (unsigned)K, 1, nullptr), "expand noise");
        if (ntt_form)
        {
            ck(k_dyadic(mods, sk_, c1, c0, n_log, (unsigned)K, 0, 1, nullptr), "a s");
            ck(ntt_forward(tb, polys(e.p, K, n, 1), 0, nullptr), "ntt noise");
            ck(k_neg_add_noise(mods, c0, e.p, scheme == Scheme::bgv ? context_.plain_modulus() : 1, words, n_log, (unsigned)K, nullptr), "c0");
        }
        else if (save_seed)
        {
            // a was sampled in coefficient form (it is what the seed re-expands to): transform a copy for the product
            Scratch an(words);
            ck(hipMemcpyAsync(an.p, c1, words * 8, hipMemcpyDeviceToDevice, nullptr), "copy a");
            ck(ntt_forward(tb, polys(an.p, K, n, 1), 0, nullptr), "ntt a");
            ck(k_dyadic(mods, sk_, an.p, c0, n_log, (unsigned)K, 0, 1, nullptr), "a s");
            ck(ntt_inverse(tb, polys(c0, K, n, 1), 0, nullptr), "intt a s");
            ck(k_neg_add_noise(mods, c0, e.p, 1, words, n_log, (unsigned)K, nullptr), "c0");
            ck(hipStreamSynchronize(nullptr), "encrypt sync");
        }
        else
        {
            // a was sampled in NTT form; the ciphertext is returned in coefficient form
            ck(k_dyadic(mods, sk_, c1, c0, n_log, (unsigned)K, 0, 1, nullptr), "a s");
            ck(ntt_inverse(tb, polys(c0, K, n, 1), 0, nullptr), "intt a s");
            ck(k_neg_add_noise(mods, c0, e.p, 1, words, n_log, (unsigned)K, nullptr), "c0");
            ck(ntt_inverse(tb, polys(c1, K, n, 1), 0, nullptr), "intt a");
        }
        ck(hipStreamSynchronize(nullptr), "encrypt sync"); // e goes back to the pool
    }

    void Encryptor::add_plain(const Plaintext &plain, Ciphertext &d)
    {
        // the three branches of Encryptor::encrypt_internal add the plaintext to c_0 exactly as Evaluator::add_plain does on a
        // fresh ciphertext (BFV: multiply_add_plain_with_scaling_variant; CKKS: add_poly_coeffmod, scale taken from the
        // plaintext; BGV: lift, transform, add - the correction factor of a fresh ciphertext is 1)
        if (context_.scheme() == Scheme::ckks)
            d.scale() = plain.scale();
        evaluator_.add_plain_inplace(d, plain);
        evaluator_.synchronize();
    }

    void Encryptor::encrypt_zero_symmetric(const uint64_t *parms_id, Ciphertext &destination)
    {
        zero(*level_for(parms_id), false, destination, nullptr);
    }
    void Encryptor::encrypt_symmetric(const Plaintext &plain, Ciphertext &destination)
    {
        zero(*level_for(plain), false, destination, nullptr);
        add_plain(plain, destination);
    }

    size_t Encryptor::symmetric_save_size(const uint64_t *parms_id) const
    {
        const Level *l = level_for(parms_id);
        const size_t n = context_.n(), K = l->K;
        return K * n < 12 ? serial::ciphertext_save_size(2, n, K) : serial::seeded_ciphertext_save_size(n, K);
    }
    size_t Encryptor::save(const Ciphertext &ct, const uint64_t *public_seed, uint8_t *out, size_t capacity) const
    {
        const size_t n = context_.n(), K = ct.level()->K, words = K * n;
        size_t off = 0, total;
        const bool seeded = words >= 12;
        if (seeded)
            total = serial::save_seeded_ciphertext(ct.level()->parms_id, ct.is_ntt_form(), n, K, ct.scale(), ct.correction_factor(), nullptr,
                                                   1, public_seed, out, capacity, &off);
        else
            total = serial::save_ciphertext(ct.level()->parms_id, ct.is_ntt_form(), 2, n, K, ct.scale(), ct.correction_factor(), nullptr, out,
                                            capacity, &off);
        ck(hipDeviceSynchronize(), "encrypt sync");
        ck(hipMemcpy(out + off, ct.data(), (seeded ? 1 : 2) * words * 8, hipMemcpyDeviceToHost), "download ciphertext");
        return total;
    }
    size_t Encryptor::encrypt_zero_symmetric_save(const uint64_t *parms_id, uint8_t *out, size_t capacity)
    {
        Ciphertext ct(context_, 1);
        uint64_t pub[8];
        zero(*level_for(parms_id), true, ct, pub);
        return save(ct, pub, out, capacity);
    }
    size_t Encryptor::encrypt_symmetric_save(const Plaintext &plain, uint8_t *out, size_t capacity)
    {
        Ciphertext ct(context_, 1);
        uint64_t pub[8];
        zero(*level_for(plain), true, ct, pub);
        add_plain(plain, ct);
        return save(ct, pub, out, capacity);
    }
} // namespace sealhip