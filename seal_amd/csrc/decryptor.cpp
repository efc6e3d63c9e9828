// See decryptor.h.  Reference: native/src/seal/decryptor.cpp.
#include "decryptor.h"
#include "hostmath.h"
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
    void SecretKey::set(const void *host_words, size_t word_count)
    {
        const size_t want = ctx_->key_level().K * ctx_->n();
        if (!host_words || word_count != want)
            throw std::invalid_argument("secret_key is not valid for encryption parameters");
        if (!dev_)
            ck(hipMalloc(reinterpret_cast<void **>(&dev_), want * 8), "hipMalloc secret key");
        ck(hipMemcpy(dev_, host_words, want * 8, hipMemcpyHostToDevice), "upload secret key");
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
} // namespace sealhip
