#include "keygen.h"
#include "hostmath.h"
#include "poly_kernels.h"
#include "pool.h"
#include "xof.h"
#include <cstdlib>
#include <cstring>
#include <stdexcept>

namespace sealhip
{
    namespace
    {
        void ck(hipError_t e, const char *what)
        {
            if (e != hipSuccess)
                throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
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

    KeyGenerator::KeyGenerator(const Context &context, const uint64_t *seed8) : context_(context), sk_(context)
    {
        if (seed8)
        {
            std::memcpy(seed_, seed8, sizeof(seed_));
            seeded_ = true;
        }
        sample_secret_key();
        configure(seed8);
    }
    KeyGenerator::KeyGenerator(const Context &context, const SecretKey &secret_key, const uint64_t *seed8) : context_(context), sk_(context)
    {
        if (&secret_key.context() != &context || !secret_key.data())
            throw std::invalid_argument("secret key is not valid for encryption parameters"); // keygenerator.cpp:46-49
        if (seed8)
        {
            std::memcpy(seed_, seed8, sizeof(seed_));
            seeded_ = true;
        }
        const size_t words = context.key_level().K * context.n();
        ck(hipMemcpy(sk_.allocate(), secret_key.data(), words * 8, hipMemcpyDeviceToDevice), "copy secret key");
        configure(seed8);
    }
    void KeyGenerator::configure(const uint64_t *seed8)
    {
        encryptor_ = std::make_unique<Encryptor>(context_, sk_);
        if (seed8)
            encryptor_->set_seed(seed8);
    }

    // generate_sk (keygenerator.cpp:56-91): s <- R_3 from a fresh PRNG of the factory, replicated into the L key primes, NTT
    void KeyGenerator::sample_secret_key()
    {
        const size_t n = context_.n(), L = context_.key_level().K;
        const unsigned n_log = (unsigned)context_.log_n();
        uint64_t seed[8];
        if (seeded_)
            std::memcpy(seed, seed_, sizeof(seed));
        else
            host::random_bytes(seed, sizeof(seed));
        uint64_t *s = sk_.allocate();
        const size_t small_words = (n + 7) / 8;
        Scratch ds(small_words + 1), stream(n >= 16 ? n / 2 : 1);
        int8_t *dsmall = reinterpret_cast<int8_t *>(ds.p);
        unsigned *redraw = reinterpret_cast<unsigned *>(ds.p + small_words);
        // the 4 n bytes of the ternary draws come off the device's BLAKE2Xb unless one of them has to be redrawn (see Encryptor)
        bool host_sampling = n < 16 || std::getenv("SEALHIP_ENCRYPT_HOST_SAMPLING");
        for (;;)
        {
            if (host_sampling)
            {
                serial::Prng prng(1, seed);
                std::vector<int8_t> small(n);
                serial::sample_small_ternary(prng, n, small.data());
                ck(hipMemcpy(ds.p, small.data(), n, hipMemcpyHostToDevice), "upload s");
            }
            else
            {
                XofSeed xs;
                std::memcpy(xs.w, seed, sizeof(xs.w));
                ck(hipMemsetAsync(redraw, 0, 8, nullptr), "clear flag");
                ck(k_blake2xb_stream(xs, 0, 4 * n / 64, stream.p, nullptr), "secret key stream");
                ck(k_small_from_stream(reinterpret_cast<const uint8_t *>(stream.p), n, 0, 0, dsmall, redraw, nullptr), "sample s");
            }
            ck(k_expand_small(context_.dev_mods(), dsmall, s, n_log, (unsigned)L, 1, nullptr), "expand s");
            unsigned flag = 0;
            if (!host_sampling)
                ck(hipMemcpy(&flag, redraw, sizeof(flag), hipMemcpyDeviceToHost), "read flag");
            if (!flag)
                break;
            host_sampling = true;
        }
        ck(ntt_forward(context_.ntt_tables(), polys(s, L, n, 1), 0, nullptr), "ntt s");
        // the draws s was made from go back to the pool cleared (the reference's seal_memzero of its secret-key copies)
        ck(hipMemsetAsync(ds.p, 0, (small_words + 1) * 8, nullptr), "clear s bytes");
        if (n >= 16)
            ck(hipMemsetAsync(stream.p, 0, n / 2 * 8, nullptr), "clear stream");
        ck(hipStreamSynchronize(nullptr), "keygen sync");
        std::memset(seed, 0, sizeof(seed));
    }

    void KeyGenerator::create_public_key(PublicKey &destination)
    {
        if (&destination.context() != &context_)
            throw std::invalid_argument("destination belongs to another context");
        const Level &kl = context_.key_level();
        const size_t words = kl.K * context_.n();
        Ciphertext ct(context_, 1);
        encryptor_->zero(kl, false, ct, nullptr, false, true);
        uint64_t *pk = destination.allocate();
        ck(hipMemcpy(pk, ct.plane(0), words * 8, hipMemcpyDeviceToDevice), "copy c0");
        ck(hipMemcpy(pk + words, ct.plane(1), words * 8, hipMemcpyDeviceToDevice), "copy c1");
    }

    size_t KeyGenerator::key_words() const
    {
        return context_.first_level().K * 2 * context_.key_level().K * context_.n();
    }

    void KeyGenerator::one_kswitch_key(const uint64_t *new_key, uint64_t *out, uint64_t *public_seeds)
    {
        if (!context_.using_keyswitching())
            throw std::logic_error("keyswitching is not supported by the context"); // keygenerator.cpp:324-327
        const Level &kl = context_.key_level();
        const size_t n = context_.n(), L = kl.K, digits = context_.first_level().K, words = L * n;
        const unsigned n_log = (unsigned)context_.log_n();
        const ModDesc *mods = context_.dev_mods();
        const uint64_t special = context_.coeff_modulus()[L - 1];
        Ciphertext ct(context_, 1);
        Scratch temp(n);
        const PlaneGeom one{ n_log, 1, 1 };
        for (size_t j = 0; j < digits; j++)
        {
            // digit j = an encryption of zero under s with (special prime mod q_j) * new_key added into component j of c_0
            encryptor_->zero(kl, false, ct, public_seeds ? public_seeds + 8 * j : nullptr, false, true);
            uint64_t *dj = out + j * 2 * words;
            ck(hipMemcpyAsync(dj, ct.plane(0), words * 8, hipMemcpyDeviceToDevice, nullptr), "copy c0");
            ck(hipMemcpyAsync(dj + words, ct.plane(1), words * 8, hipMemcpyDeviceToDevice, nullptr), "copy c1");
            const uint64_t factor = special % context_.coeff_modulus()[j];
            ck(k_mul_scalar(mods + j, new_key + j * n, temp.p, factor, one, 1, nullptr), "factor * key");
            ck(k_addsub(mods + j, dj + j * n, temp.p, dj + j * n, 0, one, 1, nullptr), "c0 + factor * key");
            ck(hipStreamSynchronize(nullptr), "keygen sync");
        }
    }
    void KeyGenerator::relin_key(uint64_t *out, uint64_t *public_seeds)
    {
        // compute_secret_key_array (keygenerator.cpp:220-290): s^2 = s .* s in NTT form
        const size_t n = context_.n(), L = context_.key_level().K;
        Scratch s2(L * n);
        ck(k_dyadic(context_.dev_mods(), sk_.data(), sk_.data(), s2.p, (unsigned)context_.log_n(), (unsigned)L, 0, 1, nullptr), "s^2");
        one_kswitch_key(s2.p, out, public_seeds);
        ck(hipMemsetAsync(s2.p, 0, L * n * 8, nullptr), "clear s^2");
        ck(hipStreamSynchronize(nullptr), "keygen sync");
    }
    void KeyGenerator::galois_key(uint32_t galois_elt, uint64_t *out, uint64_t *public_seeds)
    {
        const size_t n = context_.n(), L = context_.key_level().K;
        if (!(galois_elt & 1) || galois_elt >= 2 * n)
            throw std::invalid_argument("Galois element is not valid"); // keygenerator.cpp:186-189
        Scratch rotated(L * n);
        const PlaneGeom g{ (unsigned)context_.log_n(), (unsigned)L, 1 };
        ck(k_apply_galois(context_.dev_mods(), sk_.data(), rotated.p, galois_elt, 1, g, 1, nullptr), "rotate s");
        one_kswitch_key(rotated.p, out, public_seeds);
        ck(hipMemsetAsync(rotated.p, 0, L * n * 8, nullptr), "clear rotated s");
        ck(hipStreamSynchronize(nullptr), "keygen sync");
    }

    void KeyGenerator::create_relin_keys(KSwitchKeys &destination)
    {
        Scratch key(key_words());
        relin_key(key.p);
        destination.set_key(context_, 0, context_.first_level().K, key.p, true); // RelinKeys::get_index(2) = 0
        ck(hipDeviceSynchronize(), "keygen sync");
    }
    void KeyGenerator::create_galois_keys(const uint32_t *galois_elts, size_t count, KSwitchKeys &destination)
    {
        if (count && !galois_elts)
            throw std::invalid_argument("galois_elts");
        Scratch key(key_words());
        destination.reserve_slots(context_.n()); // GaloisKeys::data() has a slot for every odd element (keygenerator.cpp:209-214)
        for (size_t i = 0; i < count; i++)
        {
            const uint32_t elt = galois_elts[i];
            if (!(elt & 1) || elt >= 2 * context_.n())
                throw std::invalid_argument("Galois element is not valid");
            const size_t index = (elt - 1) >> 1; // GaloisKeys::get_index (galoiskeys.h:48)
            if (destination.has_key(index))
                continue;
            galois_key(elt, key.p);
            destination.set_key(context_, index, context_.first_level().K, key.p, true);
            ck(hipDeviceSynchronize(), "keygen sync");
        }
    }
    void KeyGenerator::create_galois_keys_from_steps(const int *steps, size_t count, KSwitchKeys &destination)
    {
        if (!context_.using_batching())
            throw std::logic_error("encryption parameters do not support batching"); // keygenerator.h:221-224
        if (count && !steps)
            throw std::invalid_argument("steps");
        std::vector<uint32_t> elts(count);
        for (size_t i = 0; i < count; i++)
            elts[i] = encryptor_->evaluator_.galois_elt_from_step(steps[i]);
        create_galois_keys(elts.data(), elts.size(), destination);
    }
    // GaloisTool::get_elts_all (util/galois.cpp:106-131)
    std::vector<uint32_t> KeyGenerator::galois_elts_all() const
    {
        const uint64_t m = (uint64_t)context_.n() << 1;
        std::vector<uint32_t> elts{ (uint32_t)(m - 1) };
        uint64_t pos = 3, neg = 1;
        // 3^-1 mod m: the unit group of Z / 2^k has exponent m / 4, so 3^(m/2) = 1 and 3^(m/2 - 1) is the inverse
        for (uint64_t e = m / 2 - 1, b = 3; e; e >>= 1, b = (b * b) & (m - 1))
            if (e & 1)
                neg = (neg * b) & (m - 1);
        for (unsigned i = 0; i + 1 < context_.log_n(); i++)
        {
            elts.push_back((uint32_t)pos);
            pos = (pos * pos) & (m - 1);
            elts.push_back((uint32_t)neg);
            neg = (neg * neg) & (m - 1);
        }
        return elts;
    }
    void KeyGenerator::create_galois_keys_all(KSwitchKeys &destination)
    {
        if (!context_.using_batching())
            throw std::logic_error("encryption parameters do not support batching");
        const std::vector<uint32_t> elts = galois_elts_all();
        create_galois_keys(elts.data(), elts.size(), destination);
    }
    size_t KeyGenerator::seeded_save_size(bool galois, size_t key_count) const
    {
        const size_t n = context_.n(), L = context_.key_level().K, digits = context_.first_level().K;
        const size_t slots = galois ? n : 1;
        return 16 + 32 + 8 + slots * 8 + key_count * digits * serial::seeded_ciphertext_save_size(n, L);
    }
    size_t KeyGenerator::save_seeded(bool galois, const uint32_t *galois_elts, size_t count, uint8_t *out, size_t capacity)
    {
        if (!out || (galois && count && !galois_elts))
            throw std::invalid_argument("out");
        const Level &kl = context_.key_level();
        const size_t n = context_.n(), L = kl.K, digits = context_.first_level().K, words = L * n;
        // the slots of KSwitchKeys::data(): one for RelinKeys (count = 1), N for GaloisKeys with the key of element e at (e - 1) / 2
        std::vector<uint32_t> slot_elt(galois ? n : 1, 0);
        size_t keys = 1;
        if (galois)
        {
            keys = 0;
            for (size_t i = 0; i < count; i++)
            {
                const uint32_t e = galois_elts[i];
                if (!(e & 1) || e >= 2 * n)
                    throw std::invalid_argument("Galois element is not valid");
                if (!slot_elt[(e - 1) >> 1])
                    keys++;
                slot_elt[(e - 1) >> 1] = e;
            }
        }
        if (capacity < seeded_save_size(galois, keys))
            throw std::invalid_argument("capacity");
        size_t pos = 16;
        auto put64 = [&](uint64_t v) {
            std::memcpy(out + pos, &v, 8);
            pos += 8;
        };
        std::memcpy(out + pos, kl.parms_id, 32);
        pos += 32;
        put64(slot_elt.size());
        Scratch key(key_words());
        std::vector<uint64_t> seeds(digits * 8);
        for (size_t slot = 0; slot < slot_elt.size(); slot++)
        {
            if (galois && !slot_elt[slot])
            {
                put64(0);
                continue;
            }
            put64(digits);
            if (galois)
                galois_key(slot_elt[slot], key.p, seeds.data());
            else
                relin_key(key.p, seeds.data());
            for (size_t j = 0; j < digits; j++)
            {
                size_t off = 0;
                const size_t bytes = serial::save_seeded_ciphertext(kl.parms_id, true, n, L, 1.0, 1, nullptr, 1, seeds.data() + 8 * j, out + pos,
                                                                    capacity - pos, &off);
                ck(hipMemcpy(out + pos + off, key.p + j * 2 * words, words * 8, hipMemcpyDeviceToHost), "download c0");
                pos += bytes;
            }
        }
        // the outer SEALHeader (serialization.h: magic, header size, version, compr_mode none, reserved, size)
        const uint8_t header[8] = { 0x5E, 0xA1, serial::kHeaderSize, serial::kVersionMajor, serial::kVersionMinor, 0, 0, 0 };
        std::memcpy(out, header, 8);
        const uint64_t total = pos;
        std::memcpy(out + 8, &total, 8);
        return pos;
    }
    void KeyGenerator::key_to_host(uint32_t galois_elt, uint64_t *host_words)
    {
        if (!host_words)
            throw std::invalid_argument("host_words");
        Scratch key(key_words());
        if (galois_elt)
            galois_key(galois_elt, key.p);
        else
            relin_key(key.p);
        ck(hipMemcpy(host_words, key.p, key_words() * 8, hipMemcpyDeviceToHost), "download key");
    }
} // namespace sealhip
