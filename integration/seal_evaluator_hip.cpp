// seal::Evaluator implemented on libsealhip.so — the header-compatible link flavour of INTEGRATION.md §2.
//
// This file is what a maintainer of the reference adds NEXT TO native/src/seal/evaluator.cpp and compiles INSTEAD of it:
// it defines the out-of-line members of seal::Evaluator (native/src/seal/evaluator.h:79-1387) against the reference's own,
// unmodified headers and forwards every operation to the C ABI of include/sealhip.h.  Programs written against
// seal::Evaluator / seal::Ciphertext / seal::RelinKeys / seal::GaloisKeys link unchanged.
//
// Shape of a call: validate what the reference validates before touching data (so the same exception class is thrown for
// the same input), upload the operands (Ciphertext::data() has exactly the device layout for batch = 1), run the device
// operation, download the result and copy the metadata back.  Device state lives in a process-wide registry keyed by
// SEALContext::key_parms_id() because the class layout is fixed by the header (its only member is the context,
// evaluator.h:1385).  Key-switching keys are uploaded once per key object and cached.  This flavour pays two PCIe copies
// per call; device-resident pipelines use the batch handles of sealhip.h directly.
//
// Built by integration/Makefile into integration/_build/libsealdropin*.so together with the reference's other objects;
// tests/test_dropin.py drives it through the same flat C shim as the real reference and compares word for word.
#include "seal/evaluator.h"
#include "seal/valcheck.h"
#include "sealhip.h"
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

namespace
{
    using namespace seal;

    [[noreturn]] void raise(long hr)
    {
        uint64_t n = 0;
        SealHip_LastError(nullptr, &n);
        std::string msg(n ? n : 1, '\0');
        SealHip_LastError(&msg[0], &n);
        msg.resize(std::strlen(msg.c_str()));
        switch ((unsigned long)hr & 0xFFFFFFFFul)
        {
        case 0x80070057ul: // E_INVALIDARG
            throw std::invalid_argument(msg);
        case 0x80131509ul: // COR_E_INVALIDOPERATION
            throw std::logic_error(msg);
        case 0x80070585ul: // ERROR_INVALID_INDEX
            throw std::out_of_range(msg);
        case 0x8007000Eul:
            throw std::bad_alloc();
        default:
            throw std::runtime_error(msg.empty() ? "sealhip failure" : msg);
        }
    }
    inline void ck(long hr)
    {
        if (hr != 0)
            raise(hr);
    }

    // Device mirror of one SEALContext (all levels), its evaluator and the keys uploaded so far.
    struct Dev
    {
        void *ctx = nullptr, *ev = nullptr;
        std::mutex mu;
        std::map<std::pair<const void *, std::size_t>, void *> keys; // (first word of the key object, #indices) -> handle
    };

    Dev &device_for(const SEALContext &c)
    {
        static std::mutex mu;
        static std::map<parms_id_type, std::unique_ptr<Dev>> registry;
        std::lock_guard<std::mutex> g(mu);
        auto &slot = registry[c.key_parms_id()];
        if (slot)
            return *slot;
        auto d = std::make_unique<Dev>();
        const auto &parms = c.key_context_data()->parms();
        void *ep = nullptr;
        ck(EncParams_Create1(static_cast<uint8_t>(parms.scheme()), &ep));
        ck(EncParams_SetPolyModulusDegree(ep, parms.poly_modulus_degree()));
        std::vector<uint64_t> q;
        for (auto &m : parms.coeff_modulus())
            q.push_back(m.value());
        ck(EncParams_SetCoeffModulus(ep, q.size(), q.data()));
        if (parms.scheme() != scheme_type::ckks)
            ck(EncParams_SetPlainModulus2(ep, parms.plain_modulus().value()));
        // the chain is expanded on the device iff the reference context has levels below the first one
        const bool expanded = c.first_context_data()->next_context_data() != nullptr || c.first_context_data() == c.last_context_data();
        ck(SEALContext_Create(ep, expanded, 0, &d->ctx));
        EncParams_Destroy(ep);
        // name the levels by the reference's own parms_ids (BLAKE2b of the parameters)
        for (auto cd = c.key_context_data(); cd; cd = cd->next_context_data())
        {
            parms_id_type pid = cd->parms_id();
            ck(SEALContext_SetParmsId(d->ctx, cd->chain_index(), pid.data()));
        }
        ck(Evaluator_Create(d->ctx, &d->ev));
#ifdef SEAL_THROW_ON_TRANSPARENT_CIPHERTEXT
        ck(Evaluator_SetTransparentCheck(d->ev, true));
#endif
        slot = std::move(d);
        return *slot;
    }

    // RAII device handles
    struct DCt
    {
        void *h = nullptr;
        ~DCt()
        {
            if (h)
                Ciphertext_Destroy(h);
        }
    };
    struct DPt
    {
        void *h = nullptr;
        ~DPt()
        {
            if (h)
                Plaintext_Destroy(h);
        }
    };

    void upload(Dev &d, const Ciphertext &x, DCt &out)
    {
        ck(Ciphertext_Create3(d.ctx, nullptr, &out.h));
        if (x.size())
        {
            parms_id_type pid = x.parms_id();
            ck(Ciphertext_Resize1(out.h, d.ctx, pid.data(), x.size()));
            ck(Ciphertext_CopyFromHost(out.h, x.data(), x.size() * x.coeff_modulus_size() * x.poly_modulus_degree()));
        }
        ck(Ciphertext_SetIsNTTForm(out.h, x.is_ntt_form()));
        ck(Ciphertext_SetScale(out.h, x.scale()));
        ck(Ciphertext_SetCorrectionFactor(out.h, x.correction_factor()));
    }
    void download(const SEALContext &c, const DCt &in, Ciphertext &x)
    {
        uint64_t size = 0, pid_words[4];
        ck(Ciphertext_Size(in.h, &size));
        ck(Ciphertext_ParmsId(in.h, pid_words));
        parms_id_type pid;
        std::memcpy(pid.data(), pid_words, sizeof(pid_words));
        x.resize(c, pid, size);
        if (size)
            ck(Ciphertext_CopyToHost(in.h, x.data(), size * x.coeff_modulus_size() * x.poly_modulus_degree()));
        bool ntt = false;
        double scale = 1.0;
        uint64_t cf = 1;
        ck(Ciphertext_IsNTTForm(in.h, &ntt));
        ck(Ciphertext_Scale(in.h, &scale));
        ck(Ciphertext_CorrectionFactor(in.h, &cf));
        x.is_ntt_form() = ntt;
        x.scale() = scale;
        x.correction_factor() = cf;
    }
    void upload(Dev &d, const Plaintext &p, DPt &out)
    {
        ck(Plaintext_Create1(d.ctx, &out.h));
        ck(Plaintext_Set4(out.h, p.coeff_count(), const_cast<uint64_t *>(p.data())));
        if (p.is_ntt_form())
        {
            parms_id_type pid = p.parms_id();
            ck(Plaintext_SetParmsId(out.h, pid.data()));
        }
        ck(Plaintext_SetScale(out.h, p.scale()));
    }
    void download(const DPt &in, Plaintext &p)
    {
        uint64_t count = 0, pid_words[4];
        ck(Plaintext_CoeffCount(in.h, &count));
        ck(Plaintext_GetParmsId(in.h, pid_words));
        p.parms_id() = parms_id_zero;
        p.resize(count);
        if (count)
            ck(Plaintext_CopyToHost(in.h, p.data(), count));
        parms_id_type pid;
        std::memcpy(pid.data(), pid_words, sizeof(pid_words));
        p.parms_id() = pid;
        double scale = 1.0;
        ck(Plaintext_Scale(in.h, &scale));
        p.scale() = scale;
    }

    // KSwitchKeys::data()[index] is a vector of `digits` public keys, each a size-2 key-level ciphertext
    // (kswitchkeys.h:340): the device slab per index is their concatenation.
    void *device_keys(Dev &d, const SEALContext &c, const KSwitchKeys &keys)
    {
        if (!is_metadata_valid_for(keys, c) || !is_buffer_valid(keys))
            throw std::invalid_argument("kswitch_keys is not valid for encryption parameters");
        const void *tag = nullptr;
        for (auto &k : keys.data())
            if (!k.empty())
            {
                tag = k[0].data().data();
                break;
            }
        std::lock_guard<std::mutex> g(d.mu);
        auto key = std::make_pair(tag, keys.data().size());
        auto it = d.keys.find(key);
        if (it != d.keys.end())
            return it->second;
        void *h = nullptr;
        ck(KSwitchKeys_Create1(&h));
        for (std::size_t index = 0; index < keys.data().size(); index++)
        {
            auto &digits = keys.data()[index];
            if (digits.empty())
                continue;
            std::vector<uint64_t> words;
            for (auto &pk : digits)
            {
                const Ciphertext &kc = pk.data();
                words.insert(words.end(), kc.data(), kc.data() + kc.size() * kc.coeff_modulus_size() * kc.poly_modulus_degree());
            }
            ck(KSwitchKeys_SetKey(h, d.ctx, index, digits.size(), words.data()));
        }
        d.keys[key] = h;
        return h;
    }

    // one in-place ciphertext operation: upload, run, download
    template <class Fn>
    void unary(const SEALContext &c, Ciphertext &x, Fn fn)
    {
        Dev &d = device_for(c);
        DCt a;
        upload(d, x, a);
        ck(fn(d, a.h));
        download(c, a, x);
    }
    template <class Fn>
    void binary(const SEALContext &c, Ciphertext &x, const Ciphertext &y, Fn fn)
    {
        Dev &d = device_for(c);
        DCt a, b;
        upload(d, x, a);
        if (&x != &y)
            upload(d, y, b);
        ck(fn(d, a.h, &x == &y ? a.h : b.h));
        download(c, a, x);
    }
    void need_valid(const Ciphertext &x, const SEALContext &c, const char *what)
    {
        // is_metadata_valid_for + is_buffer_valid, as every public method of the reference does first
        if (!is_metadata_valid_for(x, c) || !is_buffer_valid(x))
            throw std::invalid_argument(std::string(what) + " is not valid for encryption parameters");
    }
} // namespace

namespace seal
{
    Evaluator::Evaluator(const SEALContext &context) : context_(context)
    {
        if (!context_.parameters_set())
            throw std::invalid_argument("encryption parameters are not set correctly");
    }

    void Evaluator::negate_inplace(Ciphertext &encrypted) const
    {
        need_valid(encrypted, context_, "encrypted");
        unary(context_, encrypted, [](Dev &d, void *a) { return Evaluator_Negate(d.ev, a, a); });
    }
    void Evaluator::add_inplace(Ciphertext &encrypted1, const Ciphertext &encrypted2) const
    {
        need_valid(encrypted1, context_, "encrypted1");
        need_valid(encrypted2, context_, "encrypted2");
        binary(context_, encrypted1, encrypted2, [](Dev &d, void *a, void *b) { return Evaluator_Add(d.ev, a, b, a); });
    }
    void Evaluator::sub_inplace(Ciphertext &encrypted1, const Ciphertext &encrypted2) const
    {
        need_valid(encrypted1, context_, "encrypted1");
        need_valid(encrypted2, context_, "encrypted2");
        binary(context_, encrypted1, encrypted2, [](Dev &d, void *a, void *b) { return Evaluator_Sub(d.ev, a, b, a); });
    }
    void Evaluator::add_many(const std::vector<Ciphertext> &encrypteds, Ciphertext &destination) const
    {
        if (encrypteds.empty())
            throw std::invalid_argument("encrypteds cannot be empty");
        for (auto &e : encrypteds)
            if (&e == &destination)
                throw std::invalid_argument("encrypteds must be different from destination");
        destination = encrypteds[0];
        for (std::size_t i = 1; i < encrypteds.size(); i++)
            add_inplace(destination, encrypteds[i]);
    }
    void Evaluator::multiply_inplace(Ciphertext &encrypted1, const Ciphertext &encrypted2, MemoryPoolHandle pool) const
    {
        need_valid(encrypted1, context_, "encrypted1");
        need_valid(encrypted2, context_, "encrypted2");
        if (!pool)
            throw std::invalid_argument("pool is uninitialized");
        binary(context_, encrypted1, encrypted2, [](Dev &d, void *a, void *b) { return Evaluator_Multiply(d.ev, a, b, a, nullptr); });
    }
    void Evaluator::square_inplace(Ciphertext &encrypted, MemoryPoolHandle pool) const
    {
        need_valid(encrypted, context_, "encrypted");
        if (!pool)
            throw std::invalid_argument("pool is uninitialized");
        unary(context_, encrypted, [](Dev &d, void *a) { return Evaluator_Square(d.ev, a, a, nullptr); });
    }
    void Evaluator::relinearize_internal(
        Ciphertext &encrypted, const RelinKeys &relin_keys, std::size_t destination_size, MemoryPoolHandle pool) const
    {
        if (!context_.get_context_data(encrypted.parms_id()))
            throw std::invalid_argument("encrypted is not valid for encryption parameters");
        if (relin_keys.parms_id() != context_.key_parms_id())
            throw std::invalid_argument("relin_keys is not valid for encryption parameters");
        if (!pool)
            throw std::invalid_argument("pool is uninitialized");
        if (destination_size != 2)
        {
            if (destination_size < 2 || destination_size > encrypted.size())
                throw std::invalid_argument("destination_size must be at least 2 and less than or equal to current count");
            throw std::logic_error("the device path relinearizes down to size 2 only");
        }
        Dev &d = device_for(context_);
        void *keys = device_keys(d, context_, relin_keys);
        unary(context_, encrypted, [keys](Dev &dd, void *a) { return Evaluator_Relinearize(dd.ev, a, keys, a, nullptr); });
    }

    void Evaluator::mod_switch_to_next(const Ciphertext &encrypted, Ciphertext &destination, MemoryPoolHandle pool) const
    {
        need_valid(encrypted, context_, "encrypted");
        if (!pool)
            throw std::invalid_argument("pool is uninitialized");
        Ciphertext work = encrypted;
        unary(context_, work, [](Dev &d, void *a) { return Evaluator_ModSwitchToNext1(d.ev, a, a, nullptr); });
        destination = std::move(work);
    }
    void Evaluator::mod_switch_to_inplace(Ciphertext &encrypted, parms_id_type parms_id, MemoryPoolHandle pool) const
    {
        if (!context_.get_context_data(encrypted.parms_id()))
            throw std::invalid_argument("encrypted is not valid for encryption parameters");
        if (!pool)
            throw std::invalid_argument("pool is uninitialized");
        unary(context_, encrypted, [&parms_id](Dev &d, void *a) { return Evaluator_ModSwitchTo1(d.ev, a, parms_id.data(), a, nullptr); });
    }
    void Evaluator::mod_switch_drop_to_next(Plaintext &plain) const
    {
        Dev &d = device_for(context_);
        DPt p;
        upload(d, plain, p);
        ck(Evaluator_ModSwitchToNext2(d.ev, p.h, p.h));
        download(p, plain);
    }
    void Evaluator::mod_switch_to_inplace(Plaintext &plain, parms_id_type parms_id) const
    {
        if (!is_valid_for(plain, context_))
            throw std::invalid_argument("plain is not valid for encryption parameters");
        Dev &d = device_for(context_);
        DPt p;
        upload(d, plain, p);
        ck(Evaluator_ModSwitchTo2(d.ev, p.h, parms_id.data(), p.h));
        download(p, plain);
    }
    void Evaluator::rescale_to_next(const Ciphertext &encrypted, Ciphertext &destination, MemoryPoolHandle pool) const
    {
        need_valid(encrypted, context_, "encrypted");
        if (!pool)
            throw std::invalid_argument("pool is uninitialized");
        Ciphertext work = encrypted;
        unary(context_, work, [](Dev &d, void *a) { return Evaluator_RescaleToNext(d.ev, a, a, nullptr); });
        destination = std::move(work);
    }
    void Evaluator::rescale_to_inplace(Ciphertext &encrypted, parms_id_type parms_id, MemoryPoolHandle pool) const
    {
        need_valid(encrypted, context_, "encrypted");
        if (!pool)
            throw std::invalid_argument("pool is uninitialized");
        unary(context_, encrypted, [&parms_id](Dev &d, void *a) { return Evaluator_RescaleTo(d.ev, a, parms_id.data(), a, nullptr); });
    }
    void Evaluator::mod_reduce_to_next_inplace(Ciphertext &encrypted, MemoryPoolHandle pool) const
    {
        need_valid(encrypted, context_, "encrypted");
        if (!pool)
            throw std::invalid_argument("pool is uninitialized");
        unary(context_, encrypted, [](Dev &d, void *a) { return Evaluator_ModReduceToNext(d.ev, a, a, nullptr); });
    }
    void Evaluator::mod_reduce_to_inplace(Ciphertext &encrypted, parms_id_type parms_id, MemoryPoolHandle pool) const
    {
        // evaluator.cpp:1620-1647: drop moduli until the target level is reached
        auto context_data_ptr = context_.get_context_data(encrypted.parms_id());
        auto target_context_data_ptr = context_.get_context_data(parms_id);
        if (!context_data_ptr)
            throw std::invalid_argument("encrypted is not valid for encryption parameters");
        if (!target_context_data_ptr)
            throw std::invalid_argument("parms_id is not valid for encryption parameters");
        if (context_data_ptr->chain_index() < target_context_data_ptr->chain_index())
            throw std::invalid_argument("cannot switch to higher level modulus");
        while (encrypted.parms_id() != parms_id)
            mod_reduce_to_next_inplace(encrypted, pool);
    }

    void Evaluator::multiply_many(
        const std::vector<Ciphertext> &encrypteds, const RelinKeys &relin_keys, Ciphertext &destination, MemoryPoolHandle pool) const
    {
        if (encrypteds.empty())
            throw std::invalid_argument("encrypteds vector must not be empty");
        if (!pool)
            throw std::invalid_argument("pool is uninitialized");
        for (auto &e : encrypteds)
            if (&e == &destination)
                throw std::invalid_argument("encrypteds must be different from destination");
        Dev &d = device_for(context_);
        void *keys = device_keys(d, context_, relin_keys);
        std::vector<DCt> handles(encrypteds.size());
        std::vector<void *> raw;
        for (std::size_t i = 0; i < encrypteds.size(); i++)
        {
            // identical operands share one device handle, as the reference detects them by data pointer (evaluator.cpp:1700)
            std::size_t same = i;
            for (std::size_t j = 0; j < i; j++)
                if (encrypteds[j].data() == encrypteds[i].data())
                    same = j;
            if (same == i)
                upload(d, encrypteds[i], handles[i]);
            raw.push_back(handles[same].h);
        }
        DCt out;
        ck(Ciphertext_Create3(d.ctx, nullptr, &out.h));
        ck(Evaluator_MultiplyMany(d.ev, raw.size(), raw.data(), keys, out.h, nullptr));
        download(context_, out, destination);
    }
    void Evaluator::exponentiate_inplace(Ciphertext &encrypted, uint64_t exponent, const RelinKeys &relin_keys, MemoryPoolHandle pool) const
    {
        if (!context_.get_context_data(encrypted.parms_id()))
            throw std::invalid_argument("encrypted is not valid for encryption parameters");
        if (!context_.get_context_data(relin_keys.parms_id()))
            throw std::invalid_argument("relin_keys is not valid for encryption parameters");
        if (!pool)
            throw std::invalid_argument("pool is uninitialized");
        if (exponent == 0)
            throw std::invalid_argument("exponent cannot be 0");
        if (exponent == 1)
            return;
        Dev &d = device_for(context_);
        void *keys = device_keys(d, context_, relin_keys);
        unary(context_, encrypted, [keys, exponent](Dev &dd, void *a) { return Evaluator_Exponentiate(dd.ev, a, exponent, keys, a, nullptr); });
    }

    void Evaluator::add_plain_inplace(Ciphertext &encrypted, const Plaintext &plain, MemoryPoolHandle) const
    {
        need_valid(encrypted, context_, "encrypted");
        if (!is_metadata_valid_for(plain, context_) || !is_buffer_valid(plain))
            throw std::invalid_argument("plain is not valid for encryption parameters");
        Dev &d = device_for(context_);
        DPt p;
        upload(d, plain, p);
        unary(context_, encrypted, [&p](Dev &dd, void *a) { return Evaluator_AddPlain(dd.ev, a, p.h, a); });
    }
    void Evaluator::sub_plain_inplace(Ciphertext &encrypted, const Plaintext &plain, MemoryPoolHandle) const
    {
        need_valid(encrypted, context_, "encrypted");
        if (!is_metadata_valid_for(plain, context_) || !is_buffer_valid(plain))
            throw std::invalid_argument("plain is not valid for encryption parameters");
        Dev &d = device_for(context_);
        DPt p;
        upload(d, plain, p);
        unary(context_, encrypted, [&p](Dev &dd, void *a) { return Evaluator_SubPlain(dd.ev, a, p.h, a); });
    }
    void Evaluator::multiply_plain_inplace(Ciphertext &encrypted, const Plaintext &plain, MemoryPoolHandle pool) const
    {
        need_valid(encrypted, context_, "encrypted");
        if (!is_metadata_valid_for(plain, context_) || !is_buffer_valid(plain))
            throw std::invalid_argument("plain is not valid for encryption parameters");
        if (!pool)
            throw std::invalid_argument("pool is uninitialized");
        Dev &d = device_for(context_);
        DPt p;
        upload(d, plain, p);
        unary(context_, encrypted, [&p](Dev &dd, void *a) { return Evaluator_MultiplyPlain(dd.ev, a, p.h, a, nullptr); });
    }
    void Evaluator::transform_to_ntt_inplace(Plaintext &plain, parms_id_type parms_id, MemoryPoolHandle pool) const
    {
        if (!is_valid_for(plain, context_))
            throw std::invalid_argument("plain is not valid for encryption parameters");
        if (!pool)
            throw std::invalid_argument("pool is uninitialized");
        Dev &d = device_for(context_);
        DPt p;
        upload(d, plain, p);
        ck(Evaluator_TransformToNTT1(d.ev, p.h, parms_id.data(), p.h, nullptr));
        download(p, plain);
    }
    void Evaluator::transform_to_ntt_inplace(Ciphertext &encrypted) const
    {
        need_valid(encrypted, context_, "encrypted");
        unary(context_, encrypted, [](Dev &d, void *a) { return Evaluator_TransformToNTT2(d.ev, a, a); });
    }
    void Evaluator::transform_from_ntt_inplace(Ciphertext &encrypted_ntt) const
    {
        need_valid(encrypted_ntt, context_, "encrypted");
        unary(context_, encrypted_ntt, [](Dev &d, void *a) { return Evaluator_TransformFromNTT(d.ev, a, a); });
    }

    void Evaluator::apply_galois_inplace(
        Ciphertext &encrypted, uint32_t galois_elt, const GaloisKeys &galois_keys, MemoryPoolHandle pool) const
    {
        need_valid(encrypted, context_, "encrypted");
        if (galois_keys.parms_id() != context_.key_parms_id())
            throw std::invalid_argument("galois_keys is not valid for encryption parameters");
        if (!pool)
            throw std::invalid_argument("pool is uninitialized");
        Dev &d = device_for(context_);
        void *keys = device_keys(d, context_, galois_keys);
        unary(context_, encrypted, [keys, galois_elt](Dev &dd, void *a) { return Evaluator_ApplyGalois(dd.ev, a, galois_elt, keys, a, nullptr); });
    }
    void Evaluator::rotate_internal(Ciphertext &encrypted, int steps, const GaloisKeys &galois_keys, MemoryPoolHandle pool) const
    {
        auto context_data_ptr = context_.get_context_data(encrypted.parms_id());
        if (!context_data_ptr)
            throw std::invalid_argument("encrypted is not valid for encryption parameters");
        if (!context_data_ptr->qualifiers().using_batching)
            throw std::logic_error("encryption parameters do not support batching");
        if (galois_keys.parms_id() != context_.key_parms_id())
            throw std::invalid_argument("galois_keys is not valid for encryption parameters");
        if (!pool)
            throw std::invalid_argument("pool is uninitialized");
        if (steps == 0)
            return;
        Dev &d = device_for(context_);
        void *keys = device_keys(d, context_, galois_keys);
        const bool ckks = context_data_ptr->parms().scheme() == scheme_type::ckks;
        unary(context_, encrypted, [keys, steps, ckks](Dev &dd, void *a) {
            return ckks ? Evaluator_RotateVector(dd.ev, a, steps, keys, a, nullptr) : Evaluator_RotateRows(dd.ev, a, steps, keys, a, nullptr);
        });
    }
} // namespace seal
