// C ABI, container surface: what a sealc binding P/Invokes on Ciphertext / KSwitchKeys / SEALContext / ContextData /
// EncryptionParameters / SecretKey / PublicKey besides the hot path (native/src/seal/c/ciphertext.h:26-60, kswitchkeys.h:29-33,
// sealcontext.h:30-34, contextdata.h, encryptionparameters.h:47-49, secretkey.h:32, publickey.h:32).  Device-batch semantics where an
// object lives in HBM (a word index addresses the slab [poly][batch][K][N] - for a batch of one that is Ciphertext::data()), host
// arithmetic for the context constants.  include/sealhip.h lists every sealc function of these seven headers that is deliberately
// NOT here (the MemoryPoolHandle accessors, context-less constructors, handles to internal members) with the reason;
// tests/test_cabi.py diffs that list and this file against the reference headers.
#include "capi_common.h"
#include "blake2.h"
#include <algorithm>

namespace
{
    // ---------------------------------------------------------------- host multi-precision helpers (little-endian words)
    typedef std::vector<uint64_t> Big;
    Big level_product(const Context &c, const Level &l)
    {
        Big q(c.coeff_modulus().begin(), c.coeff_modulus().begin() + l.K);
        Big p = host::product(q);
        p.resize(l.K, 0); // total_coeff_modulus has coeff_modulus_size words (context.cpp:262-266)
        return p;
    }
    uint64_t big_mod(const Big &a, uint64_t m)
    {
        host::u128 r = 0;
        for (size_t i = a.size(); i-- > 0;)
            r = ((r << 64) | a[i]) % m;
        return (uint64_t)r;
    }
    Big big_sub_small(Big a, uint64_t s)
    {
        for (size_t i = 0; i < a.size() && s; i++)
        {
            const uint64_t before = a[i];
            a[i] -= s;
            s = before < s ? 1 : 0;
        }
        return a;
    }
    Big big_half_up(Big a) // (a + 1) >> 1
    {
        for (size_t i = 0; i < a.size(); i++)
            if (++a[i] != 0)
                break;
        for (size_t i = 0; i < a.size(); i++)
            a[i] = (a[i] >> 1) | (i + 1 < a.size() ? a[i + 1] << 63 : 0);
        return a;
    }

    // PrepareOutputBuffer of the reference's c/utilities.h: *count carries the capacity in and the required length out
    SHL_HRESULT copy_out(const Big &v, uint64_t *count, uint64_t *dst)
    {
        const uint64_t capacity = *count;
        *count = v.size();
        if (!dst)
            return SHL_S_OK;
        if (capacity < v.size())
            return SHL_E_INVALIDARG;
        std::memcpy(dst, v.data(), v.size() * 8);
        return SHL_S_OK;
    }

    struct Qualifiers // EncryptionParameterQualifiers (context.h:44-179) of one level of a context that was accepted
    {
        bool using_fft = true, using_ntt = true, using_batching = false, using_fast_plain_lift = false, using_descending_modulus_chain = false;
        int sec_level = 0;
    };

    // a ContextData handle names one Level of one context (a Level does not know its context)
    struct LevelRef
    {
        const Context *ctx;
        const Level *level;
    };
    // the handles given out, owned here: one per (context, level) ever asked for
    std::mutex g_refs_mu;
    std::vector<std::unique_ptr<LevelRef>> g_refs; // (a handful per context; released by SEALContext_Destroy)
    void *level_handle(const Context *c, const Level *l)
    {
        if (!l)
            return nullptr;
        std::lock_guard<std::mutex> g(g_refs_mu);
        for (auto &r : g_refs)
            if (r->ctx == c && r->level == l)
                return r.get();
        g_refs.emplace_back(new LevelRef{ c, l });
        return g_refs.back().get();
    }
    bool fast_plain_lift(const Context &c, const Level &l)
    {
        if (c.scheme() == Scheme::ckks)
            return false;
        bool f = true;
        for (unsigned i = 0; i < l.K; i++)
            f &= c.coeff_modulus()[i] > c.plain_modulus();
        return f;
    }

    EncParams level_parms(const Context &c, const Level &l)
    {
        EncParams p;
        p.scheme = (uint8_t)c.scheme();
        p.n = c.n();
        p.coeff_modulus.assign(c.coeff_modulus().begin(), c.coeff_modulus().begin() + l.K);
        p.plain_modulus = c.plain_modulus();
        return p;
    }
    void parms_id_of(const EncParams &p, uint64_t *out)
    {
        // EncryptionParameters::compute_parms_id (encryptionparams.cpp:117-147): BLAKE2b-256 of (scheme, N, primes, t)
        std::vector<uint64_t> words;
        words.push_back(p.scheme);
        words.push_back(p.n);
        words.insert(words.end(), p.coeff_modulus.begin(), p.coeff_modulus.end());
        words.push_back(p.plain_modulus);
        blake2::blake2b(out, 32, words.data(), words.size() * 8);
    }

    // ---- EncryptionParameters stream (encryptionparams.cpp:15-49; Modulus::save_members, modulus.cpp:18-40)
    struct Header
    {
        uint16_t magic;
        uint8_t header_size, version_major, version_minor, compr_mode;
        uint16_t reserved;
        uint64_t size;
    };
    size_t parms_raw_size(const EncParams &p)
    {
        return 16 + 1 + 8 + 8 + (p.coeff_modulus.size() + 1) * (16 + 8);
    }
    void parms_write(const EncParams &p, uint8_t *out)
    {
        uint8_t *o = out;
        auto put = [&](const void *src, size_t n) {
            std::memcpy(o, src, n);
            o += n;
        };
        const Header h{ serial::kMagic, serial::kHeaderSize, serial::kVersionMajor, serial::kVersionMinor, 0, 0, (uint64_t)parms_raw_size(p) };
        put(&h, 16);
        put(&p.scheme, 1);
        const uint64_t n = p.n, k = p.coeff_modulus.size();
        put(&n, 8);
        put(&k, 8);
        const Header hm{ serial::kMagic, serial::kHeaderSize, serial::kVersionMajor, serial::kVersionMinor, 0, 0, 24 };
        for (uint64_t q : p.coeff_modulus)
        {
            put(&hm, 16);
            put(&q, 8);
        }
        put(&hm, 16);
        put(&p.plain_modulus, 8);
    }
} // namespace

namespace sealhip
{
    void capi_forget_context(const Context *context)
    {
        std::lock_guard<std::mutex> g(g_refs_mu);
        g_refs.erase(std::remove_if(g_refs.begin(), g_refs.end(), [&](const std::unique_ptr<LevelRef> &r) { return r->ctx == context; }), g_refs.end());
    }
} // namespace sealhip

extern "C"
{
    // ================================================================ Ciphertext (c/ciphertext.h:26-60)
    SHL_FUNC Ciphertext_Create4(void *context, uint64_t *parms_id, void *pool, void **cipher)
    {
        (void)pool;
        IfNullRet(context, SHL_E_POINTER);
        IfNullRet(parms_id, SHL_E_POINTER);
        IfNullRet(cipher, SHL_E_POINTER);
        SHL_TRY
        auto c = as<Context>(context);
        std::unique_ptr<Ciphertext> ct(new Ciphertext(*c, 1));
        hip_ok(hipDeviceSynchronize(), "sync");
        ct->reserve(c->level_by_parms_id(parms_id), 2, nullptr); // Ciphertext(context, parms_id): reserve(context, parms_id, 2) (ciphertext.h:128-133)
        *cipher = ct.release();
        SHL_CATCH
    }
    SHL_FUNC Ciphertext_Create5(void *context, uint64_t *parms_id, uint64_t capacity, void *pool, void **cipher)
    {
        (void)pool;
        IfNullRet(context, SHL_E_POINTER);
        IfNullRet(parms_id, SHL_E_POINTER);
        IfNullRet(cipher, SHL_E_POINTER);
        SHL_TRY
        auto c = as<Context>(context);
        std::unique_ptr<Ciphertext> ct(new Ciphertext(*c, 1));
        hip_ok(hipDeviceSynchronize(), "sync");
        ct->reserve(c->level_by_parms_id(parms_id), (size_t)capacity, nullptr);
        *cipher = ct.release();
        SHL_CATCH
    }
    SHL_FUNC Ciphertext_Reserve1(void *thisptr, void *context, uint64_t *parms_id, uint64_t size_capacity)
    {
        IfNullRet(context, SHL_E_POINTER);
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(parms_id, SHL_E_POINTER);
        SHL_TRY
        auto ct = as<Ciphertext>(thisptr);
        auto c = as<Context>(context);
        if (&ct->context() != c)
            throw std::invalid_argument("ciphertext belongs to another context");
        hip_ok(hipDeviceSynchronize(), "sync");
        ct->reserve(c->level_by_parms_id(parms_id), (size_t)size_capacity, nullptr);
        SHL_CATCH
    }
    SHL_FUNC Ciphertext_Reserve2(void *thisptr, void *context, uint64_t size_capacity)
    {
        IfNullRet(context, SHL_E_POINTER);
        IfNullRet(thisptr, SHL_E_POINTER);
        SHL_TRY
        auto ct = as<Ciphertext>(thisptr);
        if (&ct->context() != as<Context>(context))
            throw std::invalid_argument("ciphertext belongs to another context");
        hip_ok(hipDeviceSynchronize(), "sync");
        ct->reserve(ct->level(), (size_t)size_capacity, nullptr); // reserve(context, parms_id_, size_capacity): an unset parms_id is refused
        SHL_CATCH
    }
    SHL_FUNC Ciphertext_Reserve3(void *thisptr, uint64_t size_capacity)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        SHL_TRY
        auto ct = as<Ciphertext>(thisptr);
        if (size_capacity < 2 || size_capacity > 16)
            throw std::invalid_argument("invalid size_capacity");
        hip_ok(hipDeviceSynchronize(), "sync");
        if (ct->level()) // reserve_internal with the current geometry; without one there are no words to make room for (N = K = 0)
            ct->reserve(ct->level(), (size_t)size_capacity, nullptr);
        SHL_CATCH
    }
    SHL_FUNC Ciphertext_SizeCapacity(void *thisptr, uint64_t *size_capacity)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(size_capacity, SHL_E_POINTER);
        *size_capacity = as<Ciphertext>(thisptr)->size_capacity();
        return SHL_S_OK;
    }
    SHL_FUNC Ciphertext_SetParmsId(void *thisptr, uint64_t *parms_id)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(parms_id, SHL_E_POINTER);
        SHL_TRY
        auto ct = as<Ciphertext>(thisptr);
        static const uint64_t zero[4] = { 0, 0, 0, 0 };
        const Level *l = ct->context().level_by_parms_id(parms_id);
        // the reference stores any 256 bits (c/ciphertext.cpp:239-247); a device object names its level by pointer, so only the
        // ids of this context's chain - or parms_id_zero - can be taken
        if (!l && std::memcmp(parms_id, zero, 32) != 0)
            throw std::invalid_argument("parms_id is not valid for encryption parameters");
        if (l && ct->level() && l->K != ct->level()->K && ct->word_count())
            throw std::invalid_argument("parms_id names a level with another coeff_modulus_size than the stored polynomials");
        // ... and never one whose polynomials would not fit the slab (parms_id_zero first and then a longer level's id
        // passed the check above - no level, word_count() 0 - and left size x K x N beyond capacity_words()).  The reference's
        // is_buffer_valid catches the same state later (valcheck.cpp:172-198); a device object refuses to enter it.
        if (l && ct->size() * ct->batch() * l->K * ct->context().n() > ct->capacity_words())
            throw std::invalid_argument("parms_id names a level whose polynomials do not fit the allocated capacity");
        ct->set_level_unchecked(l);
        SHL_CATCH
    }
    SHL_FUNC Ciphertext_Resize2(void *thisptr, void *context, uint64_t size)
    {
        IfNullRet(context, SHL_E_POINTER);
        IfNullRet(thisptr, SHL_E_POINTER);
        SHL_TRY
        auto ct = as<Ciphertext>(thisptr);
        if (&ct->context() != as<Context>(context))
            throw std::invalid_argument("ciphertext belongs to another context");
        hip_ok(hipDeviceSynchronize(), "sync");
        ct->resize(ct->level(), (size_t)size, nullptr); // resize(context, parms_id_, size): throws for an unset parms_id
        SHL_CATCH
    }
    SHL_FUNC Ciphertext_Resize3(void *thisptr, uint64_t size)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        SHL_TRY
        auto ct = as<Ciphertext>(thisptr);
        if ((size < 2 && size != 0) || size > 16)
            throw std::invalid_argument("invalid size");
        hip_ok(hipDeviceSynchronize(), "sync");
        if (ct->level())
            ct->resize(ct->level(), (size_t)size, nullptr);
        else if (size)
            throw std::logic_error("a device ciphertext without parms_id has no polynomial geometry to resize with");
        SHL_CATCH
    }
    SHL_FUNC Ciphertext_Resize4(void *thisptr, uint64_t size, uint64_t polyModulusDegree, uint64_t coeffModCount)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        SHL_TRY
        auto ct = as<Ciphertext>(thisptr);
        // the .NET loader's private resize(size, N, K) (c/ciphertext.cpp:295-306): the geometry must be one of this context's levels
        const Context &c = ct->context();
        const Level *l = ct->level() && ct->level()->K == coeffModCount ? ct->level() : nullptr;
        if (!l)
            for (auto &lv : c.levels())
                if (lv.K == coeffModCount && &lv != &c.key_level())
                    l = &lv;
        if (!l && c.key_level().K == coeffModCount)
            l = &c.key_level();
        if (!l || polyModulusDegree != c.n())
            throw std::invalid_argument("no level of the context has this poly_modulus_degree and coeff_modulus_size");
        hip_ok(hipDeviceSynchronize(), "sync");
        ct->resize(l, (size_t)size, nullptr);
        SHL_CATCH
    }
    SHL_FUNC Ciphertext_GetDataAt1(void *thisptr, uint64_t index, uint64_t *data)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(data, SHL_E_POINTER);
        SHL_TRY
        auto ct = as<Ciphertext>(thisptr);
        if (index >= ct->word_count())
            throw std::out_of_range("index must be within [0, size)"); // DynArray::at
        hip_ok(hipDeviceSynchronize(), "sync");
        copy_d2h(data, ct->data() + index, 8);
        SHL_CATCH
    }
    SHL_FUNC Ciphertext_GetDataAt2(void *thisptr, uint64_t poly_index, uint64_t coeff_index, uint64_t *data)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(data, SHL_E_POINTER);
        SHL_TRY
        auto ct = as<Ciphertext>(thisptr);
        const size_t poly_words = ct->poly_modulus_degree() * ct->coeff_modulus_size();
        if (coeff_index >= poly_words)
            return SHL_E_INVALID_INDEX;
        if (poly_index >= ct->size())
            throw std::out_of_range("poly_index must be within [0, size)"); // Ciphertext::data(poly_index) (ciphertext.h:371-379)
        hip_ok(hipDeviceSynchronize(), "sync");
        copy_d2h(data, ct->plane(poly_index) + coeff_index, 8); // batch item 0
        SHL_CATCH
    }
    SHL_FUNC Ciphertext_SetDataAt(void *thisptr, uint64_t index, uint64_t value)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        SHL_TRY
        auto ct = as<Ciphertext>(thisptr);
        if (index >= ct->word_count())
            throw std::out_of_range("index must be within [0, size)");
        hip_ok(hipDeviceSynchronize(), "sync");
        copy_h2d(ct->data() + index, &value, 8);
        SHL_CATCH
    }
    SHL_FUNC Ciphertext_Release(void *thisptr)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        SHL_TRY
        auto ct = as<Ciphertext>(thisptr);
        hip_ok(hipDeviceSynchronize(), "sync");
        ct->release(); // Ciphertext::release (ciphertext.h:583-592): parms_id_zero, size 0, no storage, scale 1, coefficient form
        ct->is_ntt_form() = false;
        ct->scale() = 1.0;
        ct->correction_factor() = 1;
        SHL_CATCH
    }

    // ================================================================ KSwitchKeys (c/kswitchkeys.h:20-33)
    SHL_FUNC KSwitchKeys_Create2(void *copy, void **kswitch_keys)
    {
        IfNullRet(copy, SHL_E_POINTER);
        IfNullRet(kswitch_keys, SHL_E_POINTER);
        SHL_TRY
        std::unique_ptr<KSwitchKeys> k(new KSwitchKeys());
        hip_ok(hipDeviceSynchronize(), "sync");
        k->assign(*as<KSwitchKeys>(copy));
        *kswitch_keys = k.release();
        SHL_CATCH
    }
    SHL_FUNC KSwitchKeys_Set(void *thisptr, void *assign)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(assign, SHL_E_POINTER);
        SHL_TRY
        hip_ok(hipDeviceSynchronize(), "sync");
        as<KSwitchKeys>(thisptr)->assign(*as<KSwitchKeys>(assign));
        SHL_CATCH
    }
    SHL_FUNC KSwitchKeys_RawSize(void *thisptr, uint64_t *key_count)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(key_count, SHL_E_POINTER);
        *key_count = as<KSwitchKeys>(thisptr)->slots();
        return SHL_S_OK;
    }
    SHL_FUNC KSwitchKeys_GetKeyList(void *thisptr, uint64_t index, uint64_t *count, void **key_list)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(count, SHL_E_POINTER);
        SHL_TRY
        auto keys = as<KSwitchKeys>(thisptr);
        if (index >= keys->slots())
            throw std::out_of_range("index");
        const size_t digits = keys->has_key(index) ? keys->key(index).digits : 0;
        *count = digits;
        if (!key_list || !digits)
            return SHL_S_OK;
        // the reference hands out pointers INTO the object (c/kswitchkeys.cpp:79-100); the device key is one slab in the
        // kernels' register order, so every digit is given out as a PublicKey of its own: the caller destroys them
        const Context &c = *keys->context();
        const size_t words = 2 * c.key_level().K * c.n();
        Scratch natural(digits * words);
        keys->key_words(index, natural.p);
        std::vector<std::unique_ptr<PublicKey>> made;
        for (size_t j = 0; j < digits; j++)
        {
            made.emplace_back(new PublicKey(c));
            hip_ok(hipMemcpy(made.back()->allocate(), natural.p + j * words, words * 8, hipMemcpyDeviceToDevice), "copy digit");
        }
        for (size_t j = 0; j < digits; j++)
            key_list[j] = made[j].release();
        SHL_CATCH
    }
    SHL_FUNC KSwitchKeys_ClearDataAndReserve(void *thisptr, uint64_t size)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        SHL_TRY
        (void)size; // vector::reserve: capacity only
        hip_ok(hipDeviceSynchronize(), "sync");
        as<KSwitchKeys>(thisptr)->clear();
        SHL_CATCH
    }
    SHL_FUNC KSwitchKeys_AddKeyList(void *thisptr, uint64_t count, void **key_list)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(key_list, SHL_E_POINTER);
        SHL_TRY
        auto keys = as<KSwitchKeys>(thisptr);
        if (count == 0)
        {
            keys->reserve_slots(keys->slots() + 1); // data().emplace_back() with nothing in it: an empty slot
            return SHL_S_OK;
        }
        for (uint64_t j = 0; j < count; j++)
            if (!key_list[j] || !as<PublicKey>(key_list[j])->data())
                throw std::invalid_argument("key_list holds an empty public key");
        const Context &c = as<PublicKey>(key_list[0])->context();
        const size_t words = 2 * c.key_level().K * c.n();
        Scratch natural(count * words);
        hip_ok(hipDeviceSynchronize(), "sync");
        for (uint64_t j = 0; j < count; j++)
        {
            auto pk = as<PublicKey>(key_list[j]);
            if (&pk->context() != &c)
                throw std::invalid_argument("key_list mixes contexts");
            hip_ok(hipMemcpy(natural.p + j * words, pk->data(), words * 8, hipMemcpyDeviceToDevice), "copy digit");
        }
        keys->set_key(c, keys->slots(), (size_t)count, natural.p, true);
        SHL_CATCH
    }
    SHL_FUNC KSwitchKeys_GetParmsId(void *thisptr, uint64_t *parms_id)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(parms_id, SHL_E_POINTER);
        as<KSwitchKeys>(thisptr)->get_parms_id(parms_id);
        return SHL_S_OK;
    }
    SHL_FUNC KSwitchKeys_SetParmsId(void *thisptr, uint64_t *parms_id)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(parms_id, SHL_E_POINTER);
        SHL_TRY
        as<KSwitchKeys>(thisptr)->set_parms_id(parms_id);
        SHL_CATCH
    }

    // ================================================================ SEALContext (c/sealcontext.h:22-40)
    SHL_FUNC SEALContext_ParametersSet(void *thisptr, bool *params_set)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(params_set, SHL_E_POINTER);
        *params_set = true; // SEALContext_Create refuses parameters the reference would mark "not set" (E_INVALIDARG): a handle means valid
        return SHL_S_OK;
    }
    SHL_FUNC SEALContext_ParameterErrorName(void *thisptr, char *outstr, uint64_t *length)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(length, SHL_E_POINTER);
        static const char name[] = "success"; // EncryptionParameterQualifiers::parameter_error_name (context.cpp:31)
        *length = sizeof(name) - 1;
        if (outstr)
            std::memcpy(outstr, name, sizeof(name));
        return SHL_S_OK;
    }
    SHL_FUNC SEALContext_ParameterErrorMessage(void *thisptr, char *outstr, uint64_t *length)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(length, SHL_E_POINTER);
        static const char msg[] = "valid"; // parameter_error_message (context.cpp:92)
        *length = sizeof(msg) - 1;
        if (outstr)
            std::memcpy(outstr, msg, sizeof(msg));
        return SHL_S_OK;
    }
    SHL_FUNC SEALContext_KeyContextData(void *thisptr, void **context_data)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(context_data, SHL_E_POINTER);
        auto c = as<Context>(thisptr);
        *context_data = level_handle(c, &c->key_level());
        return SHL_S_OK;
    }
    SHL_FUNC SEALContext_FirstContextData(void *thisptr, void **context_data)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(context_data, SHL_E_POINTER);
        auto c = as<Context>(thisptr);
        *context_data = level_handle(c, &c->first_level());
        return SHL_S_OK;
    }
    SHL_FUNC SEALContext_LastContextData(void *thisptr, void **context_data)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(context_data, SHL_E_POINTER);
        auto c = as<Context>(thisptr);
        *context_data = level_handle(c, &c->last_level());
        return SHL_S_OK;
    }
    SHL_FUNC SEALContext_GetContextData(void *thisptr, uint64_t *parms_id, void **context_data)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(parms_id, SHL_E_POINTER);
        IfNullRet(context_data, SHL_E_POINTER);
        auto c = as<Context>(thisptr);
        *context_data = level_handle(c, c->level_by_parms_id(parms_id)); // null for an unknown id, as get_context_data
        return SHL_S_OK;
    }

    // ================================================================ ContextData (c/contextdata.h)
    // A handle is owned by the library (it names one level of a context and stays valid as long as that context does), as the
    // pointers sealc returns are owned by the SEALContext; ContextData_Destroy therefore has nothing to delete.
    SHL_FUNC ContextData_Destroy(void *thisptr)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        return SHL_S_OK;
    }
    SHL_FUNC ContextData_TotalCoeffModulus(void *thisptr, uint64_t *count, uint64_t *total_coeff_modulus)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(count, SHL_E_POINTER);
        SHL_TRY
        auto r = as<LevelRef>(thisptr);
        return copy_out(level_product(*r->ctx, *r->level), count, total_coeff_modulus);
        SHL_CATCH
    }
    SHL_FUNC ContextData_TotalCoeffModulusBitCount(void *thisptr, int *bit_count)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(bit_count, SHL_E_POINTER);
        *bit_count = as<LevelRef>(thisptr)->level->total_coeff_modulus_bit_count;
        return SHL_S_OK;
    }
    SHL_FUNC ContextData_Parms(void *thisptr, void **parms)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(parms, SHL_E_POINTER);
        SHL_TRY
        auto r = as<LevelRef>(thisptr);
        *parms = new EncParams(level_parms(*r->ctx, *r->level)); // the caller's: EncParams_Destroy
        SHL_CATCH
    }
    SHL_FUNC ContextData_Qualifiers(void *thisptr, void **epq)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(epq, SHL_E_POINTER);
        SHL_TRY
        auto r = as<LevelRef>(thisptr);
        auto q = new Qualifiers();
        q->using_batching = r->ctx->using_batching();
        q->using_fast_plain_lift = fast_plain_lift(*r->ctx, *r->level);
        // context.cpp:437-443: every prime of the level larger than the next one
        q->using_descending_modulus_chain = true;
        for (unsigned i = 0; i + 1 < r->level->K; i++)
            q->using_descending_modulus_chain &= r->ctx->coeff_modulus()[i] > r->ctx->coeff_modulus()[i + 1];
        q->sec_level = r->ctx->sec_level();
        *epq = q; // the caller's: EPQ_Destroy
        SHL_CATCH
    }
    SHL_FUNC ContextData_CoeffDivPlainModulus(void *thisptr, uint64_t *count, uint64_t *coeff_div)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(count, SHL_E_POINTER);
        SHL_TRY
        auto r = as<LevelRef>(thisptr);
        const Context &c = *r->ctx;
        if (c.scheme() == Scheme::ckks)
        {
            *count = 0; // not computed for these parameters (c/contextdata.cpp:97-101)
            return SHL_S_OK;
        }
        // floor(Q / t) mod q_i (context.cpp:331-352) = -(Q mod t) t^-1 mod q_i, Q being 0 mod q_i
        const uint64_t t = c.plain_modulus(), rem = big_mod(level_product(c, *r->level), t);
        Big v;
        for (unsigned i = 0; i < r->level->K; i++)
        {
            const uint64_t q = c.coeff_modulus()[i], rm = rem % q;
            v.push_back(host::mulmod(rm ? q - rm : 0, host::invmod(t % q, q), q));
        }
        return copy_out(v, count, coeff_div);
        SHL_CATCH
    }
    SHL_FUNC ContextData_PlainUpperHalfThreshold(void *thisptr, uint64_t *puht)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(puht, SHL_E_POINTER);
        const Context &c = *as<LevelRef>(thisptr)->ctx;
        *puht = c.scheme() == Scheme::ckks ? uint64_t(1) << 63 : (c.plain_modulus() + 1) >> 1; // context.cpp:357, 396
        return SHL_S_OK;
    }
    SHL_FUNC ContextData_PlainUpperHalfIncrement(void *thisptr, uint64_t *count, uint64_t *puhi)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(count, SHL_E_POINTER);
        SHL_TRY
        auto r = as<LevelRef>(thisptr);
        const Context &c = *r->ctx;
        Big v;
        if (c.scheme() == Scheme::ckks)
        {
            // 2^64 mod q_i written as (2^63 mod q_i)(q_i - 2) ... the reference's expression (context.cpp:399-406), word for word
            for (unsigned i = 0; i < r->level->K; i++)
            {
                const uint64_t q = c.coeff_modulus()[i];
                v.push_back(host::mulmod((uint64_t(1) << 63) % q, q - 2, q));
            }
        }
        else if (fast_plain_lift(c, *r->level))
            for (unsigned i = 0; i < r->level->K; i++)
                v.push_back(c.coeff_modulus()[i] - c.plain_modulus()); // context.cpp:361-368
        else
            v = big_sub_small(level_product(c, *r->level), c.plain_modulus()); // Q - t as one integer (context.cpp:369-375)
        return copy_out(v, count, puhi);
        SHL_CATCH
    }
    SHL_FUNC ContextData_UpperHalfThreshold(void *thisptr, uint64_t *count, uint64_t *uht)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(count, SHL_E_POINTER);
        SHL_TRY
        auto r = as<LevelRef>(thisptr);
        if (r->ctx->scheme() != Scheme::ckks)
        {
            *count = 0; // only available for CKKS (c/contextdata.cpp:166-170)
            return SHL_S_OK;
        }
        return copy_out(big_half_up(level_product(*r->ctx, *r->level)), count, uht); // (Q + 1) >> 1 (context.cpp:408-415)
        SHL_CATCH
    }
    SHL_FUNC ContextData_UpperHalfIncrement(void *thisptr, uint64_t *count, uint64_t *uhi)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(count, SHL_E_POINTER);
        SHL_TRY
        auto r = as<LevelRef>(thisptr);
        const Context &c = *r->ctx;
        if (c.scheme() == Scheme::ckks)
        {
            *count = 0;
            return SHL_S_OK;
        }
        // (Q mod t) decomposed into the RNS base (context.cpp:339-354)
        const uint64_t rem = big_mod(level_product(c, *r->level), c.plain_modulus());
        Big v;
        for (unsigned i = 0; i < r->level->K; i++)
            v.push_back(rem % c.coeff_modulus()[i]);
        return copy_out(v, count, uhi);
        SHL_CATCH
    }
    SHL_FUNC ContextData_PrevContextData(void *thisptr, void **prev_data)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(prev_data, SHL_E_POINTER);
        auto r = as<LevelRef>(thisptr);
        // prev = one step towards the key level (chain_index + 1); the key level has none (context.h:286-293)
        *prev_data = level_handle(r->ctx, r->ctx->level_by_chain_index(r->level->chain_index + 1));
        return SHL_S_OK;
    }
    SHL_FUNC ContextData_NextContextData(void *thisptr, void **next_data)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(next_data, SHL_E_POINTER);
        auto r = as<LevelRef>(thisptr);
        *next_data = level_handle(r->ctx, r->ctx->next_level(*r->level));
        return SHL_S_OK;
    }
    SHL_FUNC ContextData_ChainIndex(void *thisptr, uint64_t *index)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(index, SHL_E_POINTER);
        *index = as<LevelRef>(thisptr)->level->chain_index;
        return SHL_S_OK;
    }
    // the level's parms_id: what a binding reads through ContextData_Parms + EncParams_GetParmsId, without the temporary
    SHL_FUNC ContextData_ParmsId(void *thisptr, uint64_t *parms_id)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(parms_id, SHL_E_POINTER);
        std::memcpy(parms_id, as<LevelRef>(thisptr)->level->parms_id, 32);
        return SHL_S_OK;
    }

    // ---- EncryptionParameterQualifiers of a ContextData (c/encryptionparameterqualifiers.h)
    SHL_FUNC EPQ_Destroy(void *thisptr)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        delete as<Qualifiers>(thisptr);
        return SHL_S_OK;
    }
#define EPQ_GET(fn, type, expr)            \
    SHL_FUNC fn(void *thisptr, type *out)  \
    {                                      \
        IfNullRet(thisptr, SHL_E_POINTER); \
        IfNullRet(out, SHL_E_POINTER);     \
        auto q = as<Qualifiers>(thisptr);  \
        *out = (expr);                     \
        return SHL_S_OK;                   \
    }
    EPQ_GET(EPQ_ParametersSet, bool, ((void)q, true))
    EPQ_GET(EPQ_UsingFFT, bool, q->using_fft)
    EPQ_GET(EPQ_UsingNTT, bool, q->using_ntt)
    EPQ_GET(EPQ_UsingBatching, bool, q->using_batching)
    EPQ_GET(EPQ_UsingFastPlainLift, bool, q->using_fast_plain_lift)
    EPQ_GET(EPQ_UsingDescendingModulusChain, bool, q->using_descending_modulus_chain)
    EPQ_GET(EPQ_SecLevel, int, q->sec_level)

    // ================================================================ EncryptionParameters (c/encryptionparameters.h)
    SHL_FUNC EncParams_Create2(void *copy, void **enc_params)
    {
        IfNullRet(copy, SHL_E_POINTER);
        IfNullRet(enc_params, SHL_E_POINTER);
        SHL_TRY
        *enc_params = new EncParams(*as<EncParams>(copy));
        SHL_CATCH
    }
    SHL_FUNC EncParams_Set(void *thisptr, void *assign)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(assign, SHL_E_POINTER);
        SHL_TRY
        *as<EncParams>(thisptr) = *as<EncParams>(assign);
        SHL_CATCH
    }
    SHL_FUNC EncParams_GetParmsId(void *thisptr, uint64_t *parms_id)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(parms_id, SHL_E_POINTER);
        SHL_TRY
        parms_id_of(*as<EncParams>(thisptr), parms_id);
        SHL_CATCH
    }
    SHL_FUNC EncParams_GetPlainModulus(void *thisptr, uint64_t *plain_modulus)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(plain_modulus, SHL_E_POINTER);
        *plain_modulus = as<EncParams>(thisptr)->plain_modulus;
        return SHL_S_OK;
    }
    SHL_FUNC EncParams_Equals(void *thisptr, void *otherptr, bool *result)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(otherptr, SHL_E_POINTER);
        IfNullRet(result, SHL_E_POINTER);
        auto a = as<EncParams>(thisptr), b = as<EncParams>(otherptr);
        // operator== compares the parms_id (encryptionparams.h:372-375), i.e. scheme, degree, primes and plain modulus
        *result = a->scheme == b->scheme && a->n == b->n && a->coeff_modulus == b->coeff_modulus && a->plain_modulus == b->plain_modulus;
        return SHL_S_OK;
    }
    SHL_FUNC EncParams_SaveSize(void *thisptr, uint8_t compr_mode, int64_t *result)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(result, SHL_E_POINTER);
        SHL_TRY
        if (!serial::compr_mode_supported(compr_mode))
            throw std::invalid_argument("unsupported compression mode");
        *result = (int64_t)serial::compress_bound(parms_raw_size(*as<EncParams>(thisptr)), compr_mode);
        SHL_CATCH
    }
    SHL_FUNC EncParams_Save(void *thisptr, uint8_t *outptr, uint64_t size, uint8_t compr_mode, int64_t *out_bytes)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(outptr, SHL_E_POINTER);
        IfNullRet(out_bytes, SHL_E_POINTER);
        SHL_TRY
        if (!serial::compr_mode_supported(compr_mode))
            throw std::invalid_argument("unsupported compression mode");
        auto p = as<EncParams>(thisptr);
        std::vector<uint8_t> raw(parms_raw_size(*p));
        parms_write(*p, raw.data());
        if (compr_mode == 0)
        {
            if (size < raw.size())
                throw std::runtime_error("I/O error");
            std::memcpy(outptr, raw.data(), raw.size());
            *out_bytes = (int64_t)raw.size();
        }
        else
            *out_bytes = (int64_t)serial::compress_stream(raw.data(), raw.size(), compr_mode, outptr, (size_t)size);
        SHL_CATCH
    }
    SHL_FUNC EncParams_Load(void *thisptr, uint8_t *inptr, uint64_t size, int64_t *in_bytes)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(inptr, SHL_E_POINTER);
        IfNullRet(in_bytes, SHL_E_POINTER);
        SHL_TRY
        uint8_t scheme = 0;
        uint64_t n = 0, t = 0;
        std::vector<uint64_t> q;
        *in_bytes = (int64_t)serial::load_encryption_parameters(inptr, (size_t)size, scheme, n, q, t);
        auto p = as<EncParams>(thisptr);
        p->scheme = scheme;
        p->n = n;
        p->coeff_modulus = q;
        p->plain_modulus = t;
        SHL_CATCH
    }

    // ================================================================ SecretKey / PublicKey (c/secretkey.h, c/publickey.h)
    SHL_FUNC SecretKey_Create2(void *copy, void **secret_key)
    {
        IfNullRet(copy, SHL_E_POINTER);
        IfNullRet(secret_key, SHL_E_POINTER);
        SHL_TRY
        auto src = as<SecretKey>(copy);
        std::unique_ptr<SecretKey> k(new SecretKey(src->context()));
        if (src->data())
            hip_ok(hipMemcpy(k->allocate(), src->data(), src->context().key_level().K * src->context().n() * 8, hipMemcpyDeviceToDevice), "copy secret key");
        *secret_key = k.release();
        SHL_CATCH
    }
    SHL_FUNC SecretKey_Assign(void *thisptr, void *assign) // sealc SecretKey_Set(thisptr, assign); SecretKey_Set here takes host words (sealhip.h)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(assign, SHL_E_POINTER);
        SHL_TRY
        auto dst = as<SecretKey>(thisptr), src = as<SecretKey>(assign);
        if (&dst->context() != &src->context())
            throw std::invalid_argument("secret keys belong to different contexts");
        if (!src->data())
            throw std::invalid_argument("secret key is not set");
        hip_ok(hipMemcpy(dst->allocate(), src->data(), src->context().key_level().K * src->context().n() * 8, hipMemcpyDeviceToDevice), "copy secret key");
        SHL_CATCH
    }
    SHL_FUNC SecretKey_ParmsId(void *thisptr, uint64_t *parms_id)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(parms_id, SHL_E_POINTER);
        auto k = as<SecretKey>(thisptr);
        if (k->data())
            std::memcpy(parms_id, k->context().key_level().parms_id, 32);
        else
            std::memset(parms_id, 0, 32); // an empty SecretKey carries parms_id_zero
        return SHL_S_OK;
    }
    SHL_FUNC SecretKey_SaveSize(void *thisptr, uint8_t compr_mode, int64_t *result)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(result, SHL_E_POINTER);
        SHL_TRY
        if (!serial::compr_mode_supported(compr_mode))
            throw std::invalid_argument("unsupported compression mode");
        auto k = as<SecretKey>(thisptr);
        const uint64_t words = k->data() ? k->context().key_level().K * k->context().n() : 0;
        *result = (int64_t)serial::compress_bound(serial::plaintext_save_size(words), compr_mode);
        SHL_CATCH
    }
    SHL_FUNC SecretKey_Save(void *thisptr, uint8_t *outptr, uint64_t size, uint8_t compr_mode, int64_t *out_bytes)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(outptr, SHL_E_POINTER);
        IfNullRet(out_bytes, SHL_E_POINTER);
        SHL_TRY
        if (!serial::compr_mode_supported(compr_mode))
            throw std::invalid_argument("unsupported compression mode");
        // SecretKey::save = its Plaintext's (secretkey.h:129-140): key-level parms_id, L*N coefficients, scale 1
        auto k = as<SecretKey>(thisptr);
        static const uint64_t zero[4] = { 0, 0, 0, 0 };
        const uint64_t words = k->data() ? k->context().key_level().K * k->context().n() : 0;
        std::vector<uint8_t> raw;
        uint8_t *dst = outptr;
        size_t cap = (size_t)size;
        if (compr_mode != 0)
        {
            raw.resize(serial::plaintext_save_size(words));
            dst = raw.data();
            cap = raw.size();
        }
        size_t data_offset = 0;
        *out_bytes = (int64_t)serial::save_plaintext(words ? k->context().key_level().parms_id : zero, words, 1.0, nullptr, dst, cap, &data_offset);
        if (words)
        {
            hip_ok(hipDeviceSynchronize(), "sync");
            hip_ok(hipMemcpy(dst + data_offset, k->data(), words * 8, hipMemcpyDeviceToHost), "D2H");
        }
        if (compr_mode != 0)
            *out_bytes = (int64_t)serial::compress_stream(raw.data(), raw.size(), compr_mode, outptr, (size_t)size);
        SHL_CATCH
    }
    SHL_FUNC PublicKey_Create2(void *copy, void **public_key)
    {
        IfNullRet(copy, SHL_E_POINTER);
        IfNullRet(public_key, SHL_E_POINTER);
        SHL_TRY
        auto src = as<PublicKey>(copy);
        std::unique_ptr<PublicKey> k(new PublicKey(src->context()));
        if (src->data())
            hip_ok(hipMemcpy(k->allocate(), src->data(), 2 * src->context().key_level().K * src->context().n() * 8, hipMemcpyDeviceToDevice), "copy public key");
        *public_key = k.release();
        SHL_CATCH
    }
    SHL_FUNC PublicKey_Assign(void *thisptr, void *assign) // sealc PublicKey_Set(thisptr, assign); PublicKey_Set here takes host words
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(assign, SHL_E_POINTER);
        SHL_TRY
        auto dst = as<PublicKey>(thisptr), src = as<PublicKey>(assign);
        if (&dst->context() != &src->context())
            throw std::invalid_argument("public keys belong to different contexts");
        if (!src->data())
            throw std::invalid_argument("public key is not set");
        hip_ok(hipMemcpy(dst->allocate(), src->data(), 2 * src->context().key_level().K * src->context().n() * 8, hipMemcpyDeviceToDevice), "copy public key");
        SHL_CATCH
    }
    SHL_FUNC PublicKey_ParmsId(void *thisptr, uint64_t *parms_id)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(parms_id, SHL_E_POINTER);
        auto k = as<PublicKey>(thisptr);
        if (k->data())
            std::memcpy(parms_id, k->context().key_level().parms_id, 32);
        else
            std::memset(parms_id, 0, 32);
        return SHL_S_OK;
    }
    SHL_FUNC PublicKey_SaveSize(void *thisptr, uint8_t compr_mode, int64_t *result)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(result, SHL_E_POINTER);
        SHL_TRY
        if (!serial::compr_mode_supported(compr_mode))
            throw std::invalid_argument("unsupported compression mode");
        auto k = as<PublicKey>(thisptr);
        const Context &c = k->context();
        *result = (int64_t)serial::compress_bound(
            k->data() ? serial::ciphertext_save_size(2, c.n(), c.key_level().K) : serial::ciphertext_save_size(0, 0, 0), compr_mode);
        SHL_CATCH
    }
    SHL_FUNC PublicKey_Save(void *thisptr, uint8_t *outptr, uint64_t size, uint8_t compr_mode, int64_t *out_bytes)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(outptr, SHL_E_POINTER);
        IfNullRet(out_bytes, SHL_E_POINTER);
        SHL_TRY
        if (!serial::compr_mode_supported(compr_mode))
            throw std::invalid_argument("unsupported compression mode");
        // PublicKey::save = its Ciphertext's (publickey.h:106-131): size 2, key level, NTT form, scale 1, correction factor 1
        auto k = as<PublicKey>(thisptr);
        const Context &c = k->context();
        static const uint64_t zero[4] = { 0, 0, 0, 0 };
        const bool set = k->data() != nullptr;
        const uint64_t K = set ? c.key_level().K : 0, n = set ? c.n() : 0, polys = set ? 2 : 0;
        const size_t raw_bytes = serial::ciphertext_save_size(polys, n, K);
        std::vector<uint8_t> raw;
        uint8_t *dst = outptr;
        size_t cap = (size_t)size;
        if (compr_mode != 0)
        {
            raw.resize(raw_bytes);
            dst = raw.data();
            cap = raw.size();
        }
        size_t data_offset = 0;
        *out_bytes = (int64_t)serial::save_ciphertext(set ? c.key_level().parms_id : zero, set, polys, n, K, 1.0, 1, nullptr, dst, cap, &data_offset);
        if (set)
        {
            hip_ok(hipDeviceSynchronize(), "sync");
            hip_ok(hipMemcpy(dst + data_offset, k->data(), polys * K * n * 8, hipMemcpyDeviceToHost), "D2H");
        }
        if (compr_mode != 0)
            *out_bytes = (int64_t)serial::compress_stream(raw.data(), raw.size(), compr_mode, outptr, (size_t)size);
        SHL_CATCH
    }
}
