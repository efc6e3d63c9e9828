// extern "C" layer of libsealhip.so — see include/sealhip.h.  Every body is closed by the same
// exception-to-HRESULT ladder as the reference's C export layer (SEAL_C_CATCH_ALL,
// native/src/seal/c/defines.h:75-97); null handles return E_POINTER like IfNullRet
// (native/src/seal/c/utilities.h).
#include "../../include/sealhip.h"
#include "evaluator.h"
#include "ckks_encoder.h"
#include "decryptor.h"
#include "keygen.h"
#include "serial.h"
#include "xof.h"
#include <atomic>
#include <cstring>
#include <new>
#include <string>

using namespace sealhip;

namespace
{
    thread_local std::string g_last_error;

    struct EncParams
    {
        uint8_t scheme = 0;
        uint64_t n = 0;
        std::vector<uint64_t> coeff_modulus;
        uint64_t plain_modulus = 0;
    };
    struct Timer
    {
        hipEvent_t e0 = nullptr, e1 = nullptr;
    };

#define SHL_TRY \
    try         \
    {
#define SHL_CATCH                               \
    }                                           \
    catch (const std::invalid_argument &e)      \
    {                                           \
        g_last_error = e.what();                \
        return SHL_E_INVALIDARG;                \
    }                                           \
    catch (const std::out_of_range &e)          \
    {                                           \
        g_last_error = e.what();                \
        return SHL_E_INVALID_INDEX;             \
    }                                           \
    catch (const std::logic_error &e)           \
    {                                           \
        g_last_error = e.what();                \
        return SHL_COR_E_INVALIDOPERATION;      \
    }                                           \
    catch (const std::runtime_error &e)         \
    {                                           \
        g_last_error = e.what();                \
        return SHL_COR_E_IO;                    \
    }                                           \
    catch (const std::bad_alloc &)              \
    {                                           \
        g_last_error = "out of device memory";  \
        return SHL_E_OUTOFMEMORY;               \
    }                                           \
    catch (...)                                 \
    {                                           \
        g_last_error = "unexpected exception";  \
        return SHL_E_UNEXPECTED;                \
    }                                           \
    return SHL_S_OK;

#define IfNullRet(p, r) \
    if (!(p))           \
    return (r)

    template <typename T>
    T *as(void *p)
    {
        return static_cast<T *>(p);
    }

    void hip_ok(hipError_t e, const char *what)
    {
        if (e != hipSuccess)
            throw std::runtime_error(std::string("HIP failure in ") + what + ": " + hipGetErrorString(e));
    }

    // dest = src unless they are the same object (the sealc "destination" convention)
    Ciphertext &prepare_dest(void *encrypted, void *destination)
    {
        Ciphertext *src = as<Ciphertext>(encrypted), *dst = as<Ciphertext>(destination);
        if (src != dst)
            *dst = *src;
        return *dst;
    }
} // namespace

extern "C"
{
    // ------------------------------------------------------------------ library / device
    SHL_FUNC SealHip_Version(uint32_t *major, uint32_t *minor, uint32_t *patch)
    {
        IfNullRet(major, SHL_E_POINTER);
        IfNullRet(minor, SHL_E_POINTER);
        IfNullRet(patch, SHL_E_POINTER);
        *major = 0;
        *minor = 1;
        *patch = 0;
        return SHL_S_OK;
    }
    SHL_FUNC SealHip_DeviceInfo(char *name, uint64_t name_capacity, int *compute_units, uint64_t *hbm_bytes)
    {
        SHL_TRY
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
            throw std::runtime_error("no HIP device visible: libsealhip has no CPU fallback");
        int dev = 0;
        hip_ok(hipGetDevice(&dev), "hipGetDevice");
        hipDeviceProp_t prop;
        hip_ok(hipGetDeviceProperties(&prop, dev), "hipGetDeviceProperties");
        if (name && name_capacity)
        {
            std::strncpy(name, prop.name, name_capacity - 1);
            name[name_capacity - 1] = 0;
        }
        if (compute_units)
            *compute_units = prop.multiProcessorCount;
        if (hbm_bytes)
            *hbm_bytes = prop.totalGlobalMem;
        SHL_CATCH
    }
    SHL_FUNC SealHip_LastError(char *outstr, uint64_t *length)
    {
        IfNullRet(length, SHL_E_POINTER);
        if (outstr && *length > g_last_error.size())
            std::memcpy(outstr, g_last_error.c_str(), g_last_error.size() + 1);
        *length = g_last_error.size() + 1;
        return SHL_S_OK;
    }

    // ------------------------------------------------------------------ parameter helpers
    SHL_FUNC CoeffModulus_Create1(uint64_t poly_modulus_degree, uint64_t length, int *bit_sizes, uint64_t *coeffs)
    {
        IfNullRet(bit_sizes, SHL_E_POINTER);
        IfNullRet(coeffs, SHL_E_POINTER);
        SHL_TRY
        if (poly_modulus_degree < 2 || poly_modulus_degree > 131072 || (poly_modulus_degree & (poly_modulus_degree - 1)))
            throw std::invalid_argument("poly_modulus_degree is invalid");
        std::vector<int> bits(bit_sizes, bit_sizes + length);
        auto v = host::coeff_modulus_create(poly_modulus_degree, bits);
        for (size_t i = 0; i < v.size(); i++)
            coeffs[i] = v[i];
        SHL_CATCH
    }
    SHL_FUNC PlainModulus_Batching(uint64_t poly_modulus_degree, int bit_size, uint64_t *value)
    {
        IfNullRet(value, SHL_E_POINTER);
        SHL_TRY
        *value = host::plain_modulus_batching(poly_modulus_degree, bit_size);
        SHL_CATCH
    }

    SHL_FUNC EncParams_Create1(uint8_t scheme, void **enc_params)
    {
        IfNullRet(enc_params, SHL_E_POINTER);
        SHL_TRY
        if (scheme > 3)
            throw std::invalid_argument("unsupported scheme");
        auto p = new EncParams();
        p->scheme = scheme;
        *enc_params = p;
        SHL_CATCH
    }
    SHL_FUNC EncParams_Destroy(void *thisptr)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        delete as<EncParams>(thisptr);
        return SHL_S_OK;
    }
    SHL_FUNC EncParams_SetPolyModulusDegree(void *thisptr, uint64_t degree)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        as<EncParams>(thisptr)->n = degree;
        return SHL_S_OK;
    }
    SHL_FUNC EncParams_GetPolyModulusDegree(void *thisptr, uint64_t *degree)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(degree, SHL_E_POINTER);
        *degree = as<EncParams>(thisptr)->n;
        return SHL_S_OK;
    }
    SHL_FUNC EncParams_SetCoeffModulus(void *thisptr, uint64_t length, const uint64_t *coeffs)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(coeffs, SHL_E_POINTER);
        SHL_TRY
        if (length < 1 || length > kMaxComps)
            throw std::invalid_argument("coeff_modulus is invalid");
        as<EncParams>(thisptr)->coeff_modulus.assign(coeffs, coeffs + length);
        SHL_CATCH
    }
    SHL_FUNC EncParams_GetCoeffModulus(void *thisptr, uint64_t *length, uint64_t *coeffs)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(length, SHL_E_POINTER);
        auto &v = as<EncParams>(thisptr)->coeff_modulus;
        *length = v.size();
        if (coeffs)
            std::memcpy(coeffs, v.data(), v.size() * 8);
        return SHL_S_OK;
    }
    SHL_FUNC EncParams_SetPlainModulus2(void *thisptr, uint64_t plain_modulus)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        SHL_TRY
        auto p = as<EncParams>(thisptr);
        // EncryptionParameters::set_plain_modulus (encryptionparams.h): CKKS takes none
        if (p->scheme == 2 && plain_modulus != 0)
            throw std::logic_error("plain_modulus is not supported for this scheme");
        p->plain_modulus = plain_modulus;
        SHL_CATCH
    }
    SHL_FUNC EncParams_GetScheme(void *thisptr, uint8_t *scheme)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(scheme, SHL_E_POINTER);
        *scheme = as<EncParams>(thisptr)->scheme;
        return SHL_S_OK;
    }

    // ------------------------------------------------------------------ SEALContext
    SHL_FUNC SEALContext_Create(void *encryptionParams, bool expand_mod_chain, int sec_level, void **context)
    {
        (void)sec_level;
        IfNullRet(encryptionParams, SHL_E_POINTER);
        IfNullRet(context, SHL_E_POINTER);
        SHL_TRY
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
            throw std::runtime_error("no HIP device visible: libsealhip has no CPU fallback");
        auto p = as<EncParams>(encryptionParams);
        *context = new Context(static_cast<Scheme>(p->scheme), p->n, p->coeff_modulus, p->plain_modulus, expand_mod_chain);
        SHL_CATCH
    }
    SHL_FUNC SEALContext_Destroy(void *thisptr)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        delete as<Context>(thisptr);
        return SHL_S_OK;
    }
    SHL_FUNC SEALContext_KeyParmsId(void *thisptr, uint64_t *parms_id)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(parms_id, SHL_E_POINTER);
        std::memcpy(parms_id, as<Context>(thisptr)->key_level().parms_id, 32);
        return SHL_S_OK;
    }
    SHL_FUNC SEALContext_FirstParmsId(void *thisptr, uint64_t *parms_id)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(parms_id, SHL_E_POINTER);
        std::memcpy(parms_id, as<Context>(thisptr)->first_level().parms_id, 32);
        return SHL_S_OK;
    }
    SHL_FUNC SEALContext_LastParmsId(void *thisptr, uint64_t *parms_id)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(parms_id, SHL_E_POINTER);
        std::memcpy(parms_id, as<Context>(thisptr)->last_level().parms_id, 32);
        return SHL_S_OK;
    }
    SHL_FUNC SEALContext_UsingKeyswitching(void *thisptr, bool *using_keyswitching)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(using_keyswitching, SHL_E_POINTER);
        *using_keyswitching = as<Context>(thisptr)->using_keyswitching();
        return SHL_S_OK;
    }
    SHL_FUNC SEALContext_ChainIndex(void *thisptr, uint64_t *parms_id, uint64_t *chain_index)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(parms_id, SHL_E_POINTER);
        IfNullRet(chain_index, SHL_E_POINTER);
        SHL_TRY
        auto l = as<Context>(thisptr)->level_by_parms_id(parms_id);
        if (!l)
            throw std::invalid_argument("parms_id is not valid for encryption parameters");
        *chain_index = l->chain_index;
        SHL_CATCH
    }
    SHL_FUNC SEALContext_ParmsIdAt(void *thisptr, uint64_t chain_index, uint64_t *parms_id)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(parms_id, SHL_E_POINTER);
        SHL_TRY
        auto l = as<Context>(thisptr)->level_by_chain_index(chain_index);
        if (!l)
            throw std::out_of_range("chain_index");
        std::memcpy(parms_id, l->parms_id, 32);
        SHL_CATCH
    }
    SHL_FUNC SEALContext_CoeffModulusAt(void *thisptr, uint64_t chain_index, uint64_t *length, uint64_t *coeffs)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(length, SHL_E_POINTER);
        SHL_TRY
        auto c = as<Context>(thisptr);
        auto l = c->level_by_chain_index(chain_index);
        if (!l)
            throw std::out_of_range("chain_index");
        *length = l->K;
        if (coeffs)
            std::memcpy(coeffs, c->coeff_modulus().data(), l->K * 8);
        SHL_CATCH
    }
    SHL_FUNC SEALContext_TotalCoeffModulusBitCount(void *thisptr, uint64_t chain_index, int *bit_count)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(bit_count, SHL_E_POINTER);
        SHL_TRY
        auto l = as<Context>(thisptr)->level_by_chain_index(chain_index);
        if (!l)
            throw std::out_of_range("chain_index");
        *bit_count = l->total_coeff_modulus_bit_count;
        SHL_CATCH
    }
    SHL_FUNC SEALContext_SetParmsId(void *thisptr, uint64_t chain_index, uint64_t *parms_id)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(parms_id, SHL_E_POINTER);
        SHL_TRY
        auto c = as<Context>(thisptr);
        if (!c->level_by_chain_index(chain_index))
            throw std::out_of_range("chain_index");
        c->set_parms_id(chain_index, parms_id);
        SHL_CATCH
    }
    SHL_FUNC SEALContext_NTTRoot(void *thisptr, uint64_t prime_index, uint64_t *root)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(root, SHL_E_POINTER);
        SHL_TRY
        auto c = as<Context>(thisptr);
        if (prime_index >= c->pool_primes().size())
            throw std::out_of_range("prime_index");
        *root = c->ntt_root((unsigned)prime_index);
        SHL_CATCH
    }
    SHL_FUNC SEALContext_BaseBsk(void *thisptr, uint64_t chain_index, uint64_t *length, uint64_t *primes)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(length, SHL_E_POINTER);
        SHL_TRY
        auto l = as<Context>(thisptr)->level_by_chain_index(chain_index);
        if (!l)
            throw std::out_of_range("chain_index");
        *length = l->bsk.size();
        if (primes)
            std::memcpy(primes, l->bsk.data(), l->bsk.size() * 8);
        SHL_CATCH
    }

    // ------------------------------------------------------------------ Ciphertext
    SHL_FUNC Ciphertext_Create3(void *context, void *pool, void **cipher)
    {
        (void)pool;
        IfNullRet(context, SHL_E_POINTER);
        IfNullRet(cipher, SHL_E_POINTER);
        SHL_TRY
        *cipher = new Ciphertext(*as<Context>(context), 1);
        SHL_CATCH
    }
    SHL_FUNC Ciphertext_CreateBatch(void *context, uint64_t batch, void **cipher)
    {
        IfNullRet(context, SHL_E_POINTER);
        IfNullRet(cipher, SHL_E_POINTER);
        SHL_TRY
        if (batch == 0)
            throw std::invalid_argument("batch must be positive");
        *cipher = new Ciphertext(*as<Context>(context), batch);
        SHL_CATCH
    }
    SHL_FUNC Ciphertext_Create2(void *copy, void **cipher)
    {
        IfNullRet(copy, SHL_E_POINTER);
        IfNullRet(cipher, SHL_E_POINTER);
        SHL_TRY
        hip_ok(hipDeviceSynchronize(), "sync");
        *cipher = new Ciphertext(*as<Ciphertext>(copy));
        hip_ok(hipDeviceSynchronize(), "sync");
        SHL_CATCH
    }
    SHL_FUNC Ciphertext_Set(void *thisptr, void *assign)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(assign, SHL_E_POINTER);
        SHL_TRY
        hip_ok(hipDeviceSynchronize(), "sync");
        *as<Ciphertext>(thisptr) = *as<Ciphertext>(assign);
        hip_ok(hipDeviceSynchronize(), "sync");
        SHL_CATCH
    }
    SHL_FUNC Ciphertext_Destroy(void *thisptr)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        delete as<Ciphertext>(thisptr);
        return SHL_S_OK;
    }
    SHL_FUNC Ciphertext_Resize1(void *thisptr, void *context, uint64_t *parms_id, uint64_t size)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(context, SHL_E_POINTER);
        IfNullRet(parms_id, SHL_E_POINTER);
        SHL_TRY
        auto ct = as<Ciphertext>(thisptr);
        auto c = as<Context>(context);
        if (&ct->context() != c)
            throw std::invalid_argument("ciphertext belongs to another context");
        ct->resize(c->level_by_parms_id(parms_id), size, nullptr);
        SHL_CATCH
    }
#define CT_GET(fn, type, expr)                      \
    SHL_FUNC fn(void *thisptr, type *out)           \
    {                                               \
        IfNullRet(thisptr, SHL_E_POINTER);          \
        IfNullRet(out, SHL_E_POINTER);              \
        auto ct = as<Ciphertext>(thisptr);          \
        *out = (expr);                              \
        return SHL_S_OK;                            \
    }
    CT_GET(Ciphertext_Size, uint64_t, ct->size())
    CT_GET(Ciphertext_BatchCount, uint64_t, ct->batch())
    CT_GET(Ciphertext_PolyModulusDegree, uint64_t, ct->poly_modulus_degree())
    CT_GET(Ciphertext_CoeffModulusSize, uint64_t, ct->coeff_modulus_size())
    CT_GET(Ciphertext_IsNTTForm, bool, ct->is_ntt_form())
    CT_GET(Ciphertext_Scale, double, ct->scale())
    CT_GET(Ciphertext_CorrectionFactor, uint64_t, ct->correction_factor())
    SHL_FUNC Ciphertext_ParmsId(void *thisptr, uint64_t *parms_id)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(parms_id, SHL_E_POINTER);
        auto ct = as<Ciphertext>(thisptr);
        if (ct->level())
            std::memcpy(parms_id, ct->level()->parms_id, 32);
        else
            std::memset(parms_id, 0, 32); // parms_id_zero
        return SHL_S_OK;
    }
    SHL_FUNC Ciphertext_SetIsNTTForm(void *thisptr, bool is_ntt_form)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        as<Ciphertext>(thisptr)->is_ntt_form() = is_ntt_form;
        return SHL_S_OK;
    }
    SHL_FUNC Ciphertext_SetScale(void *thisptr, double scale)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        as<Ciphertext>(thisptr)->scale() = scale;
        return SHL_S_OK;
    }
    SHL_FUNC Ciphertext_SetCorrectionFactor(void *thisptr, uint64_t correction_factor)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        as<Ciphertext>(thisptr)->correction_factor() = correction_factor;
        return SHL_S_OK;
    }
    SHL_FUNC Ciphertext_IsTransparent(void *thisptr, bool *result)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(result, SHL_E_POINTER);
        SHL_TRY
        auto ct = as<Ciphertext>(thisptr);
        Evaluator ev(ct->context());
        hip_ok(hipDeviceSynchronize(), "sync");
        *result = ev.is_transparent(*ct);
        SHL_CATCH
    }
    SHL_FUNC Ciphertext_DevicePtr(void *thisptr, uint64_t **data, uint64_t *word_count)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(data, SHL_E_POINTER);
        auto ct = as<Ciphertext>(thisptr);
        *data = ct->data();
        if (word_count)
            *word_count = ct->word_count();
        return SHL_S_OK;
    }
    SHL_FUNC Ciphertext_CopyFromHost(void *thisptr, const uint64_t *src, uint64_t word_count)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(src, SHL_E_POINTER);
        SHL_TRY
        auto ct = as<Ciphertext>(thisptr);
        if (word_count != ct->word_count())
            throw std::invalid_argument("word_count does not match the ciphertext slab");
        hip_ok(hipDeviceSynchronize(), "sync");
        copy_h2d(ct->data(), src, word_count * 8);
        SHL_CATCH
    }
    SHL_FUNC Ciphertext_CopyToHost(void *thisptr, uint64_t *dst, uint64_t word_count)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(dst, SHL_E_POINTER);
        SHL_TRY
        auto ct = as<Ciphertext>(thisptr);
        if (word_count != ct->word_count())
            throw std::invalid_argument("word_count does not match the ciphertext slab");
        hip_ok(hipDeviceSynchronize(), "sync");
        copy_d2h(dst, ct->data(), word_count * 8);
        SHL_CATCH
    }
    SHL_FUNC Ciphertext_CopyWordsToHost(void *thisptr, uint64_t word_offset, uint64_t word_count, uint64_t *dst)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(dst, SHL_E_POINTER);
        SHL_TRY
        auto ct = as<Ciphertext>(thisptr);
        if (word_offset > ct->word_count() || word_count > ct->word_count() - word_offset)
            throw std::invalid_argument("word range outside the ciphertext slab");
        hip_ok(hipDeviceSynchronize(), "sync");
        if (word_count)
            copy_d2h(dst, ct->data() + word_offset, word_count * 8);
        SHL_CATCH
    }
    SHL_FUNC Ciphertext_CopyFromDevice(void *thisptr, const uint64_t *src, uint64_t word_count, void *hip_stream)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(src, SHL_E_POINTER);
        SHL_TRY
        auto ct = as<Ciphertext>(thisptr);
        if (word_count != ct->word_count())
            throw std::invalid_argument("word_count does not match the ciphertext slab");
        hip_ok(hipMemcpyAsync(ct->data(), src, word_count * 8, hipMemcpyDeviceToDevice, (hipStream_t)hip_stream), "D2D");
        SHL_CATCH
    }

    // ---- wire format (native/src/seal/c/ciphertext.h:80-86; seal_amd/csrc/serial.h)
    namespace
    {
        // one host image -> batch slot `item` of a device-resident batch.  set_meta: the image defines the batch's metadata
        // (first item / batch of one); otherwise it has to agree with the items already there.
        void upload_image(Ciphertext &ct, const Context &c, serial::CiphertextImage &img, size_t item, bool set_meta)
        {
            if (&ct.context() != &c)
                throw std::invalid_argument("ciphertext belongs to another context");
            if (item >= ct.batch())
                throw std::out_of_range("batch item");
            // BGV ciphertexts are serialized in coefficient form and transformed on load (ciphertext.cpp:384-403)
            const bool to_ntt = c.scheme() == Scheme::bgv && !img.is_ntt_form && img.word_count() != 0;
            const bool ntt_form = img.is_ntt_form || to_ntt;
            hip_ok(hipDeviceSynchronize(), "sync");
            if (set_meta)
            {
                ct.resize(img.level, (size_t)img.size, nullptr);
                ct.is_ntt_form() = ntt_form;
                ct.scale() = img.scale;
                ct.correction_factor() = img.correction_factor;
            }
            else if (ct.level() != img.level || ct.size() != img.size || ct.is_ntt_form() != ntt_form || ct.scale() != img.scale ||
                     ct.correction_factor() != img.correction_factor)
                throw std::invalid_argument("serialized ciphertext does not match the metadata of the batch");
            if (img.word_count() == 0)
                return;
            const size_t n = c.n(), K = img.level->K, poly_words = K * n;
            uint64_t *tmp = nullptr;
            hip_ok(hipMalloc(reinterpret_cast<void **>(&tmp), img.word_count() * 8), "hipMalloc");
            // the stored piece goes to the device straight from the caller's stream buffer
            hipError_t e = img.stored_words ? hipMemcpy(tmp, img.stored, img.stored_words * 8, hipMemcpyHostToDevice) : hipSuccess;
            if (e == hipSuccess && !img.expanded.empty())
                e = hipMemcpy(tmp + img.stored_words, img.expanded.data(), img.expanded.size() * 8, hipMemcpyHostToDevice);
            if (e == hipSuccess && img.pending_words)
            {
                // the seeded c_1: sample_poly_uniform over Blake2xb on the device (xof.h)
                XofJob job;
                std::memcpy(job.seed, img.pending_seed, sizeof(job.seed));
                job.prng_type = img.pending_type;
                job.dst = tmp + img.stored_words;
                try
                {
                    sample_uniform_device(c, K, { job });
                }
                catch (...)
                {
                    (void)hipFree(tmp);
                    throw;
                }
            }
            if (e == hipSuccess && to_ntt)
            {
                NttBatch b{};
                b.data = tmp;
                b.outer_stride = poly_words;
                b.ncomp = (unsigned)K;
                b.nouter = (unsigned)img.size;
                b.prime_first = 0;
                e = ntt_forward(c.ntt_tables(), b, 0, nullptr);
            }
            for (size_t p = 0; e == hipSuccess && p < img.size; p++)
                e = hipMemcpyAsync(ct.plane(p) + item * poly_words, tmp + p * poly_words, poly_words * 8, hipMemcpyDeviceToDevice, nullptr);
            if (e == hipSuccess)
                e = hipDeviceSynchronize();
            (void)hipFree(tmp);
            hip_ok(e, "ciphertext upload");
        }
        SHL_HRESULT ct_load(void *thisptr, void *context, uint64_t item, bool whole, uint8_t *inptr, uint64_t size, int64_t *in_bytes, bool check)
        {
            IfNullRet(thisptr, SHL_E_POINTER);
            IfNullRet(context, SHL_E_POINTER);
            IfNullRet(inptr, SHL_E_POINTER);
            IfNullRet(in_bytes, SHL_E_POINTER);
            SHL_TRY
            auto ct = as<Ciphertext>(thisptr);
            auto c = as<Context>(context);
            if (whole && ct->batch() != 1)
                throw std::invalid_argument("Ciphertext_Load needs a batch of one: use Ciphertext_LoadItem for a slot of a batch");
            serial::CiphertextImage img;
            *in_bytes = (int64_t)serial::load_ciphertext(*c, inptr, (size_t)size, check, img, true);
            // the first item loaded into an empty batch defines its metadata
            upload_image(*ct, *c, img, (size_t)item, whole || ct->size() == 0);
            SHL_CATCH
        }
        SHL_HRESULT ks_load(void *thisptr, void *context, uint8_t *inptr, uint64_t size, int64_t *in_bytes, bool check)
        {
            IfNullRet(thisptr, SHL_E_POINTER);
            IfNullRet(context, SHL_E_POINTER);
            IfNullRet(inptr, SHL_E_POINTER);
            IfNullRet(in_bytes, SHL_E_POINTER);
            SHL_TRY
            auto c = as<Context>(context);
            serial::KSwitchKeysImage img;
            *in_bytes = (int64_t)serial::load_kswitchkeys(*c, inptr, (size_t)size, check, img, true);
            auto keys = as<KSwitchKeys>(thisptr);
            keys->clear(); // the loaded object replaces the previous contents, indices absent from the stream included
            for (size_t index = 0; index < img.keys.size(); index++)
            {
                auto &digits = img.keys[index];
                if (digits.empty())
                    continue;
                // [digit][2][L][N], the layout of KSwitchKeys::keys_[index][digit].data() (kswitchkeys.h:340); every piece is
                // copied to the device from where it lies (the stream buffer / the expanded c_1): no host staging copy
                keys->set_key_with(*c, index, digits.size(), [&](uint64_t *dst) {
                    std::vector<XofJob> seeded; // the c_1 halves a Blake2xb seed stands for: expanded on the device, all digits at once
                    for (auto &d : digits)
                    {
                        if (d.stored_words)
                            hip_ok(hipMemcpy(dst, d.stored, d.stored_words * 8, hipMemcpyHostToDevice), "upload key");
                        if (!d.expanded.empty())
                            hip_ok(hipMemcpy(dst + d.stored_words, d.expanded.data(), d.expanded.size() * 8, hipMemcpyHostToDevice), "upload key");
                        if (d.pending_words)
                        {
                            XofJob job;
                            std::memcpy(job.seed, d.pending_seed, sizeof(job.seed));
                            job.prng_type = d.pending_type;
                            job.dst = dst + d.stored_words;
                            seeded.push_back(job);
                        }
                        dst += d.word_count();
                    }
                    sample_uniform_device(*c, c->key_level().K, seeded);
                });
                for (auto &d : digits)
                    std::vector<uint64_t>().swap(d.expanded);
            }
            SHL_CATCH
        }
    } // namespace
    SHL_FUNC Ciphertext_Load(void *thisptr, void *context, uint8_t *inptr, uint64_t size, int64_t *in_bytes)
    {
        return ct_load(thisptr, context, 0, true, inptr, size, in_bytes, true);
    }
    SHL_FUNC Ciphertext_UnsafeLoad(void *thisptr, void *context, uint8_t *inptr, uint64_t size, int64_t *in_bytes)
    {
        return ct_load(thisptr, context, 0, true, inptr, size, in_bytes, false);
    }
    SHL_FUNC Ciphertext_LoadItem(void *thisptr, void *context, uint64_t item, uint8_t *inptr, uint64_t size, int64_t *in_bytes)
    {
        return ct_load(thisptr, context, item, false, inptr, size, in_bytes, true);
    }
    SHL_FUNC Ciphertext_SaveSize(void *thisptr, uint8_t compr_mode, int64_t *result)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(result, SHL_E_POINTER);
        SHL_TRY
        auto ct = as<Ciphertext>(thisptr);
        *result = (int64_t)serial::compress_bound(
            serial::ciphertext_save_size(ct->size(), ct->poly_modulus_degree(), ct->coeff_modulus_size()), compr_mode);
        SHL_CATCH
    }
    SHL_FUNC Ciphertext_SaveItem(void *thisptr, uint64_t item, uint8_t *outptr, uint64_t size, uint8_t compr_mode, int64_t *out_bytes)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(outptr, SHL_E_POINTER);
        IfNullRet(out_bytes, SHL_E_POINTER);
        SHL_TRY
        if (!serial::compr_mode_supported(compr_mode))
            throw std::invalid_argument("unsupported compression mode");
        auto ct = as<Ciphertext>(thisptr);
        if (item >= ct->batch())
            throw std::out_of_range("batch item");
        const size_t poly_words = ct->coeff_modulus_size() * ct->poly_modulus_degree();
        static const uint64_t zero_id[4] = { 0, 0, 0, 0 };
        // uncompressed: straight into the caller's buffer; compressed: through a host image of the raw stream
        std::vector<uint8_t> raw;
        uint8_t *dst = outptr;
        size_t cap = (size_t)size;
        if (compr_mode != 0)
        {
            raw.resize(serial::ciphertext_save_size(ct->size(), ct->poly_modulus_degree(), ct->coeff_modulus_size()));
            dst = raw.data();
            cap = raw.size();
        }
        size_t data_offset = 0;
        *out_bytes = (int64_t)serial::save_ciphertext(
            ct->level() ? ct->level()->parms_id : zero_id, ct->is_ntt_form(), ct->size(), ct->poly_modulus_degree(),
            ct->coeff_modulus_size(), ct->scale(), ct->correction_factor(), nullptr, dst, cap, &data_offset);
        // the coefficient words go from the device slab straight into the stream
        hip_ok(hipDeviceSynchronize(), "sync");
        for (size_t p = 0; p < ct->size(); p++)
            hip_ok(hipMemcpy(dst + data_offset + p * poly_words * 8, ct->plane(p) + item * poly_words, poly_words * 8, hipMemcpyDeviceToHost), "D2H");
        if (compr_mode != 0)
            *out_bytes = (int64_t)serial::compress_stream(raw.data(), raw.size(), compr_mode, outptr, (size_t)size);
        SHL_CATCH
    }
    SHL_FUNC Ciphertext_Save(void *thisptr, uint8_t *outptr, uint64_t size, uint8_t compr_mode, int64_t *out_bytes)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        if (as<Ciphertext>(thisptr)->batch() != 1)
        {
            g_last_error = "Ciphertext_Save needs a batch of one: use Ciphertext_SaveItem for a slot of a batch";
            return SHL_E_INVALIDARG;
        }
        return Ciphertext_SaveItem(thisptr, 0, outptr, size, compr_mode, out_bytes);
    }
    namespace
    {
        SHL_HRESULT pt_load(void *thisptr, void *context, uint8_t *inptr, uint64_t size, int64_t *in_bytes, bool check)
        {
            IfNullRet(thisptr, SHL_E_POINTER);
            IfNullRet(context, SHL_E_POINTER);
            IfNullRet(inptr, SHL_E_POINTER);
            IfNullRet(in_bytes, SHL_E_POINTER);
            SHL_TRY
            auto pt = as<Plaintext>(thisptr);
            auto c = as<Context>(context);
            if (&pt->context() != c)
                throw std::invalid_argument("plaintext belongs to another context");
            serial::PlaintextImage img;
            *in_bytes = (int64_t)serial::load_plaintext(*c, inptr, (size_t)size, check, img);
            hip_ok(hipDeviceSynchronize(), "sync");
            pt->set(reinterpret_cast<const uint64_t *>(img.stored), (size_t)img.coeff_count, false); // H2D straight from the stream
            pt->set_level(img.level);
            pt->scale() = img.scale;
            SHL_CATCH
        }
    } // namespace
    SHL_FUNC Plaintext_Load(void *thisptr, void *context, uint8_t *inptr, uint64_t size, int64_t *in_bytes)
    {
        return pt_load(thisptr, context, inptr, size, in_bytes, true);
    }
    SHL_FUNC Plaintext_UnsafeLoad(void *thisptr, void *context, uint8_t *inptr, uint64_t size, int64_t *in_bytes)
    {
        return pt_load(thisptr, context, inptr, size, in_bytes, false);
    }
    SHL_FUNC Plaintext_SaveSize(void *thisptr, uint8_t compr_mode, int64_t *result)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(result, SHL_E_POINTER);
        SHL_TRY
        *result = (int64_t)serial::compress_bound(serial::plaintext_save_size(as<Plaintext>(thisptr)->coeff_count()), compr_mode);
        SHL_CATCH
    }
    SHL_FUNC Plaintext_Save(void *thisptr, uint8_t *outptr, uint64_t size, uint8_t compr_mode, int64_t *out_bytes)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(outptr, SHL_E_POINTER);
        IfNullRet(out_bytes, SHL_E_POINTER);
        SHL_TRY
        if (!serial::compr_mode_supported(compr_mode))
            throw std::invalid_argument("unsupported compression mode");
        auto pt = as<Plaintext>(thisptr);
        static const uint64_t zero_id[4] = { 0, 0, 0, 0 };
        std::vector<uint8_t> raw;
        uint8_t *dst = outptr;
        size_t cap = (size_t)size;
        if (compr_mode != 0)
        {
            raw.resize(serial::plaintext_save_size(pt->coeff_count()));
            dst = raw.data();
            cap = raw.size();
        }
        size_t data_offset = 0;
        *out_bytes = (int64_t)serial::save_plaintext(pt->level() ? pt->level()->parms_id : zero_id, pt->coeff_count(), pt->scale(), nullptr,
                                                     dst, cap, &data_offset);
        hip_ok(hipDeviceSynchronize(), "sync");
        if (pt->coeff_count())
            hip_ok(hipMemcpy(dst + data_offset, pt->data(), pt->coeff_count() * 8, hipMemcpyDeviceToHost), "D2H");
        if (compr_mode != 0)
            *out_bytes = (int64_t)serial::compress_stream(raw.data(), raw.size(), compr_mode, outptr, (size_t)size);
        SHL_CATCH
    }
    SHL_FUNC KSwitchKeys_Load(void *thisptr, void *context, uint8_t *inptr, uint64_t size, int64_t *in_bytes)
    {
        return ks_load(thisptr, context, inptr, size, in_bytes, true);
    }
    SHL_FUNC KSwitchKeys_UnsafeLoad(void *thisptr, void *context, uint8_t *inptr, uint64_t size, int64_t *in_bytes)
    {
        return ks_load(thisptr, context, inptr, size, in_bytes, false);
    }

    // ------------------------------------------------------------------ SecretKey / Decryptor (native/src/seal/c/secretkey.h, decryptor.h)
    SHL_FUNC SecretKey_Create(void *context, void **secret_key)
    {
        IfNullRet(context, SHL_E_POINTER);
        IfNullRet(secret_key, SHL_E_POINTER);
        SHL_TRY
        *secret_key = new SecretKey(*as<Context>(context));
        SHL_CATCH
    }
    SHL_FUNC SecretKey_Destroy(void *thisptr)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        delete as<SecretKey>(thisptr);
        return SHL_S_OK;
    }
    // ---- KeyGenerator (native/src/seal/c/keygenerator.h; keygen.h)
    SHL_FUNC KeyGenerator_Create1(void *context, const uint64_t *seed8, void **key_generator)
    {
        IfNullRet(context, SHL_E_POINTER);
        IfNullRet(key_generator, SHL_E_POINTER);
        SHL_TRY
        *key_generator = new KeyGenerator(*as<Context>(context), seed8);
        SHL_CATCH
    }
    SHL_FUNC KeyGenerator_Create2(void *context, void *secret_key, const uint64_t *seed8, void **key_generator)
    {
        IfNullRet(context, SHL_E_POINTER);
        IfNullRet(secret_key, SHL_E_POINTER);
        IfNullRet(key_generator, SHL_E_POINTER);
        SHL_TRY
        *key_generator = new KeyGenerator(*as<Context>(context), *as<SecretKey>(secret_key), seed8);
        SHL_CATCH
    }
    SHL_FUNC KeyGenerator_Destroy(void *thisptr)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        delete as<KeyGenerator>(thisptr);
        return SHL_S_OK;
    }
    SHL_FUNC KeyGenerator_SecretKey(void *thisptr, void *secret_key)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(secret_key, SHL_E_POINTER);
        SHL_TRY
        auto kg = as<KeyGenerator>(thisptr);
        auto dst = as<SecretKey>(secret_key);
        if (&dst->context() != &kg->secret_key().context())
            throw std::invalid_argument("secret key belongs to another context");
        const Context &c = dst->context();
        hip_ok(hipMemcpy(dst->allocate(), kg->secret_key().data(), c.key_level().K * c.n() * 8, hipMemcpyDeviceToDevice), "copy secret key");
        SHL_CATCH
    }
    SHL_FUNC KeyGenerator_CreatePublicKey(void *thisptr, void *public_key)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(public_key, SHL_E_POINTER);
        SHL_TRY
        as<KeyGenerator>(thisptr)->create_public_key(*as<PublicKey>(public_key));
        SHL_CATCH
    }
    SHL_FUNC KeyGenerator_CreateRelinKeys(void *thisptr, void *relin_keys)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(relin_keys, SHL_E_POINTER);
        SHL_TRY
        as<KeyGenerator>(thisptr)->create_relin_keys(*as<KSwitchKeys>(relin_keys));
        SHL_CATCH
    }
    SHL_FUNC KeyGenerator_CreateGaloisKeysFromElts(void *thisptr, uint64_t count, const uint32_t *galois_elts, void *galois_keys)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(galois_keys, SHL_E_POINTER);
        SHL_TRY
        as<KeyGenerator>(thisptr)->create_galois_keys(galois_elts, (size_t)count, *as<KSwitchKeys>(galois_keys));
        SHL_CATCH
    }
    SHL_FUNC KeyGenerator_CreateGaloisKeysFromSteps(void *thisptr, uint64_t count, const int *steps, void *galois_keys)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(galois_keys, SHL_E_POINTER);
        SHL_TRY
        as<KeyGenerator>(thisptr)->create_galois_keys_from_steps(steps, (size_t)count, *as<KSwitchKeys>(galois_keys));
        SHL_CATCH
    }
    SHL_FUNC KeyGenerator_CreateGaloisKeysAll(void *thisptr, void *galois_keys)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(galois_keys, SHL_E_POINTER);
        SHL_TRY
        as<KeyGenerator>(thisptr)->create_galois_keys_all(*as<KSwitchKeys>(galois_keys));
        SHL_CATCH
    }
    SHL_FUNC KeyGenerator_SeededSaveSize(void *thisptr, bool galois, uint64_t key_count, int64_t *result)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(result, SHL_E_POINTER);
        SHL_TRY
        *result = (int64_t)as<KeyGenerator>(thisptr)->seeded_save_size(galois, (size_t)key_count);
        SHL_CATCH
    }
    SHL_FUNC KeyGenerator_CreateRelinKeysSave(void *thisptr, uint8_t *outptr, uint64_t size, int64_t *out_bytes)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(outptr, SHL_E_POINTER);
        IfNullRet(out_bytes, SHL_E_POINTER);
        SHL_TRY
        *out_bytes = (int64_t)as<KeyGenerator>(thisptr)->save_seeded(false, nullptr, 0, outptr, (size_t)size);
        SHL_CATCH
    }
    SHL_FUNC KeyGenerator_CreateGaloisKeysFromEltsSave(void *thisptr, uint64_t count, const uint32_t *galois_elts, uint8_t *outptr,
                                                       uint64_t size, int64_t *out_bytes)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(outptr, SHL_E_POINTER);
        IfNullRet(out_bytes, SHL_E_POINTER);
        SHL_TRY
        *out_bytes = (int64_t)as<KeyGenerator>(thisptr)->save_seeded(true, galois_elts, (size_t)count, outptr, (size_t)size);
        SHL_CATCH
    }
    SHL_FUNC KeyGenerator_KeyToHost(void *thisptr, uint32_t galois_elt, uint64_t *host_words, uint64_t capacity_words)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(host_words, SHL_E_POINTER);
        SHL_TRY
        auto kg = as<KeyGenerator>(thisptr);
        if (capacity_words < kg->key_words())
            throw std::invalid_argument("capacity");
        kg->key_to_host(galois_elt, host_words);
        SHL_CATCH
    }
    SHL_FUNC SecretKey_Get(void *thisptr, uint64_t *host_words)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(host_words, SHL_E_POINTER);
        SHL_TRY
        as<SecretKey>(thisptr)->get(host_words);
        SHL_CATCH
    }
    SHL_FUNC PublicKey_Get(void *thisptr, uint64_t *host_words)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(host_words, SHL_E_POINTER);
        SHL_TRY
        as<PublicKey>(thisptr)->get(host_words);
        SHL_CATCH
    }
    SHL_FUNC SecretKey_Set(void *thisptr, const uint64_t *host_words, uint64_t word_count)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(host_words, SHL_E_POINTER);
        SHL_TRY
        as<SecretKey>(thisptr)->set(host_words, (size_t)word_count);
        SHL_CATCH
    }
    namespace
    {
        SHL_HRESULT sk_load(void *thisptr, void *context, uint8_t *inptr, uint64_t size, int64_t *in_bytes, bool check)
        {
            IfNullRet(thisptr, SHL_E_POINTER);
            IfNullRet(context, SHL_E_POINTER);
            IfNullRet(inptr, SHL_E_POINTER);
            IfNullRet(in_bytes, SHL_E_POINTER);
            SHL_TRY
            auto sk = as<SecretKey>(thisptr);
            auto c = as<Context>(context);
            if (&sk->context() != c)
                throw std::invalid_argument("secret key belongs to another context");
            // SecretKey::load = Plaintext::unsafe_load + is_valid_for(SecretKey) (secretkey.h:134-170; valcheck.cpp: key-level
            // parms_id, every coefficient reduced); the device object needs the key-level layout for unsafe_load as well
            serial::PlaintextImage img;
            const size_t n = serial::load_plaintext(*c, inptr, (size_t)size, false, img);
            if (img.level != &c->key_level() || (check && !serial::plaintext_in_range(*c, img)))
                throw std::logic_error("SecretKey data is invalid");
            sk->set(img.stored, (size_t)img.coeff_count);
            *in_bytes = (int64_t)n;
            SHL_CATCH
        }
    } // namespace
    SHL_FUNC SecretKey_Load(void *thisptr, void *context, uint8_t *inptr, uint64_t size, int64_t *in_bytes)
    {
        return sk_load(thisptr, context, inptr, size, in_bytes, true);
    }
    SHL_FUNC SecretKey_UnsafeLoad(void *thisptr, void *context, uint8_t *inptr, uint64_t size, int64_t *in_bytes)
    {
        return sk_load(thisptr, context, inptr, size, in_bytes, false);
    }
    SHL_FUNC Decryptor_Create(void *context, void *secret_key, void **decryptor)
    {
        IfNullRet(context, SHL_E_POINTER);
        IfNullRet(secret_key, SHL_E_POINTER);
        IfNullRet(decryptor, SHL_E_POINTER);
        SHL_TRY
        *decryptor = new Decryptor(*as<Context>(context), *as<SecretKey>(secret_key));
        SHL_CATCH
    }
    SHL_FUNC Decryptor_Destroy(void *thisptr)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        delete as<Decryptor>(thisptr);
        return SHL_S_OK;
    }
    SHL_FUNC Decryptor_Decrypt(void *thisptr, void *encrypted, void *destination)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(encrypted, SHL_E_POINTER);
        IfNullRet(destination, SHL_E_POINTER);
        SHL_TRY
        hip_ok(hipDeviceSynchronize(), "sync");
        as<Decryptor>(thisptr)->decrypt(*as<Ciphertext>(encrypted), *as<Plaintext>(destination));
        SHL_CATCH
    }
    SHL_FUNC Decryptor_InvariantNoiseBudget(void *thisptr, void *encrypted, int *invariant_noise_budget)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(encrypted, SHL_E_POINTER);
        IfNullRet(invariant_noise_budget, SHL_E_POINTER);
        SHL_TRY
        hip_ok(hipDeviceSynchronize(), "sync");
        *invariant_noise_budget = as<Decryptor>(thisptr)->invariant_noise_budget(*as<Ciphertext>(encrypted));
        SHL_CATCH
    }
    SHL_FUNC Decryptor_DecryptBatchWords(void *thisptr, void *encrypted, uint64_t *word_count)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(encrypted, SHL_E_POINTER);
        IfNullRet(word_count, SHL_E_POINTER);
        SHL_TRY
        *word_count = as<Decryptor>(thisptr)->decrypt_batch_words(*as<Ciphertext>(encrypted));
        SHL_CATCH
    }
    SHL_FUNC Decryptor_DecryptBatch(void *thisptr, void *encrypted, uint64_t *device_out, uint64_t word_count)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(encrypted, SHL_E_POINTER);
        IfNullRet(device_out, SHL_E_POINTER);
        SHL_TRY
        auto d = as<Decryptor>(thisptr);
        if (word_count != d->decrypt_batch_words(*as<Ciphertext>(encrypted)))
            throw std::invalid_argument("word_count does not match Decryptor_DecryptBatchWords");
        hip_ok(hipDeviceSynchronize(), "sync");
        d->decrypt_batch(*as<Ciphertext>(encrypted), device_out);
        SHL_CATCH
    }

    // ------------------------------------------------------------------ CKKSEncoder (native/src/seal/c/ckksencoder.h)
    SHL_FUNC CKKSEncoder_Create(void *context, void **ckks_encoder)
    {
        IfNullRet(context, SHL_E_POINTER);
        IfNullRet(ckks_encoder, SHL_E_POINTER);
        SHL_TRY
        *ckks_encoder = new CKKSEncoder(*as<Context>(context));
        SHL_CATCH
    }
    SHL_FUNC CKKSEncoder_Destroy(void *thisptr)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        delete as<CKKSEncoder>(thisptr);
        return SHL_S_OK;
    }
    SHL_FUNC CKKSEncoder_SlotCount(void *thisptr, uint64_t *slot_count)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(slot_count, SHL_E_POINTER);
        *slot_count = as<CKKSEncoder>(thisptr)->slot_count();
        return SHL_S_OK;
    }
    SHL_FUNC CKKSEncoder_Encode1(void *thisptr, uint64_t value_count, double *values, uint64_t *parms_id, double scale, void *destination, void *pool)
    {
        (void)pool;
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(parms_id, SHL_E_POINTER);
        IfNullRet(destination, SHL_E_POINTER);
        SHL_TRY
        as<CKKSEncoder>(thisptr)->encode(values, (size_t)value_count, false, parms_id, scale, *as<Plaintext>(destination));
        SHL_CATCH
    }
    SHL_FUNC CKKSEncoder_Encode2(void *thisptr, uint64_t value_count, double *complex_values, uint64_t *parms_id, double scale, void *destination,
                                 void *pool)
    {
        (void)pool;
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(parms_id, SHL_E_POINTER);
        IfNullRet(destination, SHL_E_POINTER);
        SHL_TRY
        as<CKKSEncoder>(thisptr)->encode(complex_values, (size_t)value_count, true, parms_id, scale, *as<Plaintext>(destination));
        SHL_CATCH
    }
    SHL_FUNC CKKSEncoder_Encode3(void *thisptr, double value, uint64_t *parms_id, double scale, void *destination, void *pool)
    {
        (void)pool;
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(parms_id, SHL_E_POINTER);
        IfNullRet(destination, SHL_E_POINTER);
        SHL_TRY
        as<CKKSEncoder>(thisptr)->encode_value(value, parms_id, scale, *as<Plaintext>(destination));
        SHL_CATCH
    }
    // one complex value in every slot (c/ckksencoder.h:35; ckks.h:795-800: the reference fills `slots` copies and encodes them)
    SHL_FUNC CKKSEncoder_Encode4(void *thisptr, double value_re, double value_im, uint64_t *parms_id, double scale, void *destination, void *pool)
    {
        (void)pool;
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(parms_id, SHL_E_POINTER);
        IfNullRet(destination, SHL_E_POINTER);
        SHL_TRY
        const size_t slots = as<CKKSEncoder>(thisptr)->slot_count();
        std::vector<double> v(2 * slots);
        for (size_t i = 0; i < slots; i++)
        {
            v[2 * i] = value_re;
            v[2 * i + 1] = value_im;
        }
        as<CKKSEncoder>(thisptr)->encode(v.data(), slots, true, parms_id, scale, *as<Plaintext>(destination));
        SHL_CATCH
    }
    SHL_FUNC CKKSEncoder_Encode5(void *thisptr, int64_t value, uint64_t *parms_id, void *destination)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(parms_id, SHL_E_POINTER);
        IfNullRet(destination, SHL_E_POINTER);
        SHL_TRY
        as<CKKSEncoder>(thisptr)->encode_integer(value, parms_id, *as<Plaintext>(destination));
        SHL_CATCH
    }
    SHL_FUNC CKKSEncoder_Decode1(void *thisptr, void *plain, uint64_t *value_count, double *values, void *pool)
    {
        (void)pool;
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(plain, SHL_E_POINTER);
        IfNullRet(value_count, SHL_E_POINTER);
        SHL_TRY
        as<CKKSEncoder>(thisptr)->decode(*as<Plaintext>(plain), values, false);
        *value_count = as<CKKSEncoder>(thisptr)->slot_count();
        SHL_CATCH
    }
    SHL_FUNC CKKSEncoder_Decode2(void *thisptr, void *plain, uint64_t *value_count, double *values, void *pool)
    {
        (void)pool;
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(plain, SHL_E_POINTER);
        IfNullRet(value_count, SHL_E_POINTER);
        SHL_TRY
        as<CKKSEncoder>(thisptr)->decode(*as<Plaintext>(plain), values, true);
        *value_count = as<CKKSEncoder>(thisptr)->slot_count();
        SHL_CATCH
    }

    // ------------------------------------------------------------------ BatchEncoder (native/src/seal/c/batchencoder.h)
    SHL_FUNC BatchEncoder_Create(void *context, void **batch_encoder)
    {
        IfNullRet(context, SHL_E_POINTER);
        IfNullRet(batch_encoder, SHL_E_POINTER);
        SHL_TRY
        *batch_encoder = new BatchEncoder(*as<Context>(context));
        SHL_CATCH
    }
    SHL_FUNC BatchEncoder_Destroy(void *thisptr)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        delete as<BatchEncoder>(thisptr);
        return SHL_S_OK;
    }
    SHL_FUNC BatchEncoder_GetSlotCount(void *thisptr, uint64_t *slot_count)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(slot_count, SHL_E_POINTER);
        *slot_count = as<BatchEncoder>(thisptr)->slot_count();
        return SHL_S_OK;
    }
    SHL_FUNC BatchEncoder_Encode1(void *thisptr, uint64_t count, uint64_t *values, void *destination)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(destination, SHL_E_POINTER);
        SHL_TRY
        as<BatchEncoder>(thisptr)->encode(values, (size_t)count, false, *as<Plaintext>(destination));
        SHL_CATCH
    }
    SHL_FUNC BatchEncoder_Encode2(void *thisptr, uint64_t count, int64_t *values, void *destination)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(destination, SHL_E_POINTER);
        SHL_TRY
        as<BatchEncoder>(thisptr)->encode(reinterpret_cast<const uint64_t *>(values), (size_t)count, true, *as<Plaintext>(destination));
        SHL_CATCH
    }
    SHL_FUNC BatchEncoder_Decode1(void *thisptr, void *plain, uint64_t *count, uint64_t *destination, void *pool)
    {
        (void)pool;
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(plain, SHL_E_POINTER);
        IfNullRet(count, SHL_E_POINTER);
        IfNullRet(destination, SHL_E_POINTER);
        SHL_TRY
        as<BatchEncoder>(thisptr)->decode(*as<Plaintext>(plain), destination, false);
        *count = as<BatchEncoder>(thisptr)->slot_count();
        SHL_CATCH
    }
    SHL_FUNC BatchEncoder_Decode2(void *thisptr, void *plain, uint64_t *count, int64_t *destination, void *pool)
    {
        (void)pool;
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(plain, SHL_E_POINTER);
        IfNullRet(count, SHL_E_POINTER);
        IfNullRet(destination, SHL_E_POINTER);
        SHL_TRY
        as<BatchEncoder>(thisptr)->decode(*as<Plaintext>(plain), reinterpret_cast<uint64_t *>(destination), true);
        *count = as<BatchEncoder>(thisptr)->slot_count();
        SHL_CATCH
    }
    SHL_FUNC BatchEncoder_EncodeDevice(void *thisptr, const uint64_t *device_values, uint64_t batch, bool is_signed, uint64_t *device_coefficients)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        SHL_TRY
        as<BatchEncoder>(thisptr)->encode_device(device_values, (unsigned)batch, is_signed, device_coefficients);
        SHL_CATCH
    }
    SHL_FUNC BatchEncoder_DecodeDevice(void *thisptr, const uint64_t *device_coefficients, uint64_t batch, bool is_signed, uint64_t *device_values)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        SHL_TRY
        as<BatchEncoder>(thisptr)->decode_device(device_coefficients, (unsigned)batch, is_signed, device_values);
        SHL_CATCH
    }

    // ------------------------------------------------------------------ Encryptor, secret-key half (native/src/seal/c/encryptor.h)
    SHL_FUNC PublicKey_Create(void *context, void **public_key)
    {
        IfNullRet(context, SHL_E_POINTER);
        IfNullRet(public_key, SHL_E_POINTER);
        SHL_TRY
        *public_key = new PublicKey(*as<Context>(context));
        SHL_CATCH
    }
    SHL_FUNC PublicKey_Destroy(void *thisptr)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        delete as<PublicKey>(thisptr);
        return SHL_S_OK;
    }
    SHL_FUNC PublicKey_Set(void *thisptr, const uint64_t *host_words, uint64_t word_count)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(host_words, SHL_E_POINTER);
        SHL_TRY
        as<PublicKey>(thisptr)->set(host_words, (size_t)word_count);
        SHL_CATCH
    }
    namespace
    {
        SHL_HRESULT pk_load(void *thisptr, void *context, uint8_t *inptr, uint64_t size, int64_t *in_bytes, bool check)
        {
            IfNullRet(thisptr, SHL_E_POINTER);
            IfNullRet(context, SHL_E_POINTER);
            IfNullRet(inptr, SHL_E_POINTER);
            IfNullRet(in_bytes, SHL_E_POINTER);
            SHL_TRY
            auto pk = as<PublicKey>(thisptr);
            auto c = as<Context>(context);
            if (&pk->context() != c)
                throw std::invalid_argument("public key belongs to another context");
            // PublicKey::load = Ciphertext::unsafe_load + is_valid_for(PublicKey) (publickey.h:144-154; valcheck.cpp: key level, NTT
            // form, size 2, every coefficient reduced - the last part only for the checked load)
            serial::CiphertextImage img;
            const size_t n = serial::load_ciphertext(*c, inptr, (size_t)size, false, img, true);
            bool ok = img.level == &c->key_level() && img.is_ntt_form && img.size == 2;
            if (ok && check)
            {
                // the words that came with the stream (a seeded half expanded on the device is reduced by construction)
                const size_t host_words = img.stored_words + img.expanded.size();
                std::vector<uint64_t> words(host_words);
                img.copy_words(words.data());
                const size_t N = c->n();
                for (size_t w = 0; w < host_words && ok; w += N)
                {
                    const uint64_t q = c->coeff_modulus()[(w / N) % img.level->K];
                    for (size_t k = 0; k < N; k++)
                        ok &= words[w + k] < q;
                }
            }
            if (!ok)
                throw std::logic_error("PublicKey data is invalid");
            if (img.pending_words)
            {
                // a seeded stream (Serializable<PublicKey>): c_0 is copied, c_1 is expanded from its seed on the device
                uint64_t *dev = pk->allocate();
                copy_h2d(dev, img.stored, img.stored_words * 8);
                XofJob job;
                std::memcpy(job.seed, img.pending_seed, sizeof(job.seed));
                job.prng_type = img.pending_type;
                job.dst = dev + img.stored_words;
                sample_uniform_device(*c, c->key_level().K, { job });
            }
            else
                pk->set_parts(img.stored, img.stored_words, img.expanded.data(), img.expanded.size());
            *in_bytes = (int64_t)n;
            SHL_CATCH
        }
    } // namespace
    SHL_FUNC PublicKey_Load(void *thisptr, void *context, uint8_t *inptr, uint64_t size, int64_t *in_bytes)
    {
        return pk_load(thisptr, context, inptr, size, in_bytes, true);
    }
    SHL_FUNC PublicKey_UnsafeLoad(void *thisptr, void *context, uint8_t *inptr, uint64_t size, int64_t *in_bytes)
    {
        return pk_load(thisptr, context, inptr, size, in_bytes, false);
    }
    SHL_FUNC Encryptor_Create(void *context, void *public_key, void *secret_key, void **encryptor)
    {
        IfNullRet(context, SHL_E_POINTER);
        IfNullRet(encryptor, SHL_E_POINTER);
        SHL_TRY
        if (!public_key && !secret_key)
            throw std::invalid_argument("neither a public key nor a secret key is set");
        *encryptor = new Encryptor(*as<Context>(context), as<PublicKey>(public_key), as<SecretKey>(secret_key));
        SHL_CATCH
    }
    SHL_FUNC Encryptor_Encrypt(void *thisptr, void *plaintext, void *destination, void *pool)
    {
        (void)pool;
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(plaintext, SHL_E_POINTER);
        IfNullRet(destination, SHL_E_POINTER);
        SHL_TRY
        as<Encryptor>(thisptr)->encrypt(*as<Plaintext>(plaintext), *as<Ciphertext>(destination));
        SHL_CATCH
    }
    SHL_FUNC Encryptor_EncryptZero1(void *thisptr, uint64_t *parms_id, void *destination, void *pool)
    {
        (void)pool;
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(parms_id, SHL_E_POINTER);
        IfNullRet(destination, SHL_E_POINTER);
        SHL_TRY
        as<Encryptor>(thisptr)->encrypt_zero(parms_id, *as<Ciphertext>(destination));
        SHL_CATCH
    }
    // the forms without a parms_id encrypt at the first data level (c/encryptor.h:26, 34; encryptor.h: encrypt_zero(destination))
    SHL_FUNC Encryptor_EncryptZero2(void *thisptr, void *destination, void *pool)
    {
        (void)pool;
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(destination, SHL_E_POINTER);
        SHL_TRY
        Encryptor &e = *as<Encryptor>(thisptr);
        e.encrypt_zero(e.context().first_level().parms_id, *as<Ciphertext>(destination));
        SHL_CATCH
    }
    SHL_FUNC Encryptor_EncryptZeroSymmetric2(void *thisptr, bool save_seed, void *destination, void *pool)
    {
        (void)pool;
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(destination, SHL_E_POINTER);
        SHL_TRY
        if (save_seed)
            throw std::invalid_argument("a device ciphertext holds both polynomials: use Encryptor_EncryptZeroSymmetricSave for the seeded stream");
        Encryptor &e = *as<Encryptor>(thisptr);
        e.encrypt_zero_symmetric(e.context().first_level().parms_id, *as<Ciphertext>(destination));
        SHL_CATCH
    }
    SHL_FUNC Encryptor_Destroy(void *thisptr)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        delete as<Encryptor>(thisptr);
        return SHL_S_OK;
    }
    SHL_FUNC Encryptor_SetSeed(void *thisptr, const uint64_t *seed)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        SHL_TRY
        if (seed)
            as<Encryptor>(thisptr)->set_seed(seed);
        else
            as<Encryptor>(thisptr)->clear_seed();
        SHL_CATCH
    }
    SHL_FUNC Encryptor_EncryptZeroSymmetric1(void *thisptr, uint64_t *parms_id, bool save_seed, void *destination, void *pool)
    {
        (void)pool;
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(parms_id, SHL_E_POINTER);
        IfNullRet(destination, SHL_E_POINTER);
        SHL_TRY
        if (save_seed)
            throw std::invalid_argument("a device ciphertext holds both polynomials: use Encryptor_EncryptZeroSymmetricSave for the seeded stream");
        as<Encryptor>(thisptr)->encrypt_zero_symmetric(parms_id, *as<Ciphertext>(destination));
        SHL_CATCH
    }
    SHL_FUNC Encryptor_EncryptSymmetric(void *thisptr, void *plaintext, bool save_seed, void *destination, void *pool)
    {
        (void)pool;
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(plaintext, SHL_E_POINTER);
        IfNullRet(destination, SHL_E_POINTER);
        SHL_TRY
        if (save_seed)
            throw std::invalid_argument("a device ciphertext holds both polynomials: use Encryptor_EncryptSymmetricSave for the seeded stream");
        as<Encryptor>(thisptr)->encrypt_symmetric(*as<Plaintext>(plaintext), *as<Ciphertext>(destination));
        SHL_CATCH
    }
    SHL_FUNC Encryptor_SymmetricSaveSize(void *thisptr, uint64_t *parms_id, int64_t *result)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(parms_id, SHL_E_POINTER);
        IfNullRet(result, SHL_E_POINTER);
        SHL_TRY
        *result = (int64_t)as<Encryptor>(thisptr)->symmetric_save_size(parms_id);
        SHL_CATCH
    }
    SHL_FUNC Encryptor_EncryptZeroSymmetricSave(void *thisptr, uint64_t *parms_id, uint8_t *outptr, uint64_t size, int64_t *out_bytes)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(parms_id, SHL_E_POINTER);
        IfNullRet(outptr, SHL_E_POINTER);
        IfNullRet(out_bytes, SHL_E_POINTER);
        SHL_TRY
        *out_bytes = (int64_t)as<Encryptor>(thisptr)->encrypt_zero_symmetric_save(parms_id, outptr, (size_t)size);
        SHL_CATCH
    }
    SHL_FUNC Encryptor_EncryptSymmetricSave(void *thisptr, void *plaintext, uint8_t *outptr, uint64_t size, int64_t *out_bytes)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(plaintext, SHL_E_POINTER);
        IfNullRet(outptr, SHL_E_POINTER);
        IfNullRet(out_bytes, SHL_E_POINTER);
        SHL_TRY
        *out_bytes = (int64_t)as<Encryptor>(thisptr)->encrypt_symmetric_save(*as<Plaintext>(plaintext), outptr, (size_t)size);
        SHL_CATCH
    }

    // ------------------------------------------------------------------ KSwitchKeys
    SHL_FUNC KSwitchKeys_Create1(void **kswitch_keys)
    {
        IfNullRet(kswitch_keys, SHL_E_POINTER);
        SHL_TRY
        *kswitch_keys = new KSwitchKeys();
        SHL_CATCH
    }
    SHL_FUNC KSwitchKeys_Destroy(void *thisptr)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        delete as<KSwitchKeys>(thisptr);
        return SHL_S_OK;
    }
    SHL_FUNC KSwitchKeys_Size(void *thisptr, uint64_t *size)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(size, SHL_E_POINTER);
        *size = as<KSwitchKeys>(thisptr)->size();
        return SHL_S_OK;
    }
    SHL_FUNC KSwitchKeys_SetKey(void *thisptr, void *context, uint64_t index, uint64_t digits, const uint64_t *host_words)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(context, SHL_E_POINTER);
        IfNullRet(host_words, SHL_E_POINTER);
        SHL_TRY
        as<KSwitchKeys>(thisptr)->set_key(*as<Context>(context), index, digits, host_words, false);
        SHL_CATCH
    }
    SHL_FUNC KSwitchKeys_SetKeyFromDevice(void *thisptr, void *context, uint64_t index, uint64_t digits, const uint64_t *device_words)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(context, SHL_E_POINTER);
        IfNullRet(device_words, SHL_E_POINTER);
        SHL_TRY
        as<KSwitchKeys>(thisptr)->set_key(*as<Context>(context), index, digits, device_words, true);
        SHL_CATCH
    }
    SHL_FUNC KSwitchKeys_SetKeyDigits(void *thisptr, void *context, uint64_t index, uint64_t digit_first, uint64_t digits,
                                      const uint64_t *host_words)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(context, SHL_E_POINTER);
        IfNullRet(host_words, SHL_E_POINTER);
        SHL_TRY
        as<KSwitchKeys>(thisptr)->set_key(*as<Context>(context), index, digits, host_words, false, digit_first);
        SHL_CATCH
    }
    SHL_FUNC KSwitchKeys_HasKey(void *thisptr, uint64_t index, bool *has_key)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(has_key, SHL_E_POINTER);
        *has_key = as<KSwitchKeys>(thisptr)->has_key(index);
        return SHL_S_OK;
    }
    SHL_FUNC RelinKeys_GetIndex(uint64_t key_power, uint64_t *index)
    {
        IfNullRet(index, SHL_E_POINTER);
        SHL_TRY
        *index = Evaluator::relin_index(key_power);
        SHL_CATCH
    }
    SHL_FUNC GaloisKeys_GetIndex(uint32_t galois_elt, uint64_t *index)
    {
        IfNullRet(index, SHL_E_POINTER);
        SHL_TRY
        *index = Evaluator::galois_index(galois_elt);
        SHL_CATCH
    }
    SHL_FUNC GaloisTool_GetEltFromStep(void *context, int step, uint32_t *galois_elt)
    {
        IfNullRet(context, SHL_E_POINTER);
        IfNullRet(galois_elt, SHL_E_POINTER);
        SHL_TRY
        Evaluator ev(*as<Context>(context));
        *galois_elt = ev.galois_elt_from_step(step);
        SHL_CATCH
    }

    // ------------------------------------------------------------------ Evaluator
    SHL_FUNC Evaluator_Create(void *context, void **evaluator)
    {
        IfNullRet(context, SHL_E_POINTER);
        IfNullRet(evaluator, SHL_E_POINTER);
        SHL_TRY
        *evaluator = new Evaluator(*as<Context>(context));
        SHL_CATCH
    }
    SHL_FUNC Evaluator_Destroy(void *thisptr)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        delete as<Evaluator>(thisptr);
        return SHL_S_OK;
    }
    SHL_FUNC Evaluator_SetStream(void *thisptr, void *hip_stream)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        SHL_TRY
        as<Evaluator>(thisptr)->set_stream((hipStream_t)hip_stream);
        SHL_CATCH
    }
    SHL_FUNC Evaluator_BeginCapture(void *thisptr)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        SHL_TRY
        as<Evaluator>(thisptr)->begin_capture();
        SHL_CATCH
    }
    SHL_FUNC Evaluator_EndCapture(void *thisptr, void **graph)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(graph, SHL_E_POINTER);
        SHL_TRY
        *graph = as<Evaluator>(thisptr)->end_capture();
        SHL_CATCH
    }
    SHL_FUNC Evaluator_LaunchGraph(void *thisptr, void *graph)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(graph, SHL_E_POINTER);
        SHL_TRY
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());
        as<Evaluator>(thisptr)->launch_graph(static_cast<const Evaluator::Graph *>(graph));
        SHL_CATCH
    }
    SHL_FUNC Graph_Destroy(void *graph)
    {
        IfNullRet(graph, SHL_E_POINTER);
        delete static_cast<Evaluator::Graph *>(graph); // its scratch blocks return to the pool
        return SHL_S_OK;
    }
    SHL_FUNC Evaluator_SetTransparentCheck(void *thisptr, bool enabled)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        as<Evaluator>(thisptr)->set_transparent_check(enabled);
        return SHL_S_OK;
    }
    // destination := encrypted on the evaluator's stream (a pipeline with several evaluators / streams copies its inputs in
    // stream order; Ciphertext_Set works on the calling thread's stream)
    SHL_FUNC Evaluator_CopyTo(void *thisptr, void *encrypted, void *destination)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(encrypted, SHL_E_POINTER);
        IfNullRet(destination, SHL_E_POINTER);
        SHL_TRY
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());
        if (encrypted != destination)
            *as<Ciphertext>(destination) = *as<Ciphertext>(encrypted);
        SHL_CATCH
    }
    SHL_FUNC Evaluator_Synchronize(void *thisptr)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        SHL_TRY
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());
        as<Evaluator>(thisptr)->synchronize();
        SHL_CATCH
    }
#define EV_UNARY(fn, call)                                           \
    SHL_FUNC fn(void *thisptr, void *encrypted, void *destination)   \
    {                                                                \
        IfNullRet(thisptr, SHL_E_POINTER);                           \
        IfNullRet(encrypted, SHL_E_POINTER);                         \
        IfNullRet(destination, SHL_E_POINTER);                       \
        SHL_TRY                                                      \
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());   \
        auto ev = as<Evaluator>(thisptr);                            \
        Ciphertext &d = prepare_dest(encrypted, destination);        \
        ev->call(d);                                                 \
        SHL_CATCH                                                    \
    }
#define EV_UNARY_POOL(fn, call)                                                  \
    SHL_FUNC fn(void *thisptr, void *encrypted, void *destination, void *pool)   \
    {                                                                            \
        (void)pool;                                                              \
        IfNullRet(thisptr, SHL_E_POINTER);                                       \
        IfNullRet(encrypted, SHL_E_POINTER);                                     \
        IfNullRet(destination, SHL_E_POINTER);                                   \
        SHL_TRY                                                                  \
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());               \
        auto ev = as<Evaluator>(thisptr);                                        \
        Ciphertext &d = prepare_dest(encrypted, destination);                    \
        ev->call(d);                                                             \
        SHL_CATCH                                                                \
    }
    EV_UNARY(Evaluator_Negate, negate_inplace)
    EV_UNARY(Evaluator_TransformToNTT2, transform_to_ntt_inplace)
    EV_UNARY(Evaluator_TransformFromNTT, transform_from_ntt_inplace)
    EV_UNARY_POOL(Evaluator_Square, square_inplace)
    EV_UNARY_POOL(Evaluator_ModSwitchToNext1, mod_switch_to_next_inplace)
    EV_UNARY_POOL(Evaluator_RescaleToNext, rescale_to_next_inplace)
    EV_UNARY_POOL(Evaluator_ModReduceToNext, mod_reduce_to_next_inplace)

    // ------------------------------------------------------------------ Plaintext (native/src/seal/c/plaintext.h)
    SHL_FUNC Plaintext_Create1(void *context, void **plaintext)
    {
        IfNullRet(context, SHL_E_POINTER);
        IfNullRet(plaintext, SHL_E_POINTER);
        SHL_TRY
        *plaintext = new Plaintext(*as<Context>(context));
        SHL_CATCH
    }
    SHL_FUNC Plaintext_Create5(void *copy, void **plaintext)
    {
        IfNullRet(copy, SHL_E_POINTER);
        IfNullRet(plaintext, SHL_E_POINTER);
        SHL_TRY
        *plaintext = new Plaintext(*as<Plaintext>(copy));
        SHL_CATCH
    }
    SHL_FUNC Plaintext_Destroy(void *thisptr)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        delete as<Plaintext>(thisptr);
        return SHL_S_OK;
    }
    SHL_FUNC Plaintext_Set4(void *thisptr, uint64_t count, uint64_t *coeffs)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        if (count)
            IfNullRet(coeffs, SHL_E_POINTER);
        SHL_TRY
        as<Plaintext>(thisptr)->set(coeffs, count, false);
        SHL_CATCH
    }
    SHL_FUNC Plaintext_SetFromDevice(void *thisptr, uint64_t count, const uint64_t *device_coeffs)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        if (count)
            IfNullRet(device_coeffs, SHL_E_POINTER);
        SHL_TRY
        as<Plaintext>(thisptr)->set(device_coeffs, count, true);
        SHL_CATCH
    }
    SHL_FUNC Plaintext_CoeffCount(void *thisptr, uint64_t *coeff_count)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(coeff_count, SHL_E_POINTER);
        *coeff_count = as<Plaintext>(thisptr)->coeff_count();
        return SHL_S_OK;
    }
    SHL_FUNC Plaintext_IsNTTForm(void *thisptr, bool *is_ntt_form)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(is_ntt_form, SHL_E_POINTER);
        *is_ntt_form = as<Plaintext>(thisptr)->is_ntt_form();
        return SHL_S_OK;
    }
    SHL_FUNC Plaintext_GetParmsId(void *thisptr, uint64_t *parms_id)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(parms_id, SHL_E_POINTER);
        auto pt = as<Plaintext>(thisptr);
        if (pt->level())
            std::memcpy(parms_id, pt->level()->parms_id, 32);
        else
            std::memset(parms_id, 0, 32); // parms_id_zero
        return SHL_S_OK;
    }
    SHL_FUNC Plaintext_SetParmsId(void *thisptr, uint64_t *parms_id)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(parms_id, SHL_E_POINTER);
        SHL_TRY
        auto pt = as<Plaintext>(thisptr);
        static const uint64_t zero[4] = { 0, 0, 0, 0 };
        if (!std::memcmp(parms_id, zero, 32))
            pt->set_level(nullptr);
        else
        {
            const Level *l = pt->context().level_by_parms_id(parms_id);
            if (!l)
                throw std::invalid_argument("parms_id is not valid for encryption parameters");
            pt->set_level(l);
        }
        SHL_CATCH
    }
    SHL_FUNC Plaintext_Scale(void *thisptr, double *scale)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(scale, SHL_E_POINTER);
        *scale = as<Plaintext>(thisptr)->scale();
        return SHL_S_OK;
    }
    SHL_FUNC Plaintext_SetScale(void *thisptr, double scale)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        as<Plaintext>(thisptr)->scale() = scale;
        return SHL_S_OK;
    }
    SHL_FUNC Plaintext_CopyToHost(void *thisptr, uint64_t *dst, uint64_t word_count)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(dst, SHL_E_POINTER);
        SHL_TRY
        auto pt = as<Plaintext>(thisptr);
        if (word_count != pt->coeff_count())
            throw std::invalid_argument("word_count does not match the plaintext");
        if (word_count)
        {
            if (hipDeviceSynchronize() != hipSuccess ||
                (copy_d2h(dst, pt->data(), word_count * 8), false))
                throw std::runtime_error("HIP failure in Plaintext_CopyToHost");
        }
        SHL_CATCH
    }

    // ------------------------------------------------------------------ plaintext operands, many-operand forms
#define EV_PLAIN(fn, call)                                                       \
    SHL_FUNC fn(void *thisptr, void *encrypted, void *plain, void *destination)  \
    {                                                                            \
        IfNullRet(thisptr, SHL_E_POINTER);                                       \
        IfNullRet(encrypted, SHL_E_POINTER);                                     \
        IfNullRet(plain, SHL_E_POINTER);                                         \
        IfNullRet(destination, SHL_E_POINTER);                                   \
        SHL_TRY                                                                  \
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());               \
        as<Evaluator>(thisptr)->call(prepare_dest(encrypted, destination), *as<Plaintext>(plain)); \
        SHL_CATCH                                                                \
    }
    EV_PLAIN(Evaluator_AddPlain, add_plain_inplace)
    EV_PLAIN(Evaluator_SubPlain, sub_plain_inplace)
    SHL_FUNC Evaluator_MultiplyPlain(void *thisptr, void *encrypted, void *plain, void *destination, void *pool)
    {
        (void)pool;
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(encrypted, SHL_E_POINTER);
        IfNullRet(plain, SHL_E_POINTER);
        IfNullRet(destination, SHL_E_POINTER);
        SHL_TRY
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());
        as<Evaluator>(thisptr)->multiply_plain_inplace(prepare_dest(encrypted, destination), *as<Plaintext>(plain));
        SHL_CATCH
    }
    static Plaintext &prepare_plain_dest(void *plain, void *destination)
    {
        Plaintext *src = as<Plaintext>(plain), *dst = as<Plaintext>(destination);
        if (src != dst)
            *dst = *src;
        return *dst;
    }
    SHL_FUNC Evaluator_TransformToNTT1(void *thisptr, void *plain, uint64_t *parms_id, void *destination_ntt, void *pool)
    {
        (void)pool;
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(plain, SHL_E_POINTER);
        IfNullRet(parms_id, SHL_E_POINTER);
        IfNullRet(destination_ntt, SHL_E_POINTER);
        SHL_TRY
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());
        as<Evaluator>(thisptr)->transform_to_ntt_inplace(prepare_plain_dest(plain, destination_ntt), parms_id);
        SHL_CATCH
    }
    SHL_FUNC Evaluator_ModSwitchToNext2(void *thisptr, void *plain, void *destination)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(plain, SHL_E_POINTER);
        IfNullRet(destination, SHL_E_POINTER);
        SHL_TRY
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());
        as<Evaluator>(thisptr)->mod_switch_to_next_inplace(prepare_plain_dest(plain, destination));
        SHL_CATCH
    }
    SHL_FUNC Evaluator_ModSwitchTo2(void *thisptr, void *plain, uint64_t *parms_id, void *destination)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(plain, SHL_E_POINTER);
        IfNullRet(parms_id, SHL_E_POINTER);
        IfNullRet(destination, SHL_E_POINTER);
        SHL_TRY
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());
        as<Evaluator>(thisptr)->mod_switch_to_inplace(prepare_plain_dest(plain, destination), parms_id);
        SHL_CATCH
    }
    SHL_FUNC Evaluator_AddMany(void *thisptr, uint64_t count, void **encrypteds, void *destination)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(encrypteds, SHL_E_POINTER);
        IfNullRet(destination, SHL_E_POINTER);
        SHL_TRY
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());
        std::vector<const Ciphertext *> v;
        for (uint64_t i = 0; i < count; i++)
            v.push_back(as<Ciphertext>(encrypteds[i]));
        as<Evaluator>(thisptr)->add_many(v, *as<Ciphertext>(destination));
        SHL_CATCH
    }
    SHL_FUNC Evaluator_MultiplyMany(void *thisptr, uint64_t count, void **encrypteds, void *relin_keys, void *destination, void *pool)
    {
        (void)pool;
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(encrypteds, SHL_E_POINTER);
        IfNullRet(relin_keys, SHL_E_POINTER);
        IfNullRet(destination, SHL_E_POINTER);
        SHL_TRY
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());
        std::vector<const Ciphertext *> v;
        for (uint64_t i = 0; i < count; i++)
            v.push_back(as<Ciphertext>(encrypteds[i]));
        as<Evaluator>(thisptr)->multiply_many(v, *as<KSwitchKeys>(relin_keys), *as<Ciphertext>(destination));
        SHL_CATCH
    }
    SHL_FUNC Evaluator_Exponentiate(void *thisptr, void *encrypted, uint64_t exponent, void *relin_keys, void *destination, void *pool)
    {
        (void)pool;
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(encrypted, SHL_E_POINTER);
        IfNullRet(relin_keys, SHL_E_POINTER);
        IfNullRet(destination, SHL_E_POINTER);
        SHL_TRY
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());
        as<Evaluator>(thisptr)->exponentiate_inplace(prepare_dest(encrypted, destination), exponent, *as<KSwitchKeys>(relin_keys));
        SHL_CATCH
    }

    SHL_FUNC Evaluator_Add(void *thisptr, void *encrypted1, void *encrypted2, void *destination)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(encrypted1, SHL_E_POINTER);
        IfNullRet(encrypted2, SHL_E_POINTER);
        IfNullRet(destination, SHL_E_POINTER);
        SHL_TRY
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());
        auto ev = as<Evaluator>(thisptr);
        if (encrypted2 == destination && encrypted1 != destination)
            ev->add_inplace(*as<Ciphertext>(destination), *as<Ciphertext>(encrypted1)); // evaluator.h add(): commutes
        else
            ev->add_inplace(prepare_dest(encrypted1, destination), *as<Ciphertext>(encrypted2));
        SHL_CATCH
    }
    SHL_FUNC Evaluator_Sub(void *thisptr, void *encrypted1, void *encrypted2, void *destination)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(encrypted1, SHL_E_POINTER);
        IfNullRet(encrypted2, SHL_E_POINTER);
        IfNullRet(destination, SHL_E_POINTER);
        SHL_TRY
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());
        auto ev = as<Evaluator>(thisptr);
        if (encrypted2 == destination && encrypted1 != destination)
        {
            // evaluator.h sub(): destination = e2 - e1, then negate
            ev->sub_inplace(*as<Ciphertext>(destination), *as<Ciphertext>(encrypted1));
            ev->negate_inplace(*as<Ciphertext>(destination));
        }
        else
            ev->sub_inplace(prepare_dest(encrypted1, destination), *as<Ciphertext>(encrypted2));
        SHL_CATCH
    }
    SHL_FUNC Evaluator_Multiply(void *thisptr, void *encrypted1, void *encrypted2, void *destination, void *pool)
    {
        (void)pool;
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(encrypted1, SHL_E_POINTER);
        IfNullRet(encrypted2, SHL_E_POINTER);
        IfNullRet(destination, SHL_E_POINTER);
        SHL_TRY
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());
        auto ev = as<Evaluator>(thisptr);
        if (encrypted1 == encrypted2 && encrypted1 == destination)
            ev->multiply_inplace(*as<Ciphertext>(destination), *as<Ciphertext>(destination));
        else
            ev->multiply(*as<Ciphertext>(encrypted1), *as<Ciphertext>(encrypted2), *as<Ciphertext>(destination));
        SHL_CATCH
    }
    SHL_FUNC Evaluator_Relinearize(void *thisptr, void *encrypted, void *relinKeys, void *destination, void *pool)
    {
        (void)pool;
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(encrypted, SHL_E_POINTER);
        IfNullRet(relinKeys, SHL_E_POINTER);
        IfNullRet(destination, SHL_E_POINTER);
        SHL_TRY
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());
        as<Evaluator>(thisptr)->relinearize_inplace(prepare_dest(encrypted, destination), *as<KSwitchKeys>(relinKeys));
        SHL_CATCH
    }
    SHL_FUNC Evaluator_SwitchKeyAccWords(void *thisptr, void *encrypted, uint64_t *words)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(encrypted, SHL_E_POINTER);
        IfNullRet(words, SHL_E_POINTER);
        SHL_TRY
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());
        *words = as<Evaluator>(thisptr)->switch_key_acc_words(*as<Ciphertext>(encrypted));
        SHL_CATCH
    }
    SHL_FUNC Evaluator_RelinearizePartial(void *thisptr, void *encrypted, void *relinKeys, uint64_t digit_first, uint64_t digit_count,
                                          uint64_t *device_acc)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(encrypted, SHL_E_POINTER);
        IfNullRet(relinKeys, SHL_E_POINTER);
        IfNullRet(device_acc, SHL_E_POINTER);
        SHL_TRY
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());
        as<Evaluator>(thisptr)->relinearize_partial(*as<Ciphertext>(encrypted), *as<KSwitchKeys>(relinKeys), (unsigned)digit_first,
                                                    (unsigned)(digit_first + digit_count), device_acc);
        SHL_CATCH
    }
    SHL_FUNC Evaluator_RelinearizeFinish(void *thisptr, void *encrypted, uint64_t *device_acc, uint64_t parts)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(encrypted, SHL_E_POINTER);
        IfNullRet(device_acc, SHL_E_POINTER);
        SHL_TRY
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());
        as<Evaluator>(thisptr)->relinearize_finish(*as<Ciphertext>(encrypted), device_acc, (unsigned)parts);
        SHL_CATCH
    }
    SHL_FUNC Evaluator_ApplyGaloisPartial(void *thisptr, void *encrypted, uint32_t galois_elt, void *galoisKeys, uint64_t digit_first,
                                          uint64_t digit_count, uint64_t *device_acc)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(encrypted, SHL_E_POINTER);
        IfNullRet(galoisKeys, SHL_E_POINTER);
        IfNullRet(device_acc, SHL_E_POINTER);
        SHL_TRY
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());
        as<Evaluator>(thisptr)->apply_galois_partial(*as<Ciphertext>(encrypted), galois_elt, *as<KSwitchKeys>(galoisKeys),
                                                     (unsigned)digit_first, (unsigned)(digit_first + digit_count), device_acc);
        SHL_CATCH
    }
    SHL_FUNC Evaluator_ApplyGaloisFinish(void *thisptr, void *encrypted, uint64_t *device_acc, uint64_t parts)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(encrypted, SHL_E_POINTER);
        IfNullRet(device_acc, SHL_E_POINTER);
        SHL_TRY
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());
        as<Evaluator>(thisptr)->apply_galois_finish(*as<Ciphertext>(encrypted), device_acc, (unsigned)parts);
        SHL_CATCH
    }
    // ------------------------------------------------------------------ communicator + digit-parallel forms with the exchange
    // inside the library (sealhip.h section 1c; comm.h)
    SHL_FUNC Comm_GetUniqueId(uint8_t *id128)
    {
        IfNullRet(id128, SHL_E_POINTER);
        SHL_TRY
        Comm::unique_id(id128);
        SHL_CATCH
    }
    SHL_FUNC Comm_RcclAvailable(bool *available)
    {
        IfNullRet(available, SHL_E_POINTER);
        SHL_TRY
        *available = Comm::rccl_available();
        SHL_CATCH
    }
    SHL_FUNC Comm_Create(const uint8_t *id128, int nranks, int rank, void **comm)
    {
        IfNullRet(comm, SHL_E_POINTER);
        SHL_TRY
        *comm = new Comm(id128, nranks, rank);
        SHL_CATCH
    }
    SHL_FUNC Comm_Destroy(void *comm)
    {
        IfNullRet(comm, SHL_E_POINTER);
        delete as<Comm>(comm);
        return SHL_S_OK;
    }
    SHL_FUNC Comm_Info(void *comm, int *nranks, int *rank, bool *loopback)
    {
        IfNullRet(comm, SHL_E_POINTER);
        SHL_TRY
        if (nranks)
            *nranks = as<Comm>(comm)->size();
        if (rank)
            *rank = as<Comm>(comm)->rank();
        if (loopback)
            *loopback = as<Comm>(comm)->loopback();
        SHL_CATCH
    }
    SHL_FUNC Comm_DigitRange(void *comm, uint64_t digits, uint64_t *first, uint64_t *count)
    {
        IfNullRet(comm, SHL_E_POINTER);
        IfNullRet(first, SHL_E_POINTER);
        IfNullRet(count, SHL_E_POINTER);
        SHL_TRY
        unsigned f, c;
        comm_split((unsigned)digits, (unsigned)as<Comm>(comm)->size(), (unsigned)as<Comm>(comm)->rank(), f, c);
        *first = f;
        *count = c;
        SHL_CATCH
    }
    SHL_FUNC Comm_AllReduceWords(void *comm, uint64_t *device_words, uint64_t count, void *hip_stream)
    {
        IfNullRet(comm, SHL_E_POINTER);
        IfNullRet(device_words, SHL_E_POINTER);
        SHL_TRY
        as<Comm>(comm)->all_reduce_sum(device_words, (size_t)count, (hipStream_t)hip_stream);
        SHL_CATCH
    }
    SHL_FUNC Comm_BroadcastWords(void *comm, uint64_t *device_words, uint64_t count, int root, void *hip_stream)
    {
        IfNullRet(comm, SHL_E_POINTER);
        IfNullRet(device_words, SHL_E_POINTER);
        SHL_TRY
        as<Comm>(comm)->broadcast(device_words, (size_t)count, root, (hipStream_t)hip_stream);
        SHL_CATCH
    }
    static Evaluator::KsExchange exchange_of(int how)
    {
        if (how != 0 && how != 1)
            throw std::invalid_argument("exchange: 0 = all-reduce, 1 = reduce-scatter + all-gather");
        return how ? Evaluator::KsExchange::reduce_scatter : Evaluator::KsExchange::all_reduce;
    }
    SHL_FUNC Evaluator_RelinearizeDigitParallel(void *thisptr, void *encrypted, void *relinKeys, void *comm, int exchange, void *destination)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(encrypted, SHL_E_POINTER);
        IfNullRet(relinKeys, SHL_E_POINTER);
        IfNullRet(comm, SHL_E_POINTER);
        IfNullRet(destination, SHL_E_POINTER);
        SHL_TRY
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());
        as<Evaluator>(thisptr)->relinearize_inplace(prepare_dest(encrypted, destination), *as<KSwitchKeys>(relinKeys), *as<Comm>(comm),
                                                    exchange_of(exchange));
        SHL_CATCH
    }
    SHL_FUNC Evaluator_ApplyGaloisDigitParallel(void *thisptr, void *encrypted, uint32_t galois_elt, void *galoisKeys, void *comm, int exchange,
                                                void *destination)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(encrypted, SHL_E_POINTER);
        IfNullRet(galoisKeys, SHL_E_POINTER);
        IfNullRet(comm, SHL_E_POINTER);
        IfNullRet(destination, SHL_E_POINTER);
        SHL_TRY
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());
        as<Evaluator>(thisptr)->apply_galois_inplace(prepare_dest(encrypted, destination), galois_elt, *as<KSwitchKeys>(galoisKeys),
                                                     *as<Comm>(comm), exchange_of(exchange));
        SHL_CATCH
    }
    SHL_FUNC Evaluator_RotateVectorDigitParallel(void *thisptr, void *encrypted, int steps, void *galoisKeys, void *comm, int exchange,
                                                 void *destination)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(encrypted, SHL_E_POINTER);
        IfNullRet(galoisKeys, SHL_E_POINTER);
        IfNullRet(comm, SHL_E_POINTER);
        IfNullRet(destination, SHL_E_POINTER);
        SHL_TRY
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());
        as<Evaluator>(thisptr)->rotate_vector_inplace(prepare_dest(encrypted, destination), steps, *as<KSwitchKeys>(galoisKeys), *as<Comm>(comm),
                                                      exchange_of(exchange));
        SHL_CATCH
    }
    SHL_FUNC Evaluator_BroadcastKeyDigits(void *thisptr, void *kswitch_keys, uint64_t index, uint64_t *device_staging, void *comm, int root)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(kswitch_keys, SHL_E_POINTER);
        IfNullRet(device_staging, SHL_E_POINTER);
        IfNullRet(comm, SHL_E_POINTER);
        SHL_TRY
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());
        as<Evaluator>(thisptr)->broadcast_key_digits(*as<KSwitchKeys>(kswitch_keys), (size_t)index, device_staging, *as<Comm>(comm), root);
        SHL_CATCH
    }
    // the local phases of the reduce-scatter exchange (tests emulate the ranks in one process)
    SHL_FUNC Evaluator_SwitchKeySlots(void *thisptr, void *encrypted, uint64_t nranks, uint64_t *slots)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(encrypted, SHL_E_POINTER);
        IfNullRet(slots, SHL_E_POINTER);
        SHL_TRY
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());
        *slots = as<Evaluator>(thisptr)->switch_key_slots(*as<Ciphertext>(encrypted), (unsigned)nranks);
        SHL_CATCH
    }
    SHL_FUNC Evaluator_SwitchKeyPackTargets(void *thisptr, void *encrypted, const uint64_t *device_acc, uint64_t nranks, uint64_t *device_send,
                                            uint64_t *device_special)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(encrypted, SHL_E_POINTER);
        SHL_TRY
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());
        as<Evaluator>(thisptr)->switch_key_pack_targets(*as<Ciphertext>(encrypted), device_acc, (unsigned)nranks, device_send, device_special);
        SHL_CATCH
    }
    SHL_FUNC Evaluator_SwitchKeyFinishOwned(void *thisptr, void *encrypted, const uint64_t *device_recv, const uint64_t *device_special,
                                            uint64_t nranks, uint64_t rank, uint64_t *device_own)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(encrypted, SHL_E_POINTER);
        SHL_TRY
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());
        as<Evaluator>(thisptr)->switch_key_finish_owned(*as<Ciphertext>(encrypted), device_recv, device_special, (unsigned)nranks, (unsigned)rank,
                                                        device_own);
        SHL_CATCH
    }
    SHL_FUNC Evaluator_SwitchKeyAddGathered(void *thisptr, void *encrypted, const uint64_t *device_all, uint64_t nranks)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(encrypted, SHL_E_POINTER);
        SHL_TRY
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());
        as<Evaluator>(thisptr)->switch_key_add_gathered(*as<Ciphertext>(encrypted), device_all, (unsigned)nranks);
        SHL_CATCH
    }
    SHL_FUNC Evaluator_ModSwitchTo1(void *thisptr, void *encrypted, uint64_t *parms_id, void *destination, void *pool)
    {
        (void)pool;
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(encrypted, SHL_E_POINTER);
        IfNullRet(parms_id, SHL_E_POINTER);
        IfNullRet(destination, SHL_E_POINTER);
        SHL_TRY
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());
        as<Evaluator>(thisptr)->mod_switch_to_inplace(prepare_dest(encrypted, destination), parms_id);
        SHL_CATCH
    }
    SHL_FUNC Evaluator_RescaleTo(void *thisptr, void *encrypted, uint64_t *parms_id, void *destination, void *pool)
    {
        (void)pool;
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(encrypted, SHL_E_POINTER);
        IfNullRet(parms_id, SHL_E_POINTER);
        IfNullRet(destination, SHL_E_POINTER);
        SHL_TRY
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());
        as<Evaluator>(thisptr)->rescale_to_inplace(prepare_dest(encrypted, destination), parms_id);
        SHL_CATCH
    }
    SHL_FUNC Evaluator_ModReduceTo(void *thisptr, void *encrypted, uint64_t *parms_id, void *destination, void *pool)
    {
        (void)pool;
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(encrypted, SHL_E_POINTER);
        IfNullRet(parms_id, SHL_E_POINTER);
        IfNullRet(destination, SHL_E_POINTER);
        SHL_TRY
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());
        as<Evaluator>(thisptr)->mod_reduce_to_inplace(prepare_dest(encrypted, destination), parms_id);
        SHL_CATCH
    }
    SHL_FUNC Evaluator_ApplyGalois(void *thisptr, void *encrypted, uint32_t galois_elt, void *galoisKeys, void *destination, void *pool)
    {
        (void)pool;
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(encrypted, SHL_E_POINTER);
        IfNullRet(galoisKeys, SHL_E_POINTER);
        IfNullRet(destination, SHL_E_POINTER);
        SHL_TRY
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());
        as<Evaluator>(thisptr)->apply_galois_inplace(prepare_dest(encrypted, destination), galois_elt, *as<KSwitchKeys>(galoisKeys));
        SHL_CATCH
    }
    SHL_FUNC Evaluator_RotateRows(void *thisptr, void *encrypted, int steps, void *galoisKeys, void *destination, void *pool)
    {
        (void)pool;
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(encrypted, SHL_E_POINTER);
        IfNullRet(galoisKeys, SHL_E_POINTER);
        IfNullRet(destination, SHL_E_POINTER);
        SHL_TRY
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());
        as<Evaluator>(thisptr)->rotate_rows_inplace(prepare_dest(encrypted, destination), steps, *as<KSwitchKeys>(galoisKeys));
        SHL_CATCH
    }
    SHL_FUNC Evaluator_RotateColumns(void *thisptr, void *encrypted, void *galois_keys, void *destination, void *pool)
    {
        (void)pool;
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(encrypted, SHL_E_POINTER);
        IfNullRet(galois_keys, SHL_E_POINTER);
        IfNullRet(destination, SHL_E_POINTER);
        SHL_TRY
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());
        as<Evaluator>(thisptr)->rotate_columns_inplace(prepare_dest(encrypted, destination), *as<KSwitchKeys>(galois_keys));
        SHL_CATCH
    }
    SHL_FUNC Evaluator_RotateVector(void *thisptr, void *encrypted, int steps, void *galoisKeys, void *destination, void *pool)
    {
        (void)pool;
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(encrypted, SHL_E_POINTER);
        IfNullRet(galoisKeys, SHL_E_POINTER);
        IfNullRet(destination, SHL_E_POINTER);
        SHL_TRY
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());
        as<Evaluator>(thisptr)->rotate_vector_inplace(prepare_dest(encrypted, destination), steps, *as<KSwitchKeys>(galoisKeys));
        SHL_CATCH
    }
    SHL_FUNC Evaluator_ComplexConjugate(void *thisptr, void *encrypted, void *galoisKeys, void *destination, void *pool)
    {
        (void)pool;
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(encrypted, SHL_E_POINTER);
        IfNullRet(galoisKeys, SHL_E_POINTER);
        IfNullRet(destination, SHL_E_POINTER);
        SHL_TRY
        StreamScope stream_scope(as<Evaluator>(thisptr)->stream());
        as<Evaluator>(thisptr)->complex_conjugate_inplace(prepare_dest(encrypted, destination), *as<KSwitchKeys>(galoisKeys));
        SHL_CATCH
    }
    SHL_FUNC Evaluator_ContextUsingKeyswitching(void *thisptr, bool *using_keyswitching)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(using_keyswitching, SHL_E_POINTER);
        *using_keyswitching = as<Evaluator>(thisptr)->context().using_keyswitching();
        return SHL_S_OK;
    }

    // ------------------------------------------------------------------ per-kernel seam
    SHL_FUNC shl_ntt_forward(void *context, uint64_t *data, uint64_t polys, uint64_t comps, uint64_t first_prime, int lazy, void *stream)
    {
        IfNullRet(context, SHL_E_POINTER);
        IfNullRet(data, SHL_E_POINTER);
        SHL_TRY
        auto c = as<Context>(context);
        if (first_prime + comps > c->pool_primes().size())
            throw std::out_of_range("first_prime + comps");
        NttBatch b{};
        b.data = data;
        b.outer_stride = (size_t)comps * c->n();
        b.ncomp = (unsigned)comps;
        b.nouter = (unsigned)polys;
        b.prime_first = (unsigned)first_prime;
        hip_ok(ntt_forward(c->ntt_tables(), b, lazy, (hipStream_t)stream), "ntt_forward");
        SHL_CATCH
    }
    SHL_FUNC shl_ntt_inverse(void *context, uint64_t *data, uint64_t polys, uint64_t comps, uint64_t first_prime, int lazy, void *stream)
    {
        IfNullRet(context, SHL_E_POINTER);
        IfNullRet(data, SHL_E_POINTER);
        SHL_TRY
        auto c = as<Context>(context);
        if (first_prime + comps > c->pool_primes().size())
            throw std::out_of_range("first_prime + comps");
        NttBatch b{};
        b.data = data;
        b.outer_stride = (size_t)comps * c->n();
        b.ncomp = (unsigned)comps;
        b.nouter = (unsigned)polys;
        b.prime_first = (unsigned)first_prime;
        hip_ok(ntt_inverse(c->ntt_tables(), b, lazy, (hipStream_t)stream), "ntt_inverse");
        SHL_CATCH
    }
    SHL_FUNC shl_dyadic_product(
        void *context, const uint64_t *a, const uint64_t *b, uint64_t *r, uint64_t polys, uint64_t comps, uint64_t first_prime,
        void *stream)
    {
        IfNullRet(context, SHL_E_POINTER);
        IfNullRet(a, SHL_E_POINTER);
        IfNullRet(b, SHL_E_POINTER);
        IfNullRet(r, SHL_E_POINTER);
        SHL_TRY
        auto c = as<Context>(context);
        if (first_prime + comps > c->pool_primes().size())
            throw std::out_of_range("first_prime + comps");
        hip_ok(k_dyadic(c->dev_mods(), a, b, r, (unsigned)c->log_n(), (unsigned)comps, (unsigned)first_prime, polys, (hipStream_t)stream), "dyadic");
        SHL_CATCH
    }
    SHL_FUNC shl_apply_galois(
        void *context, uint64_t chain_index, int ntt_form, uint32_t galois_elt, const uint64_t *in, uint64_t *out, uint64_t polys,
        void *stream)
    {
        IfNullRet(context, SHL_E_POINTER);
        IfNullRet(in, SHL_E_POINTER);
        IfNullRet(out, SHL_E_POINTER);
        SHL_TRY
        auto c = as<Context>(context);
        auto l = c->level_by_chain_index(chain_index);
        if (!l)
            throw std::out_of_range("chain_index");
        if (!(galois_elt & 1) || galois_elt >= 2 * c->n())
            throw std::invalid_argument("Galois element is not valid");
        if (in == out)
            throw std::invalid_argument("result cannot point to the same value as operand");
        PlaneGeom g{ (unsigned)c->log_n(), l->K, (unsigned)polys };
        hip_ok(k_apply_galois(c->dev_mods(), in, out, galois_elt, ntt_form, g, 1, (hipStream_t)stream), "apply_galois");
        SHL_CATCH
    }
    SHL_FUNC shl_rns_stage(void *context, uint64_t chain_index, int which, const uint64_t *in, uint64_t *out, uint64_t polys, void *stream)
    {
        IfNullRet(context, SHL_E_POINTER);
        IfNullRet(in, SHL_E_POINTER);
        IfNullRet(out, SHL_E_POINTER);
        SHL_TRY
        auto c = as<Context>(context);
        auto l = c->level_by_chain_index(chain_index);
        if (!l)
            throw std::out_of_range("chain_index");
        hipStream_t s = (hipStream_t)stream;
        const unsigned n_log = (unsigned)c->log_n();
        if (which >= 0 && which <= 3)
        {
            if (c->scheme() != Scheme::bfv)
                throw std::logic_error("BEHZ stages exist only for BFV contexts");
            hip_ok(k_behz_stage(c->dev_mods(), l->dev, which, in, out, n_log, polys, s), "behz stage");
        }
        else if (which == 4)
        {
            if (l->K < 2)
                throw std::invalid_argument("level has a single modulus");
            hip_ok(k_bfv_modswitch(c->dev_mods(), l->dev, in, out, n_log, polys, s), "divide_and_round_q_last");
        }
        else if (which == 5)
        {
            if (l->K < 2)
                throw std::invalid_argument("level has a single modulus");
            const unsigned K = l->K;
            const size_t N = c->n();
            Scratch copy(polys * K * N), tt(polys * (K - 1) * N);
            hip_ok(hipMemcpyAsync(copy.p, in, polys * K * N * 8, hipMemcpyDeviceToDevice, s), "copy");
            uint64_t *last = copy.p + (size_t)(K - 1) * N;
            NttBatch bi{};
            bi.data = last;
            bi.outer_stride = (size_t)K * N;
            bi.ncomp = 1;
            bi.nouter = (unsigned)polys;
            bi.prime_first = K - 1;
            hip_ok(ntt_inverse(c->ntt_tables(), bi, 0, s), "intt last");
            NttBatch b{};
            b.data = tt.p;
            b.outer_stride = (size_t)(K - 1) * N;
            b.ncomp = K - 1;
            b.nouter = (unsigned)polys;
            b.src = last;
            b.src_outer_stride = (size_t)K * N;
            b.src_ncomp = 1;
            b.src_mode = 2;
            b.src_half = l->dev.half_q_last;
            b.src_q = l->dev.q_last;
            b.src_fix = l->dev.round_fix;
            hip_ok(ntt_forward(c->ntt_tables(), b, 1, s), "ntt correction");
            hip_ok(k_rescale_combine(c->dev_mods(), l->dev.inv_q_last_mod_q, copy.p, tt.p, out, n_log, K, polys, s), "combine");
            hip_ok(hipStreamSynchronize(s), "sync");
        }
        else
            throw std::invalid_argument("unknown stage");
        SHL_CATCH
    }
    SHL_FUNC SealHip_SetStagedHostCopies(bool enabled)
    {
        set_staged_host_copies(enabled);
        return SHL_S_OK;
    }
    SHL_FUNC SealHip_ReleasePool(void)
    {
        SHL_TRY
        DevicePool::global().release_all();
        SHL_CATCH
    }
    SHL_FUNC SealHip_PoolStats(uint64_t *bytes_held, uint64_t *cross_stream_waits)
    {
        SHL_TRY
        if (bytes_held)
            *bytes_held = DevicePool::global().bytes_held();
        if (cross_stream_waits)
            *cross_stream_waits = DevicePool::global().cross_stream_waits();
        SHL_CATCH
    }
    SHL_FUNC SealHip_TailStats(uint64_t *folded, uint64_t *plain, uint64_t *dropped)
    {
        SHL_TRY
        uint64_t f, p, d;
        lazy_tail_stats(f, p, d);
        if (folded)
            *folded = f;
        if (plain)
            *plain = p;
        if (dropped)
            *dropped = d;
        SHL_CATCH
    }
    SHL_FUNC shl_device_count(int *count)
    {
        IfNullRet(count, SHL_E_POINTER);
        SHL_TRY
        *count = 0;
        if (hipGetDeviceCount(count) != hipSuccess)
            *count = 0;
        SHL_CATCH
    }
    SHL_FUNC shl_set_device(int device)
    {
        SHL_TRY
        hip_ok(hipSetDevice(device), "hipSetDevice");
        SHL_CATCH
    }
    SHL_FUNC shl_stream_create(bool non_blocking, void **hip_stream)
    {
        IfNullRet(hip_stream, SHL_E_POINTER);
        SHL_TRY
        hipStream_t s = nullptr;
        hip_ok(hipStreamCreateWithFlags(&s, non_blocking ? hipStreamNonBlocking : hipStreamDefault), "hipStreamCreateWithFlags");
        *hip_stream = s;
        SHL_CATCH
    }
    SHL_FUNC shl_stream_destroy(void *hip_stream)
    {
        SHL_TRY
        hip_ok(hipStreamSynchronize((hipStream_t)hip_stream), "hipStreamSynchronize");
        hip_ok(hipStreamDestroy((hipStream_t)hip_stream), "hipStreamDestroy");
        SHL_CATCH
    }
    SHL_FUNC shl_malloc(uint64_t bytes, void **device_ptr)
    {
        IfNullRet(device_ptr, SHL_E_POINTER);
        SHL_TRY
        if (hipMalloc(device_ptr, bytes) != hipSuccess)
            throw std::bad_alloc();
        SHL_CATCH
    }
    SHL_FUNC shl_free(void *device_ptr)
    {
        SHL_TRY
        hip_ok(hipFree(device_ptr), "hipFree");
        SHL_CATCH
    }
    SHL_FUNC shl_memcpy_h2d(void *device_dst, const void *host_src, uint64_t bytes)
    {
        SHL_TRY
        hip_ok(hipMemcpy(device_dst, host_src, bytes, hipMemcpyHostToDevice), "H2D");
        SHL_CATCH
    }
    SHL_FUNC shl_memcpy_d2h(void *host_dst, const void *device_src, uint64_t bytes)
    {
        SHL_TRY
        hip_ok(hipDeviceSynchronize(), "sync");
        hip_ok(hipMemcpy(host_dst, device_src, bytes, hipMemcpyDeviceToHost), "D2H");
        SHL_CATCH
    }
    SHL_FUNC shl_device_synchronize(void)
    {
        SHL_TRY
        hip_ok(hipDeviceSynchronize(), "hipDeviceSynchronize");
        SHL_CATCH
    }
    SHL_FUNC shl_timer_create(void **timer)
    {
        IfNullRet(timer, SHL_E_POINTER);
        SHL_TRY
        auto t = new Timer();
        hip_ok(hipEventCreate(&t->e0), "hipEventCreate");
        hip_ok(hipEventCreate(&t->e1), "hipEventCreate");
        *timer = t;
        SHL_CATCH
    }
    SHL_FUNC shl_timer_destroy(void *timer)
    {
        IfNullRet(timer, SHL_E_POINTER);
        auto t = as<Timer>(timer);
        (void)hipEventDestroy(t->e0);
        (void)hipEventDestroy(t->e1);
        delete t;
        return SHL_S_OK;
    }
    SHL_FUNC shl_timer_start(void *timer, void *stream)
    {
        IfNullRet(timer, SHL_E_POINTER);
        SHL_TRY
        hip_ok(hipEventRecord(as<Timer>(timer)->e0, (hipStream_t)stream), "hipEventRecord");
        SHL_CATCH
    }
    SHL_FUNC shl_timer_stop(void *timer, void *stream, float *milliseconds)
    {
        IfNullRet(timer, SHL_E_POINTER);
        IfNullRet(milliseconds, SHL_E_POINTER);
        SHL_TRY
        auto t = as<Timer>(timer);
        hip_ok(hipEventRecord(t->e1, (hipStream_t)stream), "hipEventRecord");
        hip_ok(hipEventSynchronize(t->e1), "hipEventSynchronize");
        hip_ok(hipEventElapsedTime(milliseconds, t->e0, t->e1), "hipEventElapsedTime");
        SHL_CATCH
    }
}
