// extern "C" layer, part 1: library / device, parameter helpers, SEALContext, Ciphertext and Plaintext objects, wire format (include/sealhip.h)
#include "capi_common.h"
#include <memory>

namespace sealhip
{
    std::string &capi_last_error()
    {
        thread_local std::string message;
        return message;
    }
}

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
        if (outstr && *length > sealhip::capi_last_error().size())
            std::memcpy(outstr, sealhip::capi_last_error().c_str(), sealhip::capi_last_error().size() + 1);
        *length = sealhip::capi_last_error().size() + 1;
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
        IfNullRet(encryptionParams, SHL_E_POINTER);
        IfNullRet(context, SHL_E_POINTER);
        SHL_TRY
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
            throw std::runtime_error("no HIP device visible: libsealhip has no CPU fallback");
        auto p = as<EncParams>(encryptionParams);
        // sec_level_type (0 none, 128, 192, 256): the reference marks parameters whose total coefficient modulus exceeds
        // CoeffModulus::MaxBitCount(N, sec_level) (util/hestdparms.h:19-80; 0 for degrees the standard does not list) as
        // invalid_parameters_insecure (context.cpp:219-231) and every later use throws; here the context is refused
        int max_bits = 0;
        if (sec_level != 0)
        {
            static const int table[3][6] = { { 27, 54, 109, 218, 438, 881 }, { 19, 37, 75, 152, 305, 611 }, { 14, 29, 58, 118, 237, 476 } };
            const int row = sec_level == 128 ? 0 : sec_level == 192 ? 1 : sec_level == 256 ? 2 : -1;
            if (row < 0)
                throw std::invalid_argument("invalid security level");
            for (int i = 0; i < 6; i++)
                if (p->n == (uint64_t(1024) << i))
                    max_bits = table[row][i];
        }
        // ... before any device table is built: the verdict follows from the primes alone - but behind the checks the
        // reference makes first, so that malformed AND oversized parameters report the malformation
        Context::check_basic_parameters(static_cast<Scheme>(p->scheme), p->n, p->coeff_modulus);
        if (sec_level != 0 && !p->coeff_modulus.empty() && host::significant_bits(host::product(p->coeff_modulus)) > max_bits)
            throw std::invalid_argument("encryption parameters are not set correctly: not secure for the requested security level");
        std::unique_ptr<Context> c(new Context(static_cast<Scheme>(p->scheme), p->n, p->coeff_modulus, p->plain_modulus, expand_mod_chain));
        if (sec_level != 0 && c->key_level().total_coeff_modulus_bit_count > max_bits)
            throw std::invalid_argument("encryption parameters are not set correctly: not secure for the requested security level");
        c->set_sec_level(sec_level);
        *context = c.release();
        SHL_CATCH
    }
    SHL_FUNC SEALContext_Destroy(void *thisptr)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        sealhip::capi_forget_context(as<Context>(thisptr));
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
        hip_ok(hipDeviceSynchronize(), "sync"); // the copy / zero fill below run on the NULL stream: nothing queued elsewhere may still touch the words
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
        SHL_TRY // data() completes a pending key-switch tail: kernel launches, a pool allocation - it can throw
        auto ct = as<Ciphertext>(thisptr);
        *data = ct->data();
        if (word_count)
            *word_count = ct->word_count();
        SHL_CATCH
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
            keys->reserve_slots(img.keys.size());
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
            sealhip::capi_last_error() = "Ciphertext_Save needs a batch of one: use Ciphertext_SaveItem for a slot of a batch";
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
    namespace
    {
        // KSwitchKeys::save_members (kswitchkeys.cpp:47-90): parms_id, the slot count, per slot the digit count and every digit as
        // the stream of its (full, size-2, key-level, NTT-form) ciphertext - the whole inside one SEALHeader
        size_t ks_raw_size(const KSwitchKeys &keys)
        {
            size_t bytes = 16 + 32 + 8 + 8 * keys.slots();
            if (const Context *c = keys.context())
                for (size_t i = 0; i < keys.slots(); i++)
                    if (keys.has_key(i))
                    {
                        // (the same refusal as KSwitchKeys_Save: a size for a stream that cannot be written would be a trap)
                        if (keys.key(i).digit0 != 0 || keys.key(i).digits != c->first_level().K)
                            throw std::logic_error("a digit-parallel slice of a key cannot be saved");
                        bytes += keys.key(i).digits * serial::ciphertext_save_size(2, c->n(), c->key_level().K);
                    }
            return bytes;
        }
    } // namespace
    SHL_FUNC KSwitchKeys_SaveSize(void *thisptr, uint8_t compr_mode, int64_t *result)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(result, SHL_E_POINTER);
        SHL_TRY
        if (!serial::compr_mode_supported(compr_mode))
            throw std::invalid_argument("unsupported compression mode");
        *result = (int64_t)serial::compress_bound(ks_raw_size(*as<KSwitchKeys>(thisptr)), compr_mode);
        SHL_CATCH
    }
    SHL_FUNC KSwitchKeys_Save(void *thisptr, uint8_t *outptr, uint64_t size, uint8_t compr_mode, int64_t *out_bytes)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(outptr, SHL_E_POINTER);
        IfNullRet(out_bytes, SHL_E_POINTER);
        SHL_TRY
        if (!serial::compr_mode_supported(compr_mode))
            throw std::invalid_argument("unsupported compression mode");
        auto keys = as<KSwitchKeys>(thisptr);
        const Context *c = keys->context();
        const size_t raw_bytes = ks_raw_size(*keys);
        std::vector<uint8_t> raw;
        uint8_t *dst = outptr;
        if (compr_mode != 0)
        {
            raw.resize(raw_bytes);
            dst = raw.data();
        }
        else if ((size_t)size < raw_bytes)
            throw std::invalid_argument("capacity");
        size_t pos = 16;
        auto put64 = [&](uint64_t v) {
            std::memcpy(dst + pos, &v, 8);
            pos += 8;
        };
        static const uint64_t zero_id[4] = { 0, 0, 0, 0 };
        std::memcpy(dst + pos, c ? c->key_level().parms_id : zero_id, 32);
        pos += 32;
        put64(keys->slots());
        for (size_t index = 0; index < keys->slots(); index++)
        {
            if (!keys->has_key(index))
            {
                put64(0);
                continue;
            }
            const auto &k = keys->key(index);
            const size_t n = c->n(), L = c->key_level().K, words = 2 * L * n;
            if (k.digit0 != 0 || k.digits != c->first_level().K)
                throw std::logic_error("a digit-parallel slice of a key cannot be saved");
            put64(k.digits);
            Scratch natural(words); // one digit at a time: the transient device memory of a save is 2 L N words, not a whole key
            for (size_t j = 0; j < k.digits; j++)
            {
                size_t off = 0;
                const size_t bytes = serial::save_ciphertext(c->key_level().parms_id, true, 2, n, L, 1.0, 1, nullptr, dst + pos, raw_bytes - pos, &off);
                keys->digit_words(index, j, natural.p);
                hip_ok(hipMemcpy(dst + pos + off, natural.p, words * 8, hipMemcpyDeviceToHost), "D2H");
                pos += bytes;
            }
        }
        const uint8_t header[8] = { 0x5E, 0xA1, serial::kHeaderSize, serial::kVersionMajor, serial::kVersionMinor, 0, 0, 0 };
        std::memcpy(dst, header, 8);
        const uint64_t total = pos;
        std::memcpy(dst + 8, &total, 8);
        *out_bytes = (int64_t)pos;
        if (compr_mode != 0)
            *out_bytes = (int64_t)serial::compress_stream(raw.data(), pos, compr_mode, outptr, (size_t)size);
        SHL_CATCH
    }
    SHL_FUNC KSwitchKeys_DeviceBytes(void *thisptr, uint64_t *bytes)
    {
        IfNullRet(thisptr, SHL_E_POINTER);
        IfNullRet(bytes, SHL_E_POINTER);
        *bytes = as<KSwitchKeys>(thisptr)->device_bytes();
        return SHL_S_OK;
    }
    SHL_FUNC KSwitchKeys_Load(void *thisptr, void *context, uint8_t *inptr, uint64_t size, int64_t *in_bytes)
    {
        return ks_load(thisptr, context, inptr, size, in_bytes, true);
    }
    SHL_FUNC KSwitchKeys_UnsafeLoad(void *thisptr, void *context, uint8_t *inptr, uint64_t size, int64_t *in_bytes)
    {
        return ks_load(thisptr, context, inptr, size, in_bytes, false);
    }

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

}
