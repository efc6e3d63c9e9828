// extern "C" layer, part 2: SecretKey / Decryptor, KeyGenerator, CKKSEncoder, BatchEncoder, Encryptor (include/sealhip.h)
#include "capi_common.h"

extern "C"
{
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

}
