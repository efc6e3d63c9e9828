/*
 * sealhip.h — C ABI of libsealhip.so: the MI355X (gfx950) implementation of Microsoft SEAL's
 * RNS polynomial-arithmetic hot path (negacyclic NTT/INTT, dyadic products, BEHZ base conversion,
 * and the Evaluator ciphertext operations multiply / relinearize / rescale / rotate / mod_switch).
 *
 * The reference has no plugin boundary for this path (seal::Evaluator is a concrete class,
 * native/src/seal/evaluator.h:79-1387); the FFI it does ship is the flat "sealc" export layer that
 * the .NET wrapper P/Invokes (native/src/seal/c/, SEAL_C_FUNC ... HRESULT).  This header follows
 * that layer: the same entry-point names, argument order, HRESULT values and exception-to-HRESULT
 * mapping (native/src/seal/c/defines.h:36-97), so a binding written against sealc's Evaluator_* /
 * Ciphertext_* / KSwitchKeys_* / SEALContext_* functions binds to these with two deliberate
 * differences, both forced by the data living in HBM:
 *   (1) a Ciphertext handle is a DEVICE-RESIDENT BATCH of `batch` ciphertexts that share metadata
 *       (parms_id, size, is_ntt_form, scale): word index of coefficient j of RNS component r of
 *       polynomial p of batch item b is ((p*batch + b)*coeff_modulus_size + r)*N + j — for batch == 1
 *       this is exactly Ciphertext::data() (native/src/seal/ciphertext.h:337-349);
 *   (2) the `void *pool` argument of the sealc signatures is kept for signature compatibility and
 *       must be NULL (a MemoryPoolHandle has no device meaning); work is enqueued on the
 *       Evaluator's HIP stream (Evaluator_SetStream) and is asynchronous until Evaluator_Synchronize
 *       or a Ciphertext_CopyToHost.
 * No C++ type, exception or torch type crosses this boundary.
 *
 * Section 2 ("shl_*") is the finer per-kernel seam, the analogue of the reference's only accelerator
 * hook (the HEXL #ifdef in native/src/seal/util/ntt.cpp:394-475 and polyarithsmallmod.cpp:18-284),
 * taking raw device pointers; it exists for kernel-level parity tests and roofline measurements.
 */
#ifndef SEALHIP_H
#define SEALHIP_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* HRESULT values of native/src/seal/c/defines.h:36-60 (non-Windows branch). */
typedef long SHL_HRESULT;
#define SHL_S_OK ((SHL_HRESULT)0L)
#define SHL_S_FALSE ((SHL_HRESULT)1L)
#define SHL_E_POINTER ((SHL_HRESULT)0x80004003L)
#define SHL_E_INVALIDARG ((SHL_HRESULT)0x80070057L)
#define SHL_E_OUTOFMEMORY ((SHL_HRESULT)0x8007000EL)
#define SHL_E_UNEXPECTED ((SHL_HRESULT)0x8000FFFFL)
#define SHL_COR_E_IO ((SHL_HRESULT)0x80131620L)             /* std::runtime_error (HIP failure) */
#define SHL_COR_E_INVALIDOPERATION ((SHL_HRESULT)0x80131509L) /* std::logic_error */
#define SHL_E_INVALID_INDEX ((SHL_HRESULT)0x80070585L)      /* HRESULT_FROM_WIN32(ERROR_INVALID_INDEX): std::out_of_range */

#define SHL_FUNC SHL_HRESULT

/* scheme_type (native/src/seal/encryptionparams.h): 0 none, 1 bfv, 2 ckks, 3 bgv */

/* ------------------------------------------------------------------------------------------------
 * 1. sealc-shaped handle API
 * ---------------------------------------------------------------------------------------------- */

/* library / device */
SHL_FUNC SealHip_Version(uint32_t *major, uint32_t *minor, uint32_t *patch);
/* Fails with SHL_COR_E_IO when no gfx950 device is usable: there is no CPU fallback. */
SHL_FUNC SealHip_DeviceInfo(char *name, uint64_t name_capacity, int *compute_units, uint64_t *hbm_bytes);
/* last error text of the calling thread (what() of the C++ exception that produced the HRESULT) */
SHL_FUNC SealHip_LastError(char *outstr, uint64_t *length);

/* CoeffModulus::Create / PlainModulus::Batching (native/src/seal/c/modulus.h CoeffModulus_Create1) */
SHL_FUNC CoeffModulus_Create1(uint64_t poly_modulus_degree, uint64_t length, int *bit_sizes, uint64_t *coeffs);
SHL_FUNC PlainModulus_Batching(uint64_t poly_modulus_degree, int bit_size, uint64_t *value);

/* EncryptionParameters (native/src/seal/c/encryptionparameters.h) */
SHL_FUNC EncParams_Create1(uint8_t scheme, void **enc_params);
SHL_FUNC EncParams_Destroy(void *thisptr);
SHL_FUNC EncParams_SetPolyModulusDegree(void *thisptr, uint64_t degree);
SHL_FUNC EncParams_GetPolyModulusDegree(void *thisptr, uint64_t *degree);
/* sealc passes Modulus handles; here the prime values directly */
SHL_FUNC EncParams_SetCoeffModulus(void *thisptr, uint64_t length, const uint64_t *coeffs);
SHL_FUNC EncParams_GetCoeffModulus(void *thisptr, uint64_t *length, uint64_t *coeffs);
SHL_FUNC EncParams_SetPlainModulus2(void *thisptr, uint64_t plain_modulus);
SHL_FUNC EncParams_GetScheme(void *thisptr, uint8_t *scheme);

/* SEALContext (native/src/seal/c/sealcontext.h, contextdata.h).  Builds the whole modulus-switching
 * chain and uploads NTT tables / RNSTool constants to the current HIP device.  sec_level is accepted
 * for signature compatibility; the HE-standard bound check is the caller's business (BASELINE configs
 * use sec_level_type::none). */
SHL_FUNC SEALContext_Create(void *encryptionParams, bool expand_mod_chain, int sec_level, void **context);
SHL_FUNC SEALContext_Destroy(void *thisptr);
SHL_FUNC SEALContext_KeyParmsId(void *thisptr, uint64_t *parms_id);
SHL_FUNC SEALContext_FirstParmsId(void *thisptr, uint64_t *parms_id);
SHL_FUNC SEALContext_LastParmsId(void *thisptr, uint64_t *parms_id);
SHL_FUNC SEALContext_UsingKeyswitching(void *thisptr, bool *using_keyswitching);
/* ContextData_ChainIndex / ContextData_NextContextData / ContextData_Parms collapsed onto the context */
SHL_FUNC SEALContext_ChainIndex(void *thisptr, uint64_t *parms_id, uint64_t *chain_index);
SHL_FUNC SEALContext_ParmsIdAt(void *thisptr, uint64_t chain_index, uint64_t *parms_id);
SHL_FUNC SEALContext_CoeffModulusAt(void *thisptr, uint64_t chain_index, uint64_t *length, uint64_t *coeffs);
SHL_FUNC SEALContext_TotalCoeffModulusBitCount(void *thisptr, uint64_t chain_index, int *bit_count);
/* parms_ids are computed as the reference does (BLAKE2b-256 of scheme, N, primes, t: encryptionparams.cpp:117-147), so both
 * sides name levels identically; SetParmsId overrides one (kept for bindings that registered them by hand). */
SHL_FUNC SEALContext_SetParmsId(void *thisptr, uint64_t chain_index, uint64_t *parms_id);
/* introspection used by the parity tests: minimal primitive 2N-th root of a pool prime, BEHZ base */
SHL_FUNC SEALContext_NTTRoot(void *thisptr, uint64_t prime_index, uint64_t *root);
SHL_FUNC SEALContext_BaseBsk(void *thisptr, uint64_t chain_index, uint64_t *length, uint64_t *primes);

/* Ciphertext (native/src/seal/c/ciphertext.h) — device-resident batch */
SHL_FUNC Ciphertext_Create3(void *context, void *pool, void **cipher);         /* batch = 1, empty */
SHL_FUNC Ciphertext_CreateBatch(void *context, uint64_t batch, void **cipher); /* empty batch */
SHL_FUNC Ciphertext_Create2(void *copy, void **cipher);
SHL_FUNC Ciphertext_Set(void *thisptr, void *assign);
SHL_FUNC Ciphertext_Destroy(void *thisptr);
SHL_FUNC Ciphertext_Resize1(void *thisptr, void *context, uint64_t *parms_id, uint64_t size);
SHL_FUNC Ciphertext_Size(void *thisptr, uint64_t *size);
SHL_FUNC Ciphertext_BatchCount(void *thisptr, uint64_t *batch);
SHL_FUNC Ciphertext_PolyModulusDegree(void *thisptr, uint64_t *poly_modulus_degree);
SHL_FUNC Ciphertext_CoeffModulusSize(void *thisptr, uint64_t *coeff_modulus_size);
SHL_FUNC Ciphertext_ParmsId(void *thisptr, uint64_t *parms_id);
SHL_FUNC Ciphertext_IsNTTForm(void *thisptr, bool *is_ntt_form);
SHL_FUNC Ciphertext_SetIsNTTForm(void *thisptr, bool is_ntt_form);
SHL_FUNC Ciphertext_Scale(void *thisptr, double *scale);
SHL_FUNC Ciphertext_SetScale(void *thisptr, double scale);
SHL_FUNC Ciphertext_CorrectionFactor(void *thisptr, uint64_t *correction_factor);
SHL_FUNC Ciphertext_SetCorrectionFactor(void *thisptr, uint64_t correction_factor);
SHL_FUNC Ciphertext_IsTransparent(void *thisptr, bool *result); /* synchronises */
/* device slab access (replaces Ciphertext_GetDataAt/SetDataAt) */
SHL_FUNC Ciphertext_DevicePtr(void *thisptr, uint64_t **data, uint64_t *word_count);
SHL_FUNC Ciphertext_CopyFromHost(void *thisptr, const uint64_t *src, uint64_t word_count);
SHL_FUNC Ciphertext_CopyToHost(void *thisptr, uint64_t *dst, uint64_t word_count);
/* a word range of the slab (the device-resident drop-in keeps the unaligned head / tail of a host buffer current with it) */
SHL_FUNC Ciphertext_CopyWordsToHost(void *thisptr, uint64_t word_offset, uint64_t word_count, uint64_t *dst);
SHL_FUNC Ciphertext_CopyFromDevice(void *thisptr, const uint64_t *src, uint64_t word_count, void *hip_stream);

/* Wire format (native/src/seal/c/ciphertext.h:80-86; Ciphertext::save / load / unsafe_load, native/src/seal/ciphertext.cpp:153-403):
 * the reference's own byte streams - SEALHeader-framed, compr_mode none (0), zlib (1) or zstd (2: libzstd.so.1 at run time;
 * stock reference builds write zstd by default), seeded ciphertexts expanded on load with the
 * reference's Blake2xb / SHAKE256 PRNG - are parsed straight into the device slab and written back from it.  Same argument order
 * as sealc.  Load = UnsafeLoad + is_valid_for (every coefficient reduced, data level).  A handle that is a batch of one behaves
 * exactly like seal::Ciphertext; LoadItem / SaveItem address slot `item` of a larger batch (the first item loaded into an empty
 * batch defines its metadata, later ones must agree).  BGV streams in coefficient form are transformed on load as the reference does. */
SHL_FUNC Ciphertext_SaveSize(void *thisptr, uint8_t compr_mode, int64_t *result);
SHL_FUNC Ciphertext_Save(void *thisptr, uint8_t *outptr, uint64_t size, uint8_t compr_mode, int64_t *out_bytes);
SHL_FUNC Ciphertext_UnsafeLoad(void *thisptr, void *context, uint8_t *inptr, uint64_t size, int64_t *in_bytes);
SHL_FUNC Ciphertext_Load(void *thisptr, void *context, uint8_t *inptr, uint64_t size, int64_t *in_bytes);
SHL_FUNC Ciphertext_LoadItem(void *thisptr, void *context, uint64_t item, uint8_t *inptr, uint64_t size, int64_t *in_bytes);
SHL_FUNC Ciphertext_SaveItem(void *thisptr, uint64_t item, uint8_t *outptr, uint64_t size, uint8_t compr_mode, int64_t *out_bytes);

/* SecretKey / Decryptor (native/src/seal/c/secretkey.h, native/src/seal/c/decryptor.h:16-24; seal::Decryptor::decrypt,
 * native/src/seal/decryptor.cpp:79-233): the phase c_0 + c_1 s + ... on the NTT engine, then RNSTool::decrypt_scale_and_round (BFV),
 * decrypt_modt (BGV) or nothing (CKKS: the NTT-form plaintext a CKKSEncoder::decode expects).  SecretKey_Set takes
 * SecretKey::data().data() (L*N words, key level, NTT form); SecretKey_Load the serialized key.  Decryptor_Decrypt is
 * seal::Decryptor::decrypt for a batch of one; Decryptor_DecryptBatch decrypts every item of a batch into caller-owned device
 * memory, untrimmed: [batch][K][N] words (CKKS) or [batch][N] (BFV / BGV). */
SHL_FUNC SecretKey_Create(void *context, void **secret_key);
SHL_FUNC SecretKey_Destroy(void *thisptr);
SHL_FUNC SecretKey_Set(void *thisptr, const uint64_t *host_words, uint64_t word_count);
SHL_FUNC SecretKey_UnsafeLoad(void *thisptr, void *context, uint8_t *inptr, uint64_t size, int64_t *in_bytes);
SHL_FUNC SecretKey_Load(void *thisptr, void *context, uint8_t *inptr, uint64_t size, int64_t *in_bytes);
SHL_FUNC Decryptor_Create(void *context, void *secret_key, void **decryptor);
SHL_FUNC Decryptor_Destroy(void *thisptr);
SHL_FUNC Decryptor_Decrypt(void *thisptr, void *encrypted, void *destination);
SHL_FUNC Decryptor_InvariantNoiseBudget(void *thisptr, void *encrypted, int *invariant_noise_budget); /* BFV / BGV, batch of one */
SHL_FUNC Decryptor_DecryptBatchWords(void *thisptr, void *encrypted, uint64_t *word_count);
SHL_FUNC Decryptor_DecryptBatch(void *thisptr, void *encrypted, uint64_t *device_out, uint64_t word_count);

/* CKKSEncoder (native/src/seal/c/ckksencoder.h:16-49; seal::CKKSEncoder::encode / decode, native/src/seal/ckks.h:458-789): vectors of
 * N/2 real (Encode1 / Decode1) or complex (Encode2 / Decode2: interleaved re, im) numbers <-> NTT-form plaintexts at a level with a
 * scale.  The double-precision FFT, the rounding and the CRT composition repeat the reference's IEEE operations in its order, so the
 * plaintext words of Encode and the doubles of Decode are the reference's bit for bit (incl. the multi-precision branch for coefficients above 128 bits). */
SHL_FUNC CKKSEncoder_Create(void *context, void **ckks_encoder);
SHL_FUNC CKKSEncoder_Destroy(void *thisptr);
SHL_FUNC CKKSEncoder_SlotCount(void *thisptr, uint64_t *slot_count);
SHL_FUNC CKKSEncoder_Encode1(void *thisptr, uint64_t value_count, double *values, uint64_t *parms_id, double scale, void *destination, void *pool);
SHL_FUNC CKKSEncoder_Encode2(void *thisptr, uint64_t value_count, double *complex_values, uint64_t *parms_id, double scale, void *destination,
                             void *pool);
SHL_FUNC CKKSEncoder_Encode3(void *thisptr, double value, uint64_t *parms_id, double scale, void *destination, void *pool); /* one value in every slot */
SHL_FUNC CKKSEncoder_Encode4(void *thisptr, double value_re, double value_im, uint64_t *parms_id, double scale, void *destination,
                             void *pool); /* one complex value in every slot (c/ckksencoder.h:35) */
SHL_FUNC CKKSEncoder_Encode5(void *thisptr, int64_t value, uint64_t *parms_id, void *destination);
SHL_FUNC CKKSEncoder_Decode1(void *thisptr, void *plain, uint64_t *value_count, double *values, void *pool);
SHL_FUNC CKKSEncoder_Decode2(void *thisptr, void *plain, uint64_t *value_count, double *values, void *pool);

/* BatchEncoder (native/src/seal/c/batchencoder.h:16-30; seal::BatchEncoder::encode / decode, native/src/seal/batchencoder.cpp:97-447):
 * N integers modulo t <-> one plaintext polynomial through the NTT modulo t and the matrix index map.  Encode1 / Decode1 take
 * unsigned values, Encode2 / Decode2 signed ones, host vectors as in sealc (Decode writes N values).  The *Device forms convert
 * `batch` vectors [batch][N] that are already in HBM - e.g. the output of Decryptor_DecryptBatch - without leaving it. */
SHL_FUNC BatchEncoder_Create(void *context, void **batch_encoder);
SHL_FUNC BatchEncoder_Destroy(void *thisptr);
SHL_FUNC BatchEncoder_GetSlotCount(void *thisptr, uint64_t *slot_count);
SHL_FUNC BatchEncoder_Encode1(void *thisptr, uint64_t count, uint64_t *values, void *destination);
SHL_FUNC BatchEncoder_Encode2(void *thisptr, uint64_t count, int64_t *values, void *destination);
SHL_FUNC BatchEncoder_Decode1(void *thisptr, void *plain, uint64_t *count, uint64_t *destination, void *pool);
SHL_FUNC BatchEncoder_Decode2(void *thisptr, void *plain, uint64_t *count, int64_t *destination, void *pool);
SHL_FUNC BatchEncoder_EncodeDevice(void *thisptr, const uint64_t *device_values, uint64_t batch, bool is_signed, uint64_t *device_coefficients);
SHL_FUNC BatchEncoder_DecodeDevice(void *thisptr, const uint64_t *device_coefficients, uint64_t batch, bool is_signed, uint64_t *device_values);

/* PublicKey / Encryptor (native/src/seal/c/publickey.h, native/src/seal/c/encryptor.h; seal::Encryptor::encrypt_symmetric / encrypt_zero_symmetric and
 * their Serializable<> forms, native/src/seal/encryptor.cpp:116-330, util/rlwe.cpp:270-395).  The randomness follows the
 * reference: a bootstrap Blake2xb PRNG gives the public seed of c_1 (sample_poly_uniform) and the centred-binomial noise
 * (sample_poly_cbd); c_0 = -(c_1 s + e) [+ plaintext] is computed on the device.  Encryptor_SetSeed installs the reference's seeded
 * factory (every encryption restarts from that 8-word seed: reproducible runs, parity tests); NULL returns to operating-system
 * entropy.  The *Save forms write the SEEDED stream (c_0 + the 64-byte seed of c_1: half the size) a client uploads.
 * Public-key encryption (Encryptor_Encrypt / Encryptor_EncryptZero1; util::encrypt_zero_asymmetric, rlwe.cpp:196-268, then the
 * modulus switch of encryptor.cpp:139-186) draws u from the ternary distribution the way libstdc++'s uniform_int_distribution does
 * (GCC >= 11), which is what makes it byte-identical to a reference built with that library.  PublicKey_Set takes
 * PublicKey::data().data() (2*L*N words); either key of Encryptor_Create may be NULL. */
SHL_FUNC PublicKey_Create(void *context, void **public_key);
SHL_FUNC PublicKey_Destroy(void *thisptr);
SHL_FUNC PublicKey_Set(void *thisptr, const uint64_t *host_words, uint64_t word_count);
SHL_FUNC PublicKey_UnsafeLoad(void *thisptr, void *context, uint8_t *inptr, uint64_t size, int64_t *in_bytes);
SHL_FUNC PublicKey_Load(void *thisptr, void *context, uint8_t *inptr, uint64_t size, int64_t *in_bytes);
SHL_FUNC Encryptor_Create(void *context, void *public_key, void *secret_key, void **encryptor);
SHL_FUNC Encryptor_Encrypt(void *thisptr, void *plaintext, void *destination, void *pool);
SHL_FUNC Encryptor_EncryptZero1(void *thisptr, uint64_t *parms_id, void *destination, void *pool);
SHL_FUNC Encryptor_Destroy(void *thisptr);
/* INSECURE, parity tests only (see KeyGenerator_Create1): every call restarts from this seed.  NULL restores OS entropy. */
SHL_FUNC Encryptor_SetSeed(void *thisptr, const uint64_t *seed);
SHL_FUNC Encryptor_EncryptZero2(void *thisptr, void *destination, void *pool); /* at the first data level (c/encryptor.h:26) */
SHL_FUNC Encryptor_EncryptZeroSymmetric1(void *thisptr, uint64_t *parms_id, bool save_seed, void *destination, void *pool);
SHL_FUNC Encryptor_EncryptZeroSymmetric2(void *thisptr, bool save_seed, void *destination, void *pool); /* c/encryptor.h:34 */
SHL_FUNC Encryptor_EncryptSymmetric(void *thisptr, void *plaintext, bool save_seed, void *destination, void *pool);
SHL_FUNC Encryptor_SymmetricSaveSize(void *thisptr, uint64_t *parms_id, int64_t *result);
SHL_FUNC Encryptor_EncryptZeroSymmetricSave(void *thisptr, uint64_t *parms_id, uint8_t *outptr, uint64_t size, int64_t *out_bytes);
SHL_FUNC Encryptor_EncryptSymmetricSave(void *thisptr, void *plaintext, uint8_t *outptr, uint64_t size, int64_t *out_bytes);

/* KSwitchKeys / RelinKeys / GaloisKeys (native/src/seal/c/kswitchkeys.h, relinkeys.h, galoiskeys.h).
 * A key set lives in HBM; one key (index) is uploaded as the concatenation of its decomposition
 * digits, each a size-2 key-level ciphertext in NTT form: [digit][2][L][N] words
 * (KSwitchKeys::keys_[index][digit].data(), native/src/seal/kswitchkeys.h:340). */
/* Plaintext (native/src/seal/c/plaintext.h:16-75; class seal::Plaintext, native/src/seal/plaintext.h), device resident.
 * Coefficient form: `count` coefficients modulo t (BFV/BGV), parms_id = zero.  NTT form: K*N words at a level
 * (Plaintext_Set4 the words, then Plaintext_SetParmsId; CKKS plaintexts are always in this form).  One plaintext is
 * applied to every item of a ciphertext batch.  Create1 takes the context where sealc takes a pool handle. */
SHL_FUNC Plaintext_Create1(void *context, void **plaintext);
SHL_FUNC Plaintext_Create5(void *copy, void **plaintext);
SHL_FUNC Plaintext_Destroy(void *thisptr);
SHL_FUNC Plaintext_Set4(void *thisptr, uint64_t count, uint64_t *coeffs);
SHL_FUNC Plaintext_SetFromDevice(void *thisptr, uint64_t count, const uint64_t *device_coeffs);
SHL_FUNC Plaintext_CoeffCount(void *thisptr, uint64_t *coeff_count);
SHL_FUNC Plaintext_IsNTTForm(void *thisptr, bool *is_ntt_form);
SHL_FUNC Plaintext_GetParmsId(void *thisptr, uint64_t *parms_id);
SHL_FUNC Plaintext_SetParmsId(void *thisptr, uint64_t *parms_id);
SHL_FUNC Plaintext_Scale(void *thisptr, double *scale);
SHL_FUNC Plaintext_SetScale(void *thisptr, double scale);
SHL_FUNC Plaintext_CopyToHost(void *thisptr, uint64_t *dst, uint64_t word_count); /* synchronises */
/* wire format (native/src/seal/c/plaintext.h:83-89; Plaintext::save / load / unsafe_load, native/src/seal/plaintext.cpp): e.g. the
 * output of CKKSEncoder::encode / BatchEncoder::encode serialized by the client, as Ciphertext_Load above */
SHL_FUNC Plaintext_SaveSize(void *thisptr, uint8_t compr_mode, int64_t *result);
SHL_FUNC Plaintext_Save(void *thisptr, uint8_t *outptr, uint64_t size, uint8_t compr_mode, int64_t *out_bytes);
SHL_FUNC Plaintext_UnsafeLoad(void *thisptr, void *context, uint8_t *inptr, uint64_t size, int64_t *in_bytes);
SHL_FUNC Plaintext_Load(void *thisptr, void *context, uint8_t *inptr, uint64_t size, int64_t *in_bytes);

SHL_FUNC KSwitchKeys_Create1(void **kswitch_keys);
SHL_FUNC KSwitchKeys_Destroy(void *thisptr);
SHL_FUNC KSwitchKeys_Size(void *thisptr, uint64_t *size);
SHL_FUNC KSwitchKeys_SetKey(void *thisptr, void *context, uint64_t index, uint64_t digits, const uint64_t *host_words);
SHL_FUNC KSwitchKeys_SetKeyFromDevice(void *thisptr, void *context, uint64_t index, uint64_t digits, const uint64_t *device_words);
/* digit-parallel key switching (section 1b): upload only the decomposition digits [digit_first, digit_first + digits)
 * of key `index` (host_words = those digits, [digits][2][L][N]); this rank then serves exactly that digit range */
SHL_FUNC KSwitchKeys_SetKeyDigits(void *thisptr, void *context, uint64_t index, uint64_t digit_first, uint64_t digits,
                                  const uint64_t *host_words);
SHL_FUNC KSwitchKeys_HasKey(void *thisptr, uint64_t index, bool *has_key);
/* library extension: HBM held by the keys of the object.  At 2^13 <= N <= 2^16 a key is stored in the fused key-switch kernel's
 * register order (pairs of balanced doubles for primes below 2^50, (word, Shoup quotient) pairs for larger ones): 284 MB per C5 key
 * against 252 MB of natural words */
SHL_FUNC KSwitchKeys_DeviceBytes(void *thisptr, uint64_t *bytes);
/* KSwitchKeys::load / unsafe_load (native/src/seal/c/kswitchkeys.h:45-47; kswitchkeys.cpp:92-180): a serialized RelinKeys /
 * GaloisKeys stream (seeded or full, compr_mode none) goes straight into the device key slabs, every key index it holds. */
SHL_FUNC KSwitchKeys_UnsafeLoad(void *thisptr, void *context, uint8_t *inptr, uint64_t size, int64_t *in_bytes);
/* KSwitchKeys::save_size / save (native/src/seal/c/kswitchkeys.h:41-43; kswitchkeys.cpp:47-90): the FULL keys the object holds, each digit
 * as its size-2 ciphertext stream, byte for byte the reference's stream for compr_mode none (the device words go back from the
 * key-switch kernels' register order to canonical natural order first).  A digit-parallel slice is refused (logic_error). */
SHL_FUNC KSwitchKeys_SaveSize(void *thisptr, uint8_t compr_mode, int64_t *result);
SHL_FUNC KSwitchKeys_Save(void *thisptr, uint8_t *outptr, uint64_t size, uint8_t compr_mode, int64_t *out_bytes);
SHL_FUNC KSwitchKeys_Load(void *thisptr, void *context, uint8_t *inptr, uint64_t size, int64_t *in_bytes);
SHL_FUNC RelinKeys_GetIndex(uint64_t key_power, uint64_t *index);
SHL_FUNC GaloisKeys_GetIndex(uint32_t galois_elt, uint64_t *index);
/* GaloisTool::get_elt_from_step (native/src/seal/util/galois.cpp:53-95) */
SHL_FUNC GaloisTool_GetEltFromStep(void *context, int step, uint32_t *galois_elt);

/* Evaluator (native/src/seal/c/evaluator.h).  `destination` may equal `encrypted` (in place). */
SHL_FUNC Evaluator_Create(void *context, void **evaluator);
SHL_FUNC Evaluator_Destroy(void *thisptr);
/* Work of the evaluator is enqueued on `hip_stream` (NULL = the NULL stream), blocking or hipStreamNonBlocking alike.  Device
 * memory recycled through the library's pool is ordered across streams (an event wait when a block changes stream), so
 * several evaluators may run on different streams concurrently.  The destination copy of the out-of-place forms
 * (destination != encrypted) runs on the same stream as the operation.  Three families have no such copy - the result is written
 * straight into `destination` and `encrypted` is only read: Evaluator_Multiply (CKKS 2 x 2 and BFV), and Evaluator_ApplyGalois /
 * RotateRows / RotateColumns / RotateVector / ComplexConjugate when the exact Galois key is present (the reference's
 * "destination = encrypted; op_inplace(destination)", evaluator.h:239-247, 1072-1315, gives the same words).  What differs from
 * the reference is the FAILURE path of these copy-free forms only: the reference has copied the operand into the destination
 * before it validates, so a call that throws leaves a copy of `encrypted` there; here a call that fails its argument checks
 * leaves a separate destination untouched, and a Galois form that fails later (key switch, transparent-ciphertext check) leaves
 * it EMPTY (released) rather than half-built.  The HRESULT and the exception class are the reference's either way. */
SHL_FUNC Evaluator_SetStream(void *thisptr, void *hip_stream);
/* library extension: destination := encrypted, ordered on the evaluator's stream (Ciphertext_Set copies on the calling thread's
 * stream); what a pipeline over several evaluators / streams uses to stage its inputs */
SHL_FUNC Evaluator_CopyTo(void *thisptr, void *encrypted, void *destination);
SHL_FUNC Evaluator_Synchronize(void *thisptr);
/* SEAL_THROW_ON_TRANSPARENT_CIPHERTEXT (evaluator.cpp:386-392) costs a device->host round trip per
 * operation: off by default for device-resident batches, switch on for drop-in error parity. */
SHL_FUNC Evaluator_SetTransparentCheck(void *thisptr, bool enabled);
/* hipGraph capture of a fixed operation sequence (the reference has no counterpart - its operations are host
 * calls).  Between BeginCapture and EndCapture the Evaluator_* operations issued on this evaluator are recorded instead of
 * executed; Evaluator_LaunchGraph replays them with ONE launch, stream-ordered on the evaluator's stream, on the same device
 * buffers: refresh the operand ciphertexts' contents in place (Ciphertext_CopyFromHost/Device, Ciphertext_Load), replay, read the
 * destinations.  Run the sequence once eagerly before capturing; keep the captured objects alive and un-resized; issue the
 * recorded operations from the thread that called BeginCapture.  The scratch blocks the recording used stay reserved for the
 * graph (their addresses are part of it) until Graph_Destroy. */
SHL_FUNC Evaluator_BeginCapture(void *thisptr);
SHL_FUNC Evaluator_EndCapture(void *thisptr, void **graph);
SHL_FUNC Evaluator_LaunchGraph(void *thisptr, void *graph);
SHL_FUNC Graph_Destroy(void *graph);
SHL_FUNC Evaluator_Negate(void *thisptr, void *encrypted, void *destination);
SHL_FUNC Evaluator_Add(void *thisptr, void *encrypted1, void *encrypted2, void *destination);
SHL_FUNC Evaluator_Sub(void *thisptr, void *encrypted1, void *encrypted2, void *destination);
SHL_FUNC Evaluator_Multiply(void *thisptr, void *encrypted1, void *encrypted2, void *destination, void *pool);
/* plaintext operands and many-operand forms (native/src/seal/c/evaluator.h:24-37, 47-49, 59-64;
 * Evaluator::add_many / add_plain / sub_plain / multiply_plain / multiply_many / exponentiate /
 * transform_to_ntt(Plaintext) / mod_switch_to[_next](Plaintext), native/src/seal/evaluator.cpp:242-261, 1369-1402, 1649-2287) */
SHL_FUNC Evaluator_AddMany(void *thisptr, uint64_t count, void **encrypteds, void *destination);
SHL_FUNC Evaluator_AddPlain(void *thisptr, void *encrypted, void *plain, void *destination);
SHL_FUNC Evaluator_SubPlain(void *thisptr, void *encrypted, void *plain, void *destination);
SHL_FUNC Evaluator_MultiplyMany(void *thisptr, uint64_t count, void **encrypteds, void *relin_keys, void *destination, void *pool);
SHL_FUNC Evaluator_MultiplyPlain(void *thisptr, void *encrypted, void *plain, void *destination, void *pool);
SHL_FUNC Evaluator_Exponentiate(void *thisptr, void *encrypted, uint64_t exponent, void *relin_keys, void *destination, void *pool);
SHL_FUNC Evaluator_TransformToNTT1(void *thisptr, void *plain, uint64_t *parms_id, void *destination_ntt, void *pool);
SHL_FUNC Evaluator_ModSwitchToNext2(void *thisptr, void *plain, void *destination);
SHL_FUNC Evaluator_ModSwitchTo2(void *thisptr, void *plain, uint64_t *parms_id, void *destination);
SHL_FUNC Evaluator_Square(void *thisptr, void *encrypted, void *destination, void *pool);
SHL_FUNC Evaluator_Relinearize(void *thisptr, void *encrypted, void *relinKeys, void *destination, void *pool);
SHL_FUNC Evaluator_ModSwitchToNext1(void *thisptr, void *encrypted, void *destination, void *pool);
SHL_FUNC Evaluator_ModSwitchTo1(void *thisptr, void *encrypted, uint64_t *parms_id, void *destination, void *pool);
SHL_FUNC Evaluator_RescaleToNext(void *thisptr, void *encrypted, void *destination, void *pool);
SHL_FUNC Evaluator_RescaleTo(void *thisptr, void *encrypted, uint64_t *parms_id, void *destination, void *pool);
SHL_FUNC Evaluator_ModReduceToNext(void *thisptr, void *encrypted, void *destination, void *pool);
SHL_FUNC Evaluator_ModReduceTo(void *thisptr, void *encrypted, uint64_t *parms_id, void *destination, void *pool);
SHL_FUNC Evaluator_TransformToNTT2(void *thisptr, void *encrypted, void *destination_ntt);
SHL_FUNC Evaluator_TransformFromNTT(void *thisptr, void *encrypted_ntt, void *destination);
SHL_FUNC Evaluator_ApplyGalois(void *thisptr, void *encrypted, uint32_t galois_elt, void *galoisKeys, void *destination, void *pool);
SHL_FUNC Evaluator_RotateRows(void *thisptr, void *encrypted, int steps, void *galoisKeys, void *destination, void *pool);
SHL_FUNC Evaluator_RotateColumns(void *thisptr, void *encrypted, void *galois_keys, void *destination, void *pool);
SHL_FUNC Evaluator_RotateVector(void *thisptr, void *encrypted, int steps, void *galoisKeys, void *destination, void *pool);
SHL_FUNC Evaluator_ComplexConjugate(void *thisptr, void *encrypted, void *galoisKeys, void *destination, void *pool);
SHL_FUNC Evaluator_ContextUsingKeyswitching(void *thisptr, bool *using_keyswitching);

/* ------------------------------------------------------------------------------------------------
 * 1b. Digit-parallel key switching over the GPUs of one node (BASELINE configs[4]).
 * switch_key_inplace (native/src/seal/evaluator.cpp:2561-2867) is linear in the decomposition digits J up to its
 * mod-down tail, so the digits can be spread over ranks: every rank holds the same ciphertext, computes the
 * canonical partial sums S_k[I] = sum_{J in its range} NTT_I(t_J mod q_I) * key[J][k][I] for all K+1 target moduli
 * (`*Partial`), the ranks add their buffers (ONE all-reduce of 2 (K+1) N words per ciphertext, done by the caller,
 * e.g. torch.distributed.all_reduce over RCCL), and every rank finishes locally (`*Finish`: reduce mod q_I, mod-down by
 * the special prime, accumulate into (c0, c1)).  device_acc = caller-owned device buffer of Evaluator_SwitchKeyAccWords
 * 64-bit words; parts = number of summed buffers (<= 8: eight residues below 2^60 still fit one word).
 * Results equal Evaluator_Relinearize / Evaluator_ApplyGalois bit for bit.  CKKS at the two-pass sizes: `*Finish` may leave the
 * mod-down pending exactly as Evaluator_Relinearize does (deferred key-switch tail, below: a rescale on the same evaluator then
 * folds both divisions); the reduced sums are copied first, by a kernel QUEUED on the evaluator's stream: device_acc is read
 * asynchronously, so a caller that owns the buffer on another stream (a torch tensor, a communicator's receive buffer) must
 * not overwrite or free it before the evaluator's stream has passed this call - wait for an event recorded on that stream
 * after `*Finish`, or reuse the buffer only from work queued on the same stream (seal_amd/shard.py does the latter). */
SHL_FUNC Evaluator_SwitchKeyAccWords(void *thisptr, void *encrypted, uint64_t *words);
SHL_FUNC Evaluator_RelinearizePartial(void *thisptr, void *encrypted /* size 3 */, void *relinKeys, uint64_t digit_first,
                                      uint64_t digit_count, uint64_t *device_acc);
SHL_FUNC Evaluator_RelinearizeFinish(void *thisptr, void *encrypted, uint64_t *device_acc, uint64_t parts);
SHL_FUNC Evaluator_ApplyGaloisPartial(void *thisptr, void *encrypted /* size 2 */, uint32_t galois_elt, void *galoisKeys,
                                      uint64_t digit_first, uint64_t digit_count, uint64_t *device_acc);
SHL_FUNC Evaluator_ApplyGaloisFinish(void *thisptr, void *encrypted, uint64_t *device_acc, uint64_t parts);

/* ------------------------------------------------------------------------------------------------
 * 1c. The same with the exchange INSIDE the library: RCCL over xGMI, one process per GPU, collectives enqueued on the
 * evaluator's stream (no host synchronisation between the partial sums, the exchange and the mod-down).  The reference
 * has no counterpart (it is a single-host library); the loop being distributed is native/src/seal/evaluator.cpp:2663-2755
 * (digits) and 2806-2864 (target moduli).
 *   Comm_GetUniqueId     rank 0 draws the 128-byte RCCL id and hands it to the other ranks out of band
 *   Comm_Create          ncclCommInitRank on the calling thread's current device (collective over all ranks; nranks <= 8);
 *                        nranks == 1 also works without RCCL (loopback).  librccl.so.1 is loaded at run time.
 *   Comm_DigitRange      the contiguous share [first, first + count) of `digits` this rank multiplies
 *   Evaluator_BroadcastKeyDigits  one-time key distribution: `device_staging` (K*2*L*N words on every rank; on `root` the key
 *                        [digit][2][L][N]) is broadcast, every rank keeps its own digits resident
 *   Evaluator_*DigitParallel      relinearize / apply_galois / rotate_vector; every rank calls with equal ciphertexts.
 *     exchange 0: ONE all-reduce of 2 (K+1) N words per ciphertext, every rank runs the whole mod-down;
 *     exchange 1 (CKKS): reduce-scatter of the K data moduli's sums by owner + all-reduce of the special-prime component +
 *                 mod-down of the rank's own moduli + all-gather of the increments - the same bytes on the wire, the mod-down
 *                 divided by the number of ranks (BFV / BGV fall back to exchange 0).
 *   Results are bit-identical to Evaluator_Relinearize / Evaluator_ApplyGalois / Evaluator_RotateVector on one GPU.
 *   Evaluator_SwitchKeySlots / PackTargets / FinishOwned / AddGathered are the local phases of exchange 1 (layouts in
 *   seal_amd/csrc/evaluator.h); with them a caller can run the exchange through its own collective library. */
SHL_FUNC Comm_GetUniqueId(uint8_t *id128);
SHL_FUNC Comm_RcclAvailable(bool *available);
SHL_FUNC Comm_Create(const uint8_t *id128, int nranks, int rank, void **comm);
SHL_FUNC Comm_Destroy(void *comm);
SHL_FUNC Comm_Info(void *comm, int *nranks, int *rank, bool *loopback);
SHL_FUNC Comm_DigitRange(void *comm, uint64_t digits, uint64_t *first, uint64_t *count);
SHL_FUNC Comm_AllReduceWords(void *comm, uint64_t *device_words, uint64_t count, void *hip_stream);
SHL_FUNC Comm_BroadcastWords(void *comm, uint64_t *device_words, uint64_t count, int root, void *hip_stream);
SHL_FUNC Evaluator_RelinearizeDigitParallel(void *thisptr, void *encrypted, void *relinKeys, void *comm, int exchange, void *destination);
SHL_FUNC Evaluator_ApplyGaloisDigitParallel(void *thisptr, void *encrypted, uint32_t galois_elt, void *galoisKeys, void *comm, int exchange,
                                            void *destination);
SHL_FUNC Evaluator_RotateVectorDigitParallel(void *thisptr, void *encrypted, int steps, void *galoisKeys, void *comm, int exchange,
                                             void *destination);
SHL_FUNC Evaluator_BroadcastKeyDigits(void *thisptr, void *kswitch_keys, uint64_t index, uint64_t *device_staging, void *comm, int root);
SHL_FUNC Evaluator_SwitchKeySlots(void *thisptr, void *encrypted, uint64_t nranks, uint64_t *slots);
SHL_FUNC Evaluator_SwitchKeyPackTargets(void *thisptr, void *encrypted, const uint64_t *device_acc, uint64_t nranks, uint64_t *device_send,
                                        uint64_t *device_special);
SHL_FUNC Evaluator_SwitchKeyFinishOwned(void *thisptr, void *encrypted, const uint64_t *device_recv, const uint64_t *device_special,
                                        uint64_t nranks, uint64_t rank, uint64_t *device_own);
SHL_FUNC Evaluator_SwitchKeyAddGathered(void *thisptr, void *encrypted, const uint64_t *device_all, uint64_t nranks);

/* ------------------------------------------------------------------------------------------------
 * 2. Per-kernel seam on raw device slabs (device pointers; `stream` is a hipStream_t or NULL)
 * ---------------------------------------------------------------------------------------------- */

/* ntt_negacyclic_harvey[_lazy] / inverse_ntt_negacyclic_harvey[_lazy] (util/ntt.cpp:394-475) over
 * `polys` polynomials of `comps` consecutive RNS components starting at pool prime `first_prime`;
 * data = [polys][comps][N].  lazy != 0 leaves the reference's lazy range ([0,4q) fwd / [0,2q) inv). */
SHL_FUNC shl_ntt_forward(void *context, uint64_t *data, uint64_t polys, uint64_t comps, uint64_t first_prime, int lazy, void *stream);
SHL_FUNC shl_ntt_inverse(void *context, uint64_t *data, uint64_t polys, uint64_t comps, uint64_t first_prime, int lazy, void *stream);
/* dyadic_product_coeffmod (util/polyarithsmallmod.cpp:226-284): r = a .* b, operands may be lazy (< 4q) */
SHL_FUNC shl_dyadic_product(void *context, const uint64_t *a, const uint64_t *b, uint64_t *r, uint64_t polys, uint64_t comps, uint64_t first_prime, void *stream);
/* GaloisTool::apply_galois (ntt_form == 0, util/galois.cpp:148) / apply_galois_ntt (!= 0, galois.cpp:192) */
SHL_FUNC shl_apply_galois(void *context, uint64_t chain_index, int ntt_form, uint32_t galois_elt, const uint64_t *in, uint64_t *out, uint64_t polys, void *stream);
/* RNSTool stages (util/rns.cpp) on one level, `polys` polynomials each [comps][N]:
 *   0 fastbconv_m_tilde  q -> Bsk U {m~}      (rns.cpp:1086)     in K comps,        out |Bsk|+1
 *   1 sm_mrq             Bsk U {m~} -> Bsk    (rns.cpp:979)      in |Bsk|+1,        out |Bsk|
 *   2 fast_floor         q U Bsk -> Bsk       (rns.cpp:1041)     in K+|Bsk|,        out |Bsk|
 *   3 fastbconv_sk       Bsk -> q             (rns.cpp:903)      in |Bsk|,          out K
 *   4 divide_and_round_q_last_inplace         (rns.cpp:789)      in K,              out K-1
 *   5 divide_and_round_q_last_ntt_inplace     (rns.cpp:830)      in K,              out K-1 */
SHL_FUNC shl_rns_stage(void *context, uint64_t chain_index, int which, const uint64_t *in, uint64_t *out, uint64_t polys, void *stream);
/* ------------------------------------------------------------------------------------------------
 * 1d. The container surface a sealc binding uses besides the hot path (seal_amd/csrc/capi_containers.cpp)
 * ----------------------------------------------------------------------------------------------
 * Same names and argument lists as native/src/seal/c/{ciphertext,kswitchkeys,sealcontext,contextdata,encryptionparameters,
 * secretkey,publickey}.h unless a line says otherwise.  A word index addresses the device slab [poly][batch][K][N] (batch of one:
 * Ciphertext::data()); every call drains the device first (these are not hot-path functions).
 *
 * DELIBERATELY ABSENT from those seven headers (tests/test_cabi.py keeps both lists honest against the reference's headers: a sealc
 * function of these seven is declared in this file with the same argument types, or named in one of the two lists):
 *   Ciphertext_Pool, KSwitchKeys_Pool, SecretKey_Pool, PublicKey_Pool
 *                                     a MemoryPoolHandle has no device meaning (note (2) at the top)
 *   Ciphertext_Create1, SecretKey_Create1, PublicKey_Create1
 *                                     a device object is bound to a SEALContext when it is made: Ciphertext_Create3 /
 *                                     SecretKey_Create / PublicKey_Create take the context
 *   SecretKey_Data, PublicKey_Data    sealc returns a pointer to the member Plaintext / Ciphertext; the words are read with
 *                                     SecretKey_Get / PublicKey_Get and written with SecretKey_Set / PublicKey_Set
 *   EncParams_SetPlainModulus1        takes a Modulus handle; a Modulus travels as its 64-bit value here (EncParams_SetPlainModulus2)
 * SAME NAME, DIFFERENT ARGUMENTS (each for the reason given where it is declared):
 *   EncParams_GetCoeffModulus, EncParams_SetCoeffModulus, EncParams_GetPlainModulus     uint64_t values instead of Modulus handles
 *   SecretKey_Set, PublicKey_Set      take host words (section 1a); sealc's object-to-object assignment is SecretKey_Assign /
 *                                     PublicKey_Assign
 * Everything else of those headers is here or in sections 1 / 1a / 1b. */
/* Ciphertext (c/ciphertext.h:20-60).  Create4 / Create5: an empty ciphertext of batch 1 with parms_id set and room for 2 / `capacity`
 * polynomials (ciphertext.h:128-149).  SetParmsId takes the ids of the context's chain or parms_id_zero (E_INVALIDARG otherwise: a
 * device object names its level by pointer).  Resize4 is the .NET loader's resize(size, N, K): the geometry must be a level's. */
SHL_FUNC Ciphertext_Create4(void *context, uint64_t *parms_id, void *pool, void **cipher);
SHL_FUNC Ciphertext_Create5(void *context, uint64_t *parms_id, uint64_t capacity, void *pool, void **cipher);
SHL_FUNC Ciphertext_Reserve1(void *thisptr, void *context, uint64_t *parms_id, uint64_t size_capacity);
SHL_FUNC Ciphertext_Reserve2(void *thisptr, void *context, uint64_t size_capacity);
SHL_FUNC Ciphertext_Reserve3(void *thisptr, uint64_t size_capacity);
SHL_FUNC Ciphertext_SizeCapacity(void *thisptr, uint64_t *size_capacity);
SHL_FUNC Ciphertext_SetParmsId(void *thisptr, uint64_t *parms_id);
SHL_FUNC Ciphertext_Resize2(void *thisptr, void *context, uint64_t size);
SHL_FUNC Ciphertext_Resize3(void *thisptr, uint64_t size);
SHL_FUNC Ciphertext_Resize4(void *thisptr, uint64_t size, uint64_t polyModulusDegree, uint64_t coeffModCount);
SHL_FUNC Ciphertext_GetDataAt1(void *thisptr, uint64_t index, uint64_t *data);
SHL_FUNC Ciphertext_GetDataAt2(void *thisptr, uint64_t poly_index, uint64_t coeff_index, uint64_t *data); /* batch item 0 */
SHL_FUNC Ciphertext_SetDataAt(void *thisptr, uint64_t index, uint64_t value);
SHL_FUNC Ciphertext_Release(void *thisptr);
/* KSwitchKeys (c/kswitchkeys.h:20-33).  GetKeyList: sealc hands out pointers into the object; here every digit of key `index` is
 * given out as a NEW PublicKey (the device key is one slab in the kernels' order) which the caller destroys with PublicKey_Destroy;
 * key_list == NULL returns the count only.  AddKeyList appends a key made of `count` PublicKey handles (0: an empty slot). */
SHL_FUNC KSwitchKeys_Create2(void *copy, void **kswitch_keys);
SHL_FUNC KSwitchKeys_Set(void *thisptr, void *assign);
SHL_FUNC KSwitchKeys_RawSize(void *thisptr, uint64_t *key_count);
SHL_FUNC KSwitchKeys_GetKeyList(void *thisptr, uint64_t index, uint64_t *count, void **key_list);
SHL_FUNC KSwitchKeys_ClearDataAndReserve(void *thisptr, uint64_t size);
SHL_FUNC KSwitchKeys_AddKeyList(void *thisptr, uint64_t count, void **key_list);
SHL_FUNC KSwitchKeys_GetParmsId(void *thisptr, uint64_t *parms_id);
SHL_FUNC KSwitchKeys_SetParmsId(void *thisptr, uint64_t *parms_id);
/* SEALContext (c/sealcontext.h:28-40).  SEALContext_Create refuses what the reference would construct with parameters_set() ==
 * false (E_INVALIDARG, including parameters that are insecure for a sec_level of 128 / 192 / 256): an existing handle is a valid
 * context, ParametersSet is true and the error name / message are "success" / "valid".  A ContextData handle names one level of
 * the chain; it is owned by the library and valid as long as its context (ContextData_Destroy is accepted and does nothing, as
 * the pointers sealc returns belong to the SEALContext); NULL where the reference returns a null pointer. */
SHL_FUNC SEALContext_ParametersSet(void *thisptr, bool *params_set);
SHL_FUNC SEALContext_ParameterErrorName(void *thisptr, char *outstr, uint64_t *length);
SHL_FUNC SEALContext_ParameterErrorMessage(void *thisptr, char *outstr, uint64_t *length);
SHL_FUNC SEALContext_KeyContextData(void *thisptr, void **context_data);
SHL_FUNC SEALContext_FirstContextData(void *thisptr, void **context_data);
SHL_FUNC SEALContext_LastContextData(void *thisptr, void **context_data);
SHL_FUNC SEALContext_GetContextData(void *thisptr, uint64_t *parms_id, void **context_data);
/* ContextData (c/contextdata.h): array getters follow sealc's convention - *count carries the capacity in and the length out, a
 * NULL array returns the length only, a length of 0 means "not computed for these parameters" (CKKS has no coeff_div_plain_modulus,
 * BFV / BGV no upper_half_threshold).  ContextData_Parms returns a new EncryptionParameters (EncParams_Destroy), ContextData_Qualifiers
 * a new EncryptionParameterQualifiers (EPQ_Destroy; c/encryptionparameterqualifiers.h). */
SHL_FUNC ContextData_Destroy(void *thisptr);
SHL_FUNC ContextData_TotalCoeffModulus(void *thisptr, uint64_t *count, uint64_t *total_coeff_modulus);
SHL_FUNC ContextData_TotalCoeffModulusBitCount(void *thisptr, int *bit_count);
SHL_FUNC ContextData_Parms(void *thisptr, void **parms);
SHL_FUNC ContextData_Qualifiers(void *thisptr, void **epq);
SHL_FUNC ContextData_CoeffDivPlainModulus(void *thisptr, uint64_t *count, uint64_t *coeff_div);
SHL_FUNC ContextData_PlainUpperHalfThreshold(void *thisptr, uint64_t *puht);
SHL_FUNC ContextData_PlainUpperHalfIncrement(void *thisptr, uint64_t *count, uint64_t *puhi);
SHL_FUNC ContextData_UpperHalfThreshold(void *thisptr, uint64_t *count, uint64_t *uht);
SHL_FUNC ContextData_UpperHalfIncrement(void *thisptr, uint64_t *count, uint64_t *uhi);
SHL_FUNC ContextData_PrevContextData(void *thisptr, void **prev_data);
SHL_FUNC ContextData_NextContextData(void *thisptr, void **next_data);
SHL_FUNC ContextData_ChainIndex(void *thisptr, uint64_t *index);
SHL_FUNC ContextData_ParmsId(void *thisptr, uint64_t *parms_id); /* library extension: the level's parms_id without the temporary */
SHL_FUNC EPQ_Destroy(void *thisptr);
SHL_FUNC EPQ_ParametersSet(void *thisptr, bool *parameters_set);
SHL_FUNC EPQ_UsingFFT(void *thisptr, bool *using_fft);
SHL_FUNC EPQ_UsingNTT(void *thisptr, bool *using_ntt);
SHL_FUNC EPQ_UsingBatching(void *thisptr, bool *using_batching);
SHL_FUNC EPQ_UsingFastPlainLift(void *thisptr, bool *using_fast_plain_lift);
SHL_FUNC EPQ_UsingDescendingModulusChain(void *thisptr, bool *using_descending_modulus_chain);
SHL_FUNC EPQ_SecLevel(void *thisptr, int *sec_level);
/* EncryptionParameters (c/encryptionparameters.h:20-49): copy, assignment, parms_id (BLAKE2b-256, encryptionparams.cpp:117-147),
 * equality, and the wire format of EncryptionParameters::save / load (encryptionparams.cpp:15-122; all three compression modes) */
SHL_FUNC EncParams_Create2(void *copy, void **enc_params);
SHL_FUNC EncParams_Set(void *thisptr, void *assign);
SHL_FUNC EncParams_GetParmsId(void *thisptr, uint64_t *parms_id);
SHL_FUNC EncParams_GetPlainModulus(void *thisptr, uint64_t *plain_modulus); /* (uint64_t, not a Modulus handle) */
SHL_FUNC EncParams_Equals(void *thisptr, void *otherptr, bool *result);
SHL_FUNC EncParams_SaveSize(void *thisptr, uint8_t compr_mode, int64_t *result);
SHL_FUNC EncParams_Save(void *thisptr, uint8_t *outptr, uint64_t size, uint8_t compr_mode, int64_t *out_bytes);
SHL_FUNC EncParams_Load(void *thisptr, uint8_t *inptr, uint64_t size, int64_t *in_bytes);
/* SecretKey / PublicKey (c/secretkey.h, c/publickey.h): copies, parms_id, and SecretKey::save / PublicKey::save - the stream of the
 * key's Plaintext (L*N coefficients at the key level) resp. Ciphertext (size 2, key level, NTT form), byte for byte the reference's */
SHL_FUNC SecretKey_Create2(void *copy, void **secret_key);
SHL_FUNC SecretKey_Assign(void *thisptr, void *assign);
SHL_FUNC SecretKey_ParmsId(void *thisptr, uint64_t *parms_id);
SHL_FUNC SecretKey_SaveSize(void *thisptr, uint8_t compr_mode, int64_t *result);
SHL_FUNC SecretKey_Save(void *thisptr, uint8_t *outptr, uint64_t size, uint8_t compr_mode, int64_t *out_bytes);
SHL_FUNC PublicKey_Create2(void *copy, void **public_key);
SHL_FUNC PublicKey_Assign(void *thisptr, void *assign);
SHL_FUNC PublicKey_ParmsId(void *thisptr, uint64_t *parms_id);
SHL_FUNC PublicKey_SaveSize(void *thisptr, uint8_t compr_mode, int64_t *result);
SHL_FUNC PublicKey_Save(void *thisptr, uint8_t *outptr, uint64_t size, uint8_t compr_mode, int64_t *out_bytes);
/* the pool of cached HBM blocks (the role of MemoryManager / MemoryPool, native/src/seal/memorymanager.h:75-265): give the
 * cached blocks back to the driver; counters for tests */
SHL_FUNC SealHip_ReleasePool(void);
/* Ciphertext_CopyFromHost / CopyToHost / CopyWordsToHost through pinned bounce buffers owned by the library instead of a
 * direct hipMemcpy on the caller's buffer, so that the caller's pages are never registered with the driver.  For hosts that
 * change the protection of their own buffers (integration/seal_evaluator_hip.cpp); process-wide, off by default. */
SHL_FUNC SealHip_SetStagedHostCopies(bool enabled);
SHL_FUNC SealHip_PoolStats(uint64_t *bytes_held, uint64_t *cross_stream_waits);
/* Diagnostics (opt-in: process-wide handlers belong to the host program, loading the library installs none).
 * SealHip_InstallAbortTrace(path) - or SEALHIP_ABORT_TRACE=<path> in the environment when the library is loaded -
 *   * installs a std::terminate handler: an exception that escapes where none may (a destructor, a host worker thread) prints its
 *     message before the handler that was installed before it runs (no HIP call is made: the runtime may be gone by then);
 *   * catches SIGABRT, appends the call stack of the aborting thread to `path` (the ROCm runtime aborts the process itself when the
 *     device reports a memory fault: its handler on the stack tells that case from a C++ one) and then lets the abort proceed
 *     with the default disposition.  A later call only changes the path. */
SHL_FUNC SealHip_InstallAbortTrace(const char *path);
/* Environment.  The product library reads nine variables, each exercised by a test; everything else that earlier
 * rounds could switch at run time (superseded kernels, fork / no-fork of the side streams, ...) only exists in development
 * builds made with -DSEALHIP_AB_SWITCHES (seal_amd/csrc/modarith.h: shl_ab_getenv).
 *   SEALHIP_NO_FP=1                  every prime on the 64-bit integer back end (no exact double-precision arithmetic for
 *                                    primes below 2^50); same words, slower.  Read when a SEALContext is created.
 *   SEALHIP_KS_EAGER_TAIL=1          key switches complete their mod-down before they return (no deferred tails, below)
 *   SEALHIP_KS_SPLIT=<parts>         cut the digits of a key switch into <parts> in-launch groups (small batches; default:
 *                                    chosen from the batch size)
 *   SEALHIP_NTT_FCHUNKS=<n>          workgroups per component of the single-launch transforms (N = 2^13, 2^14): tests force
 *                                    the per-workgroup loop at small batches with it
 *   SEALHIP_ENCRYPT_HOST_SAMPLING=1  encryption noise is sampled on the host with the reference's own sampler instead of the
 *                                    device kernels (same distribution and, for a seeded generator, the same words)
 *   SEALHIP_ABORT_TRACE=<file>       SealHip_InstallAbortTrace(<file>) when the library is loaded (Diagnostics, above)
 *   SEALHIP_KS_CHUNK=<items>         chunk size of a key switch over a large batch (below; 0 = never cut; default: the items
 *                                    that make 8192 pass-2 workgroups - 32 at N = 2^16 with 16 moduli)
 *   SEALHIP_KS_LANES=<1..4>          streams the chunks are dealt to (default 3; 1 = one after the other on the evaluator's stream)
 *   SEALHIP_KS_SCRATCH_CAP_MIB=<n>   upper bound of the key switch's intermediate (default 16384); chunk / lanes shrink to fit */
/* Chunked key switching (round 5).  switch_key_inplace needs K (K + 1) half-transformed digits per ciphertext between its two
 * kernels (126 MB at N = 2^16, K = 15).  For a batch of 1.5 chunks or more (2^13 <= N <= 2^16, register-order keys, one digit group) the batch is cut
 * into chunks dealt round-robin to `lanes` streams forked from and joined to the evaluator's stream: the intermediate held
 * is lanes x chunk items whatever the batch (the reference holds O(K N) per ciphertext, evaluator.cpp:2561-2867); the lanes
 * hide the launch tails of the chunks (the step is as fast as with one launch over the batch, within +-1 %).  Same words as the
 * unchunked form.  While a graph is recorded the chunks stay on the recording stream.
 * Counters for tests and bench.py: calls that ran in chunks, chunks issued, the largest intermediate (bytes) any fused key switch
 * held since the previous query (the query resets it). */
SHL_FUNC SealHip_KsChunkStats(uint64_t *calls, uint64_t *chunks, uint64_t *scratch_bytes_max);
/* Deferred key-switch tails.  For CKKS and BFV at 2^13 <= N <= 2^16, Evaluator_Relinearize / ApplyGalois / RotateVector / RotateRows /
 * RotateColumns / ComplexConjugate (and, CKKS, the digit-parallel Evaluator_RelinearizeFinish / ApplyGaloisFinish and the *DigitParallel
 * forms with the all-reduce exchange) return with the mod-down by the special prime (evaluator.cpp:2806-2864) not yet run: the ciphertext
 * object keeps the key-switch sums next to its two polynomials.  Whatever needs the words next completes it first -
 * every Evaluator_* call on the object, Ciphertext_CopyToHost / Save / copies, Decryptor_Decrypt, destroying the evaluator,
 * Evaluator_SetStream, Evaluator_BeginCapture (tails from before the recording) and Evaluator_EndCapture (tails deferred inside
 * it and not consumed there become the recording's last work) - except Evaluator_RescaleToNext (in place) on the same evaluator, which performs the mod-down and
 * its own division by q_last with ONE transform per component instead of two (rns.cpp:830-901 folded in by linearity of the
 * transform; same words as the two separate steps) - and, for BFV, Evaluator_ModSwitchToNext1 (in place) on the same evaluator,
 * which does the mod-down and its own division (rns.cpp:789-828) in one element-wise pass over the inverse-transformed sums.  Metadata (size, parms_id, scale) is up to date at all times.  A deferred
 * tail runs on the stream of the evaluator that created it; a caller on another stream is made to wait for it.  Not thread
 * safe per object, like every other Ciphertext operation.  SEALHIP_KS_EAGER_TAIL=1 in the environment turns deferral off.
 * Counters for tests: tails folded into a rescale / completed on their own / discarded because the object was overwritten. */
SHL_FUNC SealHip_TailStats(uint64_t *folded, uint64_t *plain, uint64_t *dropped);
/* Deferred tensor products (round 6).  Evaluator_Multiply of two size-2 CKKS ciphertexts - into a THIRD object or in place (destination =
 * encrypted1, Evaluator_Square included) -, at the two-pass sizes and batches large enough for the un-split key switch, does not form the
 * product at once: the destination has its shape and metadata (size 3, scale, parms_id), its words are pending.  In the in-place forms
 * the destination's previous slab - its two polynomials - stays with the pending record and the destination gets a fresh slab; nothing is
 * copied.  Evaluator_Relinearize on the same evaluator then never stores the product - the third polynomial is formed inside the inverse
 * transform that opens the key switch, the first two inside the key switch's last epilogue (evaluator.cpp:604-663 folded into
 * 2561-2867; one kernel and the product's round trip through HBM less).  Anything else that touches the destination's words forms the
 * product first, and anything that writes, re-shapes or destroys a live OPERAND while a product of it is pending forms that product
 * first: the words are the reference's either way, at every point the caller can observe.  SEALHIP_LAZY_PRODUCT=0 in the environment
 * turns the deferral off.  Counters for tests: products consumed by a fused relinearisation / formed on their own / discarded because
 * the destination was overwritten. */
SHL_FUNC SealHip_ProductStats(uint64_t *fused, uint64_t *formed, uint64_t *dropped);
/* Rotations (round 6).  Evaluator_ApplyGalois / RotateVector / ComplexConjugate of a CKKS ciphertext, at the two-pass sizes and batches
 * large enough for the un-split key switch, run no permutation kernel: the key switch's own kernels read the operand's two polynomials
 * through the automorphism's index map (util/galois.cpp:18-51 folded into evaluator.cpp:2561-2867), and the result's second polynomial -
 * zero before the key switch - is neither written nor read.  Same words.  Counters for tests: rotations that took that path / that ran
 * the permutation kernels. */
SHL_FUNC SealHip_GaloisStats(uint64_t *gathered, uint64_t *permuted);
/* stream and device memory helpers for bindings without their own runtime (a PyTorch / HIP caller passes its own streams) */
/* one process per GPU: select the calling thread's device before creating a SEALContext (a PyTorch caller uses
 * torch.cuda.set_device instead) */
SHL_FUNC shl_device_count(int *count);
SHL_FUNC shl_set_device(int device);
SHL_FUNC shl_stream_create(bool non_blocking, void **hip_stream);
SHL_FUNC shl_stream_destroy(void *hip_stream);
SHL_FUNC shl_malloc(uint64_t bytes, void **device_ptr);
SHL_FUNC shl_free(void *device_ptr);
SHL_FUNC shl_memcpy_h2d(void *device_dst, const void *host_src, uint64_t bytes);
/* KeyGenerator (native/src/seal/c/keygenerator.h:16-36; seal::KeyGenerator, native/src/seal/keygenerator.cpp): secret key, public
 * key, RelinKeys and GaloisKeys generated in HBM with the reference's algorithm and randomness (device samplers over the
 * reference's BLAKE2Xb streams).  seed8 = 8 words for the reference's seeded factory (Blake2xbPRNGFactory(seed): reproducible,
 * word-for-word the reference's keys) or NULL for operating-system entropy.
 * !! seed8 != NULL IS INSECURE AND FOR PARITY TESTS ONLY.  Like the reference's seeded Blake2xbPRNGFactory (a test device there
 * too), every sampling call restarts from the same seed: the secret key, every key-switching digit and every encryption draw
 * the same (a, e), so differences of key components reveal s^2 / the rotated s, and the public seed written into saved streams
 * is the head of the stream that sampled the secret key.  Production callers pass NULL.  The same holds for Encryptor_SetSeed. !!
 * Differences from sealc: destinations are objects
 * the caller created (SecretKey_Create / PublicKey_Create / KSwitchKeys_Create1) rather than returned handles; the save_seed
 * forms write the stream directly (the *Save functions below).  KeyGenerator_KeyToHost regenerates one key in the
 * reference's layout [digit][2][L][N] into host memory (galois_elt 0 = the relinearization key); SecretKey_Get / PublicKey_Get
 * copy SecretKey::data() / PublicKey::data() to the host. */
SHL_FUNC KeyGenerator_Create1(void *context, const uint64_t *seed8, void **key_generator);
SHL_FUNC KeyGenerator_Create2(void *context, void *secret_key, const uint64_t *seed8, void **key_generator);
SHL_FUNC KeyGenerator_Destroy(void *thisptr);
SHL_FUNC KeyGenerator_SecretKey(void *thisptr, void *secret_key);
SHL_FUNC KeyGenerator_CreatePublicKey(void *thisptr, void *public_key);
SHL_FUNC KeyGenerator_CreateRelinKeys(void *thisptr, void *relin_keys);
SHL_FUNC KeyGenerator_CreateGaloisKeysFromElts(void *thisptr, uint64_t count, const uint32_t *galois_elts, void *galois_keys);
SHL_FUNC KeyGenerator_CreateGaloisKeysFromSteps(void *thisptr, uint64_t count, const int *steps, void *galois_keys);
SHL_FUNC KeyGenerator_CreateGaloisKeysAll(void *thisptr, void *galois_keys);
/* the save_seed = true forms, saved: Serializable<RelinKeys> / Serializable<GaloisKeys>::save(compr_mode none) - every digit as its seeded
 * ciphertext (c_0 + the seed of c_1), half the bytes of the full keys; byte for byte the reference's stream under its seeded factory */
SHL_FUNC KeyGenerator_SeededSaveSize(void *thisptr, bool galois, uint64_t key_count, int64_t *result);
SHL_FUNC KeyGenerator_CreateRelinKeysSave(void *thisptr, uint8_t *outptr, uint64_t size, int64_t *out_bytes);
SHL_FUNC KeyGenerator_CreateGaloisKeysFromEltsSave(void *thisptr, uint64_t count, const uint32_t *galois_elts, uint8_t *outptr, uint64_t size,
                                                   int64_t *out_bytes);
SHL_FUNC KeyGenerator_KeyToHost(void *thisptr, uint32_t galois_elt, uint64_t *host_words, uint64_t capacity_words);
SHL_FUNC SecretKey_Get(void *thisptr, uint64_t *host_words);
SHL_FUNC PublicKey_Get(void *thisptr, uint64_t *host_words);
SHL_FUNC shl_memcpy_d2h(void *host_dst, const void *device_src, uint64_t bytes);
SHL_FUNC shl_device_synchronize(void);
/* HIP-event timing on the library's stream (bench.py measures kernels where they are launched) */
SHL_FUNC shl_timer_create(void **timer);
SHL_FUNC shl_timer_destroy(void *timer);
SHL_FUNC shl_timer_start(void *timer, void *stream);
SHL_FUNC shl_timer_stop(void *timer, void *stream, float *milliseconds); /* synchronises on the stop event */

#ifdef __cplusplus
}
#endif
#endif /* SEALHIP_H */
