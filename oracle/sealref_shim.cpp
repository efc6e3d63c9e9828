// TEST INFRASTRUCTURE ONLY — never linked into, loaded by, or shipped with seal_amd/.
//
// Flat C ABI over the REAL reference (Microsoft SEAL 4.4.3 compiled from /root/reference by
// oracle/Makefile into oracle/_ref/libsealref.so).  tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg drive it through ctypes as the checker / CPU baseline
// (cpu_baseline.kind == "reference").  Every entry point wraps the reference API it names;
// raw slabs use the Ciphertext::data() layout [poly][rns][coeff] (ciphertext.h:337-349).
//
// Error convention: 0 = ok, 1 = std::invalid_argument, 2 = std::logic_error,
// 3 = std::out_of_range, 4 = other exception.  (Mirrors the classes the C export layer maps
// to HRESULTs at native/src/seal/c/defines.h:75-97.)

#include "seal/seal.h"
#include "seal/util/galois.h"
#include "seal/util/ntt.h"
#include "seal/util/polyarithsmallmod.h"
#include "seal/util/rns.h"
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <complex>
#include <cstring>
#include <memory>
#include <random>
#include <thread>
#include <vector>

using namespace seal;
using namespace seal::util;

namespace
{
    struct RefCtx
    {
        std::unique_ptr<SEALContext> context;
        std::unique_ptr<KeyGenerator> keygen;
        std::unique_ptr<Evaluator> evaluator;
        PublicKey pk;
        RelinKeys rlk;
        GaloisKeys glk;
        bool have_rlk = false, have_glk = false, have_pk = false;
        scheme_type scheme;

        std::shared_ptr<const SEALContext::ContextData> level(uint64_t chain_index) const
        {
            auto p = context->key_context_data();
            while (p && p->chain_index() != chain_index)
                p = p->next_context_data();
            return p;
        }
    };

    struct RefCt
    {
        Ciphertext ct;
    };

#define REF_TRY try {
#define REF_CATCH                          \
    }                                      \
    catch (const std::invalid_argument &)  \
    {                                      \
        return 1;                          \
    }                                      \
    catch (const std::out_of_range &)      \
    {                                      \
        return 3;                          \
    }                                      \
    catch (const std::logic_error &)       \
    {                                      \
        return 2;                          \
    }                                      \
    catch (...)                            \
    {                                      \
        return 4;                          \
    }                                      \
    return 0;
} // namespace

extern "C"
{
    // ---- parameter helpers -------------------------------------------------------------
    // CoeffModulus::Create (modulus.cpp) / PlainModulus::Batching (modulus.h:540) / BFVDefault.
    int ref_coeff_modulus_create(uint64_t n, const int *bit_sizes, uint64_t count, uint64_t *out)
    {
        REF_TRY
        std::vector<int> bits(bit_sizes, bit_sizes + count);
        auto v = CoeffModulus::Create(n, bits);
        for (size_t i = 0; i < v.size(); i++)
            out[i] = v[i].value();
        REF_CATCH
    }
    int ref_plain_modulus_batching(uint64_t n, int bits, uint64_t *out)
    {
        REF_TRY
        *out = PlainModulus::Batching(n, bits).value();
        REF_CATCH
    }
    int ref_bfv_default(uint64_t n, uint64_t *out, uint64_t *count)
    {
        REF_TRY
        auto v = CoeffModulus::BFVDefault(n, sec_level_type::tc128);
        *count = v.size();
        for (size_t i = 0; i < v.size(); i++)
            out[i] = v[i].value();
        REF_CATCH
    }

    // ---- context -----------------------------------------------------------------------
    // scheme: 1 = bfv, 2 = ckks, 3 = bgv (scheme_type, encryptionparams.h).
    int ref_ctx_create(
        int scheme, uint64_t n, const uint64_t *primes, uint64_t nprimes, uint64_t plain_modulus, uint64_t seed,
        void **out)
    {
        REF_TRY
        EncryptionParameters parms(static_cast<scheme_type>(scheme));
        parms.set_poly_modulus_degree(n);
        std::vector<Modulus> mods;
        for (uint64_t i = 0; i < nprimes; i++)
            mods.emplace_back(primes[i]);
        parms.set_coeff_modulus(mods);
        if (scheme != 2)
            parms.set_plain_modulus(plain_modulus);
        prng_seed_type s{};
        s[0] = seed;
        parms.set_random_generator(std::make_shared<Blake2xbPRNGFactory>(s));
        auto c = std::make_unique<RefCtx>();
        c->scheme = static_cast<scheme_type>(scheme);
        c->context = std::make_unique<SEALContext>(parms, true, sec_level_type::none);
        if (!c->context->parameters_set())
            return 1;
        c->evaluator = std::make_unique<Evaluator>(*c->context);
        c->keygen = std::make_unique<KeyGenerator>(*c->context);
        *out = c.release();
        REF_CATCH
    }
    void ref_ctx_destroy(void *ctx)
    {
        delete static_cast<RefCtx *>(ctx);
    }
    // chain indices: key level = levels-1 ... last = 0 (context.cpp:556-565)
    int ref_ctx_info(void *ctx, uint64_t *key_chain_index, uint64_t *first_chain_index, int *using_keyswitching)
    {
        auto c = static_cast<RefCtx *>(ctx);
        *key_chain_index = c->context->key_context_data()->chain_index();
        *first_chain_index = c->context->first_context_data()->chain_index();
        *using_keyswitching = c->context->using_keyswitching();
        return 0;
    }
    int ref_ctx_level_primes(void *ctx, uint64_t chain_index, uint64_t *out, uint64_t *count)
    {
        auto c = static_cast<RefCtx *>(ctx);
        auto l = c->level(chain_index);
        if (!l)
            return 1;
        auto &m = l->parms().coeff_modulus();
        *count = m.size();
        for (size_t i = 0; i < m.size(); i++)
            out[i] = m[i].value();
        return 0;
    }
    // NTTTables of prime `idx` at `chain_index`: root, inv_degree and the two tables' operands
    // (ntt.h:69-183).  out_fwd/out_inv may be null.
    int ref_ctx_ntt_tables(
        void *ctx, uint64_t chain_index, uint64_t idx, uint64_t *root, uint64_t *inv_degree, uint64_t *out_fwd,
        uint64_t *out_inv)
    {
        auto c = static_cast<RefCtx *>(ctx);
        auto l = c->level(chain_index);
        if (!l)
            return 1;
        auto &t = l->small_ntt_tables()[idx];
        *root = t.get_root();
        *inv_degree = t.inv_degree_modulo().operand;
        size_t n = t.coeff_count();
        for (size_t i = 0; i < n; i++)
        {
            if (out_fwd)
                out_fwd[i] = t.get_from_root_powers()[i].operand;
            if (out_inv)
                out_inv[i] = t.get_from_inv_root_powers()[i].operand;
        }
        return 0;
    }
    // RNSTool bases of a level (rns.h:246-309): out = [B primes..., m_sk], m_tilde, gamma.
    int ref_ctx_behz_bases(
        void *ctx, uint64_t chain_index, uint64_t *bsk, uint64_t *bsk_count, uint64_t *m_tilde, uint64_t *gamma)
    {
        auto c = static_cast<RefCtx *>(ctx);
        auto l = c->level(chain_index);
        if (!l)
            return 1;
        auto rt = l->rns_tool();
        *bsk_count = rt->base_Bsk()->size();
        for (size_t i = 0; i < *bsk_count; i++)
            bsk[i] = (*rt->base_Bsk())[i].value();
        *m_tilde = rt->m_tilde().value();
        *gamma = rt->gamma().value();
        return 0;
    }

    // ---- keys --------------------------------------------------------------------------
    int ref_keygen_relin(void *ctx)
    {
        REF_TRY
        auto c = static_cast<RefCtx *>(ctx);
        c->keygen->create_relin_keys(c->rlk);
        c->have_rlk = true;
        REF_CATCH
    }
    int ref_keygen_galois_elts(void *ctx, const uint32_t *elts, uint64_t count)
    {
        REF_TRY
        auto c = static_cast<RefCtx *>(ctx);
        std::vector<uint32_t> v(elts, elts + count);
        c->keygen->create_galois_keys(v, c->glk);
        c->have_glk = true;
        REF_CATCH
    }
    int ref_keygen_galois_steps(void *ctx, const int *steps, uint64_t count)
    {
        REF_TRY
        auto c = static_cast<RefCtx *>(ctx);
        std::vector<int> v(steps, steps + count);
        c->keygen->create_galois_keys(v, c->glk);
        c->have_glk = true;
        REF_CATCH
    }
    int ref_keygen_public(void *ctx)
    {
        REF_TRY
        auto c = static_cast<RefCtx *>(ctx);
        c->keygen->create_public_key(c->pk);
        c->have_pk = true;
        REF_CATCH
    }
    // Copy out one key-switching key: kind 0 = relin (index = RelinKeys::get_index(key_power),
    // relinkeys.h:58), kind 1 = galois (index = GaloisKeys::get_index(elt), galoiskeys.h:48).
    // Layout written: [digit J][k in 0..1][rns comp 0..L-1][coeff]  (kswitchkeys.h:340).
    int ref_key_digits(void *ctx, int kind, uint64_t index, uint64_t *ndigits)
    {
        REF_TRY
        auto c = static_cast<RefCtx *>(ctx);
        const KSwitchKeys &k =
            kind == 0 ? static_cast<const KSwitchKeys &>(c->rlk) : static_cast<const KSwitchKeys &>(c->glk);
        if (index >= k.data().size())
            return 3;
        *ndigits = k.data()[index].size();
        REF_CATCH
    }
    int ref_key_copy(void *ctx, int kind, uint64_t index, uint64_t *out)
    {
        REF_TRY
        auto c = static_cast<RefCtx *>(ctx);
        const KSwitchKeys &k =
            kind == 0 ? static_cast<const KSwitchKeys &>(c->rlk) : static_cast<const KSwitchKeys &>(c->glk);
        if (index >= k.data().size())
            return 3;
        auto &vec = k.data()[index];
        for (size_t j = 0; j < vec.size(); j++)
        {
            const Ciphertext &kc = vec[j].data();
            size_t words = kc.size() * kc.coeff_modulus_size() * kc.poly_modulus_degree();
            std::memcpy(out, kc.data(), words * sizeof(uint64_t));
            out += words;
        }
        REF_CATCH
    }
    // Overwrite the words of an existing key (same layout as ref_key_copy) with caller-supplied residues:
    // lets a test use a seeded synthetic key, so that golden fixtures need not store key material.
    int ref_key_set(void *ctx, int kind, uint64_t index, const uint64_t *in)
    {
        REF_TRY
        auto c = static_cast<RefCtx *>(ctx);
        KSwitchKeys &k = kind == 0 ? static_cast<KSwitchKeys &>(c->rlk) : static_cast<KSwitchKeys &>(c->glk);
        if (index >= k.data().size())
            return 3;
        auto &vec = k.data()[index];
        for (size_t j = 0; j < vec.size(); j++)
        {
            Ciphertext &kc = vec[j].data();
            size_t words = kc.size() * kc.coeff_modulus_size() * kc.poly_modulus_degree();
            std::memcpy(kc.data(), in, words * sizeof(uint64_t));
            in += words;
        }
        REF_CATCH
    }
    uint64_t ref_galois_elt_from_step(void *ctx, int step)
    {
        auto c = static_cast<RefCtx *>(ctx);
        try
        {
            return c->context->key_context_data()->galois_tool()->get_elt_from_step(step);
        }
        catch (...)
        {
            return 0;
        }
    }

    // ---- ciphertext handles ------------------------------------------------------------
    int ref_ct_create(
        void *ctx, uint64_t chain_index, uint64_t size, int is_ntt, double scale, uint64_t correction_factor,
        const uint64_t *data, void **out)
    {
        REF_TRY
        auto c = static_cast<RefCtx *>(ctx);
        auto l = c->level(chain_index);
        if (!l)
            return 1;
        auto h = std::make_unique<RefCt>();
        h->ct.resize(*c->context, l->parms_id(), size);
        h->ct.is_ntt_form() = is_ntt != 0;
        h->ct.scale() = scale;
        h->ct.correction_factor() = correction_factor;
        if (data)
            std::memcpy(
                h->ct.data(), data,
                size * l->parms().coeff_modulus().size() * l->parms().poly_modulus_degree() * sizeof(uint64_t));
        *out = h.release();
        REF_CATCH
    }
    void ref_ct_destroy(void *ct)
    {
        delete static_cast<RefCt *>(ct);
    }
    int ref_ct_info(
        void *ctx, void *ct, uint64_t *chain_index, uint64_t *size, uint64_t *coeff_modulus_size, int *is_ntt,
        double *scale, uint64_t *correction_factor)
    {
        auto c = static_cast<RefCtx *>(ctx);
        auto &x = static_cast<RefCt *>(ct)->ct;
        auto l = c->context->get_context_data(x.parms_id());
        *chain_index = l ? l->chain_index() : ~uint64_t(0);
        *size = x.size();
        *coeff_modulus_size = x.coeff_modulus_size();
        *is_ntt = x.is_ntt_form();
        *scale = x.scale();
        *correction_factor = x.correction_factor();
        return 0;
    }
    int ref_ct_data(void *ct, uint64_t *out)
    {
        auto &x = static_cast<RefCt *>(ct)->ct;
        std::memcpy(out, x.data(), x.size() * x.coeff_modulus_size() * x.poly_modulus_degree() * sizeof(uint64_t));
        return 0;
    }
    int ref_ct_copy(void *src, void **out)
    {
        REF_TRY
        auto h = std::make_unique<RefCt>();
        h->ct = static_cast<RefCt *>(src)->ct;
        *out = h.release();
        REF_CATCH
    }

    // dst = src through seal::Ciphertext::operator= (a host-side write into dst's buffer)
    int ref_ct_assign(void *dst, void *src)
    {
        REF_TRY
        static_cast<RefCt *>(dst)->ct = static_cast<RefCt *>(src)->ct;
        REF_CATCH
    }

    // ---- Evaluator ops (evaluator.h) ----------------------------------------------------
#define CT(x) (static_cast<RefCt *>(x)->ct)
#define EV (static_cast<RefCtx *>(ctx)->evaluator)
    int ref_multiply_inplace(void *ctx, void *a, void *b)
    {
        REF_TRY EV->multiply_inplace(CT(a), CT(b));
        REF_CATCH
    }
    int ref_square_inplace(void *ctx, void *a)
    {
        REF_TRY EV->square_inplace(CT(a));
        REF_CATCH
    }
    int ref_relinearize_inplace(void *ctx, void *a)
    {
        REF_TRY EV->relinearize_inplace(CT(a), static_cast<RefCtx *>(ctx)->rlk);
        REF_CATCH
    }
    int ref_rescale_to_next_inplace(void *ctx, void *a)
    {
        REF_TRY EV->rescale_to_next_inplace(CT(a));
        REF_CATCH
    }
    int ref_mod_switch_to_next_inplace(void *ctx, void *a)
    {
        REF_TRY EV->mod_switch_to_next_inplace(CT(a));
        REF_CATCH
    }
    int ref_mod_reduce_to_next_inplace(void *ctx, void *a)
    {
        REF_TRY EV->mod_reduce_to_next_inplace(CT(a));
        REF_CATCH
    }
    // the multi-level forms (evaluator.cpp:1451-1473, 1543-1595, 1625-1647): target = the level with this chain index
    int ref_rescale_to_inplace(void *ctx, void *a, uint64_t chain_index)
    {
        REF_TRY auto l = static_cast<RefCtx *>(ctx)->level(chain_index);
        if (!l)
            return 1;
        EV->rescale_to_inplace(CT(a), l->parms_id());
        REF_CATCH
    }
    int ref_mod_switch_to_inplace(void *ctx, void *a, uint64_t chain_index)
    {
        REF_TRY auto l = static_cast<RefCtx *>(ctx)->level(chain_index);
        if (!l)
            return 1;
        EV->mod_switch_to_inplace(CT(a), l->parms_id());
        REF_CATCH
    }
    int ref_mod_reduce_to_inplace(void *ctx, void *a, uint64_t chain_index)
    {
        REF_TRY auto l = static_cast<RefCtx *>(ctx)->level(chain_index);
        if (!l)
            return 1;
        EV->mod_reduce_to_inplace(CT(a), l->parms_id());
        REF_CATCH
    }
    int ref_rotate_vector_inplace(void *ctx, void *a, int steps)
    {
        REF_TRY EV->rotate_vector_inplace(CT(a), steps, static_cast<RefCtx *>(ctx)->glk);
        REF_CATCH
    }
    int ref_rotate_rows_inplace(void *ctx, void *a, int steps)
    {
        REF_TRY EV->rotate_rows_inplace(CT(a), steps, static_cast<RefCtx *>(ctx)->glk);
        REF_CATCH
    }
    int ref_rotate_columns_inplace(void *ctx, void *a)
    {
        REF_TRY EV->rotate_columns_inplace(CT(a), static_cast<RefCtx *>(ctx)->glk);
        REF_CATCH
    }
    int ref_complex_conjugate_inplace(void *ctx, void *a)
    {
        REF_TRY EV->complex_conjugate_inplace(CT(a), static_cast<RefCtx *>(ctx)->glk);
        REF_CATCH
    }
    int ref_apply_galois_inplace(void *ctx, void *a, uint32_t elt)
    {
        REF_TRY EV->apply_galois_inplace(CT(a), elt, static_cast<RefCtx *>(ctx)->glk);
        REF_CATCH
    }
    int ref_transform_to_ntt_inplace(void *ctx, void *a)
    {
        REF_TRY EV->transform_to_ntt_inplace(CT(a));
        REF_CATCH
    }
    int ref_transform_from_ntt_inplace(void *ctx, void *a)
    {
        REF_TRY EV->transform_from_ntt_inplace(CT(a));
        REF_CATCH
    }
    int ref_add_inplace(void *ctx, void *a, void *b)
    {
        REF_TRY EV->add_inplace(CT(a), CT(b));
        REF_CATCH
    }
    int ref_sub_inplace(void *ctx, void *a, void *b)
    {
        REF_TRY EV->sub_inplace(CT(a), CT(b));
        REF_CATCH
    }
    int ref_negate_inplace(void *ctx, void *a)
    {
        REF_TRY EV->negate_inplace(CT(a));
        REF_CATCH
    }

    // ---- plaintext operands (evaluator.cpp:1649-2287): plaintext handles built from raw words --------------
    // chain_index == ~0: coefficient form (parms_id_zero), `count` coefficients modulo t (BFV/BGV);
    // otherwise NTT form at that level, count must be K*N words.
    struct RefPt
    {
        Plaintext pt;
    };
    int ref_pt_create(void *ctx, uint64_t chain_index, uint64_t count, double scale, const uint64_t *data, void **out)
    {
        REF_TRY
        auto c = static_cast<RefCtx *>(ctx);
        auto h = std::make_unique<RefPt>();
        h->pt.resize(count);
        std::memcpy(h->pt.data(), data, count * sizeof(uint64_t));
        if (chain_index != ~uint64_t(0))
        {
            auto l = c->level(chain_index);
            if (!l)
                return 1;
            h->pt.parms_id() = l->parms_id();
        }
        h->pt.scale() = scale;
        *out = h.release();
        REF_CATCH
    }
    void ref_pt_destroy(void *pt)
    {
        delete static_cast<RefPt *>(pt);
    }
    int ref_pt_info(void *ctx, void *pt, uint64_t *chain_index, uint64_t *count, int *is_ntt, double *scale)
    {
        auto c = static_cast<RefCtx *>(ctx);
        auto &x = static_cast<RefPt *>(pt)->pt;
        auto l = x.is_ntt_form() ? c->context->get_context_data(x.parms_id()) : nullptr;
        *chain_index = l ? l->chain_index() : ~uint64_t(0);
        *count = x.coeff_count();
        *is_ntt = x.is_ntt_form();
        *scale = x.scale();
        return 0;
    }
    int ref_pt_data(void *pt, uint64_t *out)
    {
        auto &x = static_cast<RefPt *>(pt)->pt;
        std::memcpy(out, x.data(), x.coeff_count() * sizeof(uint64_t));
        return 0;
    }
#define PT(x) (static_cast<RefPt *>(x)->pt)
    int ref_add_plain_inplace(void *ctx, void *a, void *p)
    {
        REF_TRY EV->add_plain_inplace(CT(a), PT(p));
        REF_CATCH
    }
    int ref_sub_plain_inplace(void *ctx, void *a, void *p)
    {
        REF_TRY EV->sub_plain_inplace(CT(a), PT(p));
        REF_CATCH
    }
    int ref_multiply_plain_inplace(void *ctx, void *a, void *p)
    {
        REF_TRY EV->multiply_plain_inplace(CT(a), PT(p));
        REF_CATCH
    }
    int ref_pt_transform_to_ntt_inplace(void *ctx, void *p, uint64_t chain_index)
    {
        REF_TRY
        auto l = static_cast<RefCtx *>(ctx)->level(chain_index);
        if (!l)
            return 1;
        EV->transform_to_ntt_inplace(PT(p), l->parms_id());
        REF_CATCH
    }
    int ref_pt_mod_switch_to_next_inplace(void *ctx, void *p)
    {
        REF_TRY EV->mod_switch_to_next_inplace(PT(p));
        REF_CATCH
    }
    // add_many / multiply_many / exponentiate (evaluator.cpp:298-350, 1649-1757); result replaces cts[0]
    int ref_add_many(void *ctx, void **cts, uint64_t count)
    {
        REF_TRY
        std::vector<Ciphertext> v;
        for (uint64_t i = 0; i < count; i++)
            v.push_back(CT(cts[i]));
        Ciphertext dest;
        EV->add_many(v, dest);
        CT(cts[0]) = dest;
        REF_CATCH
    }
    int ref_multiply_many(void *ctx, void **cts, uint64_t count)
    {
        REF_TRY
        auto c = static_cast<RefCtx *>(ctx);
        std::vector<Ciphertext> v;
        for (uint64_t i = 0; i < count; i++)
            v.push_back(CT(cts[i]));
        Ciphertext dest;
        EV->multiply_many(v, c->rlk, dest);
        CT(cts[0]) = dest;
        REF_CATCH
    }
    int ref_exponentiate_inplace(void *ctx, void *a, uint64_t exponent)
    {
        REF_TRY
        auto c = static_cast<RefCtx *>(ctx);
        EV->exponentiate_inplace(CT(a), exponent, c->rlk);
        REF_CATCH
    }

    // ---- L1 kernels on raw components (the HEXL seam, ntt.cpp:394-475, polyarithsmallmod.cpp) ----
    // mode: 0 fwd, 1 fwd lazy, 2 inv, 3 inv lazy.  data = `count` consecutive RNS components
    // starting at prime index `first` of level `chain_index` (each N words).
    int ref_ntt(void *ctx, uint64_t chain_index, uint64_t first, uint64_t count, int mode, uint64_t *data)
    {
        REF_TRY
        auto c = static_cast<RefCtx *>(ctx);
        auto l = c->level(chain_index);
        if (!l)
            return 1;
        size_t n = l->parms().poly_modulus_degree();
        for (uint64_t i = 0; i < count; i++)
        {
            auto &t = l->small_ntt_tables()[first + i];
            CoeffIter it(data + i * n);
            switch (mode)
            {
            case 0:
                ntt_negacyclic_harvey(it, t);
                break;
            case 1:
                ntt_negacyclic_harvey_lazy(it, t);
                break;
            case 2:
                inverse_ntt_negacyclic_harvey(it, t);
                break;
            default:
                inverse_ntt_negacyclic_harvey_lazy(it, t);
            }
        }
        REF_CATCH
    }
    int ref_dyadic_product(void *ctx, uint64_t chain_index, uint64_t idx, const uint64_t *a, const uint64_t *b, uint64_t *r)
    {
        REF_TRY
        auto c = static_cast<RefCtx *>(ctx);
        auto l = c->level(chain_index);
        size_t n = l->parms().poly_modulus_degree();
        dyadic_product_coeffmod(ConstCoeffIter(a), ConstCoeffIter(b), n, l->parms().coeff_modulus()[idx], CoeffIter(r));
        REF_CATCH
    }
    // GaloisTool::apply_galois (galois.cpp:148) / apply_galois_ntt (galois.cpp:192) on K comps.
    int ref_apply_galois_raw(void *ctx, uint64_t chain_index, int ntt_form, uint32_t elt, const uint64_t *in, uint64_t *out)
    {
        REF_TRY
        auto c = static_cast<RefCtx *>(ctx);
        auto l = c->level(chain_index);
        size_t n = l->parms().poly_modulus_degree();
        size_t k = l->parms().coeff_modulus().size();
        auto gt = c->context->key_context_data()->galois_tool();
        if (ntt_form)
            gt->apply_galois_ntt(ConstRNSIter(in, n), k, elt, RNSIter(out, n));
        else
            gt->apply_galois(ConstRNSIter(in, n), k, elt, iter(l->parms().coeff_modulus()), RNSIter(out, n));
        REF_CATCH
    }
    // RNSTool stages (rns.cpp): which = 0 fastbconv_m_tilde (q -> Bsk U m~), 1 sm_mrq (Bsk U m~ -> Bsk),
    // 2 fast_floor (q U Bsk -> Bsk), 3 fastbconv_sk (Bsk -> q), 4 divide_and_round_q_last_inplace,
    // 5 divide_and_round_q_last_ntt_inplace.  For 4/5 `out` receives the full in-place buffer.
    int ref_rns_stage(void *ctx, uint64_t chain_index, int which, const uint64_t *in, uint64_t *out)
    {
        REF_TRY
        auto c = static_cast<RefCtx *>(ctx);
        auto l = c->level(chain_index);
        size_t n = l->parms().poly_modulus_degree();
        size_t k = l->parms().coeff_modulus().size();
        auto rt = l->rns_tool();
        auto pool = MemoryManager::GetPool();
        switch (which)
        {
        case 0:
            rt->fastbconv_m_tilde(ConstRNSIter(in, n), RNSIter(out, n), pool);
            break;
        case 1:
            rt->sm_mrq(ConstRNSIter(in, n), RNSIter(out, n), pool);
            break;
        case 2:
            rt->fast_floor(ConstRNSIter(in, n), RNSIter(out, n), pool);
            break;
        case 3:
            rt->fastbconv_sk(ConstRNSIter(in, n), RNSIter(out, n), pool);
            break;
        case 4:
            std::memcpy(out, in, k * n * sizeof(uint64_t));
            rt->divide_and_round_q_last_inplace(RNSIter(out, n), pool);
            break;
        case 5:
            std::memcpy(out, in, k * n * sizeof(uint64_t));
            rt->divide_and_round_q_last_ntt_inplace(RNSIter(out, n), iter(l->small_ntt_tables()), pool);
            break;
        default:
            return 1;
        }
        REF_CATCH
    }

    // ---- encode / encrypt / decrypt (host-side helpers for the semantic round-trip tests) ----
    int ref_ckks_encrypt(void *ctx, const double *values, uint64_t count, double scale, void **out)
    {
        REF_TRY
        auto c = static_cast<RefCtx *>(ctx);
        if (!c->have_pk)
        {
            c->keygen->create_public_key(c->pk);
            c->have_pk = true;
        }
        CKKSEncoder enc(*c->context);
        Plaintext p;
        std::vector<double> v(values, values + count);
        enc.encode(v, scale, p);
        Encryptor e(*c->context, c->pk);
        auto h = std::make_unique<RefCt>();
        e.encrypt(p, h->ct);
        *out = h.release();
        REF_CATCH
    }
    int ref_ckks_decrypt(void *ctx, void *ct, double *values, uint64_t count)
    {
        REF_TRY
        auto c = static_cast<RefCtx *>(ctx);
        Decryptor d(*c->context, c->keygen->secret_key());
        Plaintext p;
        d.decrypt(CT(ct), p);
        CKKSEncoder enc(*c->context);
        std::vector<double> v;
        enc.decode(p, v);
        for (uint64_t i = 0; i < count && i < v.size(); i++)
            values[i] = v[i];
        REF_CATCH
    }
    int ref_batch_encrypt(void *ctx, const uint64_t *values, uint64_t count, void **out)
    {
        REF_TRY
        auto c = static_cast<RefCtx *>(ctx);
        if (!c->have_pk)
        {
            c->keygen->create_public_key(c->pk);
            c->have_pk = true;
        }
        BatchEncoder enc(*c->context);
        Plaintext p;
        std::vector<uint64_t> v(values, values + count);
        v.resize(enc.slot_count());
        enc.encode(v, p);
        Encryptor e(*c->context, c->pk);
        auto h = std::make_unique<RefCt>();
        e.encrypt(p, h->ct);
        *out = h.release();
        REF_CATCH
    }
    int ref_batch_decrypt(void *ctx, void *ct, uint64_t *values, uint64_t count)
    {
        REF_TRY
        auto c = static_cast<RefCtx *>(ctx);
        Decryptor d(*c->context, c->keygen->secret_key());
        Plaintext p;
        d.decrypt(CT(ct), p);
        BatchEncoder enc(*c->context);
        std::vector<uint64_t> v;
        enc.decode(p, v);
        for (uint64_t i = 0; i < count && i < v.size(); i++)
            values[i] = v[i];
        REF_CATCH
    }

    // ---- wire format (Ciphertext::save / load, KSwitchKeys::save / load, Serializable<> seeded forms) ----
    // parms_id of a level (EncryptionParameters::parms_id, encryptionparams.cpp:117-147)
    int ref_ctx_parms_id(void *ctx, uint64_t chain_index, uint64_t *out)
    {
        REF_TRY
        auto l = static_cast<RefCtx *>(ctx)->level(chain_index);
        if (!l)
            return 3;
        std::memcpy(out, l->parms_id().data(), 32);
        REF_CATCH
    }
    int ref_ct_save(void *ct, uint8_t *out, uint64_t cap, uint64_t *bytes)
    {
        REF_TRY
        *bytes = static_cast<uint64_t>(CT(ct).save(reinterpret_cast<seal_byte *>(out), cap, compr_mode_type::none));
        REF_CATCH
    }
    // the same with a compression mode (0 none, 1 zlib; this build has no zstd), for ciphertexts, plaintexts and key sets
    int ref_ct_save_mode(void *ct, int mode, uint8_t *out, uint64_t cap, uint64_t *bytes)
    {
        REF_TRY
        *bytes = static_cast<uint64_t>(CT(ct).save(reinterpret_cast<seal_byte *>(out), cap, static_cast<compr_mode_type>(mode)));
        REF_CATCH
    }
    int ref_keys_save_mode(void *ctx, int kind, int mode, uint8_t *out, uint64_t cap, uint64_t *bytes)
    {
        REF_TRY
        auto c = static_cast<RefCtx *>(ctx);
        const KSwitchKeys &k = kind == 0 ? static_cast<const KSwitchKeys &>(c->rlk) : static_cast<const KSwitchKeys &>(c->glk);
        *bytes = static_cast<uint64_t>(k.save(reinterpret_cast<seal_byte *>(out), cap, static_cast<compr_mode_type>(mode)));
        REF_CATCH
    }
    int ref_ct_load(void *ctx, const uint8_t *in, uint64_t size, int unsafe, void **out, uint64_t *bytes)
    {
        REF_TRY
        auto c = static_cast<RefCtx *>(ctx);
        auto h = std::make_unique<RefCt>();
        if (unsafe)
            *bytes = static_cast<uint64_t>(h->ct.unsafe_load(*c->context, reinterpret_cast<const seal_byte *>(in), size));
        else
            *bytes = static_cast<uint64_t>(h->ct.load(*c->context, reinterpret_cast<const seal_byte *>(in), size));
        *out = h.release();
        REF_CATCH
    }
    // Encryptor::encrypt_zero_symmetric(parms_id) as a Serializable<Ciphertext>: the SEEDED stream (c_1 replaced by its seed,
    // ciphertext.cpp:153-228) when seeded != 0, the full one otherwise
    int ref_encrypt_zero_symmetric_save(void *ctx, uint64_t chain_index, int seeded, uint8_t *out, uint64_t cap, uint64_t *bytes)
    {
        REF_TRY
        auto c = static_cast<RefCtx *>(ctx);
        auto l = c->level(chain_index);
        if (!l)
            return 3;
        Encryptor e(*c->context, c->keygen->secret_key());
        if (seeded)
            *bytes = static_cast<uint64_t>(
                e.encrypt_zero_symmetric(l->parms_id()).save(reinterpret_cast<seal_byte *>(out), cap, compr_mode_type::none));
        else
        {
            Ciphertext ct;
            e.encrypt_zero_symmetric(l->parms_id(), ct);
            *bytes = static_cast<uint64_t>(ct.save(reinterpret_cast<seal_byte *>(out), cap, compr_mode_type::none));
        }
        REF_CATCH
    }
    // kind 0: RelinKeys, 1: GaloisKeys for `elts`.  seeded != 0: a fresh Serializable<> key set is generated and saved in its
    // seeded form; either way the saved stream is loaded back into the context's key object, so that ref_key_copy and the
    // Evaluator calls of this context use exactly the keys the stream describes.
    int ref_keys_save(void *ctx, int kind, int seeded, const uint32_t *elts, uint64_t nelts, uint8_t *out, uint64_t cap, uint64_t *bytes)
    {
        REF_TRY
        auto c = static_cast<RefCtx *>(ctx);
        auto o = reinterpret_cast<seal_byte *>(out);
        std::vector<uint32_t> v(elts, elts + nelts);
        if (kind == 0)
        {
            if (seeded)
                *bytes = static_cast<uint64_t>(c->keygen->create_relin_keys().save(o, cap, compr_mode_type::none));
            else
            {
                if (!c->have_rlk)
                    c->keygen->create_relin_keys(c->rlk);
                *bytes = static_cast<uint64_t>(c->rlk.save(o, cap, compr_mode_type::none));
            }
            c->rlk.load(*c->context, o, *bytes);
            c->have_rlk = true;
        }
        else
        {
            if (seeded)
                *bytes = static_cast<uint64_t>(c->keygen->create_galois_keys(v).save(o, cap, compr_mode_type::none));
            else
            {
                c->keygen->create_galois_keys(v, c->glk);
                *bytes = static_cast<uint64_t>(c->glk.save(o, cap, compr_mode_type::none));
            }
            c->glk.load(*c->context, o, *bytes);
            c->have_glk = true;
        }
        REF_CATCH
    }
    // GaloisTool::get_elts_all of the key level
    int ref_galois_elts_all(void *ctx, uint32_t *out, uint64_t cap, uint64_t *count)
    {
        REF_TRY
        auto c = static_cast<RefCtx *>(ctx);
        auto v = c->context->key_context_data()->galois_tool()->get_elts_all();
        if (v.size() > cap)
            return 1;
        std::copy(v.begin(), v.end(), out);
        *count = v.size();
        REF_CATCH
    }
    // Plaintext::save / load / unsafe_load
    int ref_pt_save_mode(void *pt, int mode, uint8_t *out, uint64_t cap, uint64_t *bytes)
    {
        REF_TRY
        *bytes = static_cast<uint64_t>(
            static_cast<RefPt *>(pt)->pt.save(reinterpret_cast<seal_byte *>(out), cap, static_cast<compr_mode_type>(mode)));
        REF_CATCH
    }
    int ref_pt_save(void *pt, uint8_t *out, uint64_t cap, uint64_t *bytes)
    {
        REF_TRY
        *bytes = static_cast<uint64_t>(static_cast<RefPt *>(pt)->pt.save(reinterpret_cast<seal_byte *>(out), cap, compr_mode_type::none));
        REF_CATCH
    }
    int ref_pt_load(void *ctx, const uint8_t *in, uint64_t size, int unsafe, void **out, uint64_t *bytes)
    {
        REF_TRY
        auto c = static_cast<RefCtx *>(ctx);
        auto h = std::make_unique<RefPt>();
        if (unsafe)
            *bytes = static_cast<uint64_t>(h->pt.unsafe_load(*c->context, reinterpret_cast<const seal_byte *>(in), size));
        else
            *bytes = static_cast<uint64_t>(h->pt.load(*c->context, reinterpret_cast<const seal_byte *>(in), size));
        *out = h.release();
        REF_CATCH
    }
    // CKKSEncoder::encode(values, parms_id(chain_index), scale) / BatchEncoder::encode(values): the plaintexts a client serializes
    int ref_ckks_encode(void *ctx, const double *values, uint64_t count, uint64_t chain_index, double scale, void **out)
    {
        REF_TRY
        auto c = static_cast<RefCtx *>(ctx);
        auto l = c->level(chain_index);
        if (!l)
            return 3;
        CKKSEncoder enc(*c->context);
        auto h = std::make_unique<RefPt>();
        std::vector<double> v(values, values + count);
        enc.encode(v, l->parms_id(), scale, h->pt);
        *out = h.release();
        REF_CATCH
    }
    int ref_batch_encode(void *ctx, const uint64_t *values, uint64_t count, void **out)
    {
        REF_TRY
        auto c = static_cast<RefCtx *>(ctx);
        BatchEncoder enc(*c->context);
        auto h = std::make_unique<RefPt>();
        std::vector<uint64_t> v(values, values + count);
        v.resize(enc.slot_count());
        enc.encode(v, h->pt);
        *out = h.release();
        REF_CATCH
    }
    // ---- decryption: the secret key's words / stream and Decryptor::decrypt as a Plaintext handle ----
    int ref_secret_key_copy(void *ctx, uint64_t *out)
    {
        REF_TRY
        auto c = static_cast<RefCtx *>(ctx);
        const Plaintext &sk = c->keygen->secret_key().data();
        std::memcpy(out, sk.data(), sk.coeff_count() * sizeof(uint64_t));
        REF_CATCH
    }
    int ref_secret_key_save(void *ctx, uint8_t *out, uint64_t cap, uint64_t *bytes)
    {
        REF_TRY
        auto c = static_cast<RefCtx *>(ctx);
        *bytes = static_cast<uint64_t>(c->keygen->secret_key().save(reinterpret_cast<seal_byte *>(out), cap, compr_mode_type::none));
        REF_CATCH
    }
    int ref_decrypt(void *ctx, void *ct, void **out)
    {
        REF_TRY
        auto c = static_cast<RefCtx *>(ctx);
        Decryptor d(*c->context, c->keygen->secret_key());
        auto h = std::make_unique<RefPt>();
        d.decrypt(CT(ct), h->pt);
        *out = h.release();
        REF_CATCH
    }
    // Encryptor::encrypt_symmetric(plain) saved: seeded != 0 -> the Serializable<Ciphertext> form, else the full ciphertext
    int ref_encrypt_symmetric_save(void *ctx, void *pt, int seeded, uint8_t *out, uint64_t cap, uint64_t *bytes)
    {
        REF_TRY
        auto c = static_cast<RefCtx *>(ctx);
        Encryptor e(*c->context, c->keygen->secret_key());
        const Plaintext &p = static_cast<RefPt *>(pt)->pt;
        if (seeded)
            *bytes = static_cast<uint64_t>(e.encrypt_symmetric(p).save(reinterpret_cast<seal_byte *>(out), cap, compr_mode_type::none));
        else
        {
            Ciphertext ct;
            e.encrypt_symmetric(p, ct);
            *bytes = static_cast<uint64_t>(ct.save(reinterpret_cast<seal_byte *>(out), cap, compr_mode_type::none));
        }
        REF_CATCH
    }
    // BatchEncoder::decode of a plaintext handle (unsigned / signed)
    int ref_batch_decode(void *ctx, void *pt, int is_signed, uint64_t *out)
    {
        REF_TRY
        auto c = static_cast<RefCtx *>(ctx);
        BatchEncoder enc(*c->context);
        const Plaintext &p = static_cast<RefPt *>(pt)->pt;
        if (is_signed)
        {
            std::vector<int64_t> v;
            enc.decode(p, v);
            std::memcpy(out, v.data(), v.size() * 8);
        }
        else
        {
            std::vector<uint64_t> v;
            enc.decode(p, v);
            std::memcpy(out, v.data(), v.size() * 8);
        }
        REF_CATCH
    }
    int ref_batch_encode_signed(void *ctx, const int64_t *values, uint64_t count, void **out)
    {
        REF_TRY
        auto c = static_cast<RefCtx *>(ctx);
        BatchEncoder enc(*c->context);
        auto h = std::make_unique<RefPt>();
        std::vector<int64_t> v(values, values + count);
        enc.encode(v, h->pt);
        *out = h.release();
        REF_CATCH
    }
    // public-key encryption: PublicKey words, Encryptor::encrypt_zero(parms_id) / encrypt(plain) saved
    int ref_public_key_copy(void *ctx, uint64_t *out)
    {
        REF_TRY
        auto c = static_cast<RefCtx *>(ctx);
        if (!c->have_pk)
        {
            c->keygen->create_public_key(c->pk);
            c->have_pk = true;
        }
        const Ciphertext &k = c->pk.data();
        std::memcpy(out, k.data(), k.size() * k.coeff_modulus_size() * k.poly_modulus_degree() * sizeof(uint64_t));
        REF_CATCH
    }
    int ref_encrypt_asymmetric_save(void *ctx, void *pt /* null: encrypt_zero at chain_index */, uint64_t chain_index, uint8_t *out,
                                    uint64_t cap, uint64_t *bytes)
    {
        REF_TRY
        auto c = static_cast<RefCtx *>(ctx);
        if (!c->have_pk)
        {
            c->keygen->create_public_key(c->pk);
            c->have_pk = true;
        }
        Encryptor e(*c->context, c->pk);
        Ciphertext ct;
        if (pt)
            e.encrypt(static_cast<RefPt *>(pt)->pt, ct);
        else
        {
            auto l = c->level(chain_index);
            if (!l)
                return 3;
            e.encrypt_zero(l->parms_id(), ct);
        }
        *bytes = static_cast<uint64_t>(ct.save(reinterpret_cast<seal_byte *>(out), cap, compr_mode_type::none));
        REF_CATCH
    }
    // CKKSEncoder::encode of complex values / decode of a plaintext handle (real or complex)
    int ref_ckks_encode_complex(void *ctx, const double *re_im, uint64_t count, uint64_t chain_index, double scale, void **out)
    {
        REF_TRY
        auto c = static_cast<RefCtx *>(ctx);
        auto l = c->level(chain_index);
        if (!l)
            return 3;
        CKKSEncoder enc(*c->context);
        auto h = std::make_unique<RefPt>();
        std::vector<std::complex<double>> v(count);
        for (uint64_t i = 0; i < count; i++)
            v[i] = { re_im[2 * i], re_im[2 * i + 1] };
        enc.encode(v, l->parms_id(), scale, h->pt);
        *out = h.release();
        REF_CATCH
    }
    int ref_ckks_decode(void *ctx, void *pt, int want_complex, double *out)
    {
        REF_TRY
        auto c = static_cast<RefCtx *>(ctx);
        CKKSEncoder enc(*c->context);
        const Plaintext &p = static_cast<RefPt *>(pt)->pt;
        if (want_complex)
        {
            std::vector<std::complex<double>> v;
            enc.decode(p, v);
            std::memcpy(out, v.data(), v.size() * 16);
        }
        else
        {
            std::vector<double> v;
            enc.decode(p, v);
            std::memcpy(out, v.data(), v.size() * 8);
        }
        REF_CATCH
    }
    // CKKSEncoder::encode(double value, ...) / encode(int64_t value, ...): one value in every slot
    int ref_ckks_encode_value(void *ctx, double value, int is_integer, int64_t ivalue, uint64_t chain_index, double scale, void **out)
    {
        REF_TRY
        auto c = static_cast<RefCtx *>(ctx);
        auto l = c->level(chain_index);
        if (!l)
            return 3;
        CKKSEncoder enc(*c->context);
        auto h = std::make_unique<RefPt>();
        if (is_integer)
            enc.encode(ivalue, l->parms_id(), h->pt);
        else
            enc.encode(value, l->parms_id(), scale, h->pt);
        *out = h.release();
        REF_CATCH
    }
    int ref_noise_budget(void *ctx, void *ct, int *bits)
    {
        REF_TRY
        auto c = static_cast<RefCtx *>(ctx);
        Decryptor d(*c->context, c->keygen->secret_key());
        *bits = d.invariant_noise_budget(CT(ct));
        REF_CATCH
    }
    // KSwitchKeys::load / unsafe_load into a scratch object (error-class checks)
    int ref_keys_load(void *ctx, const uint8_t *in, uint64_t size, int unsafe, uint64_t *bytes)
    {
        REF_TRY
        auto c = static_cast<RefCtx *>(ctx);
        KSwitchKeys k;
        if (unsafe)
            *bytes = static_cast<uint64_t>(k.unsafe_load(*c->context, reinterpret_cast<const seal_byte *>(in), size));
        else
            *bytes = static_cast<uint64_t>(k.load(*c->context, reinterpret_cast<const seal_byte *>(in), size));
        REF_CATCH
    }
    // the context's relinearization (kind 0) / Galois (kind 1) key object := the stream (RelinKeys::load / GaloisKeys::load)
    int ref_keys_install(void *ctx, int kind, const uint8_t *in, uint64_t size, uint64_t *bytes)
    {
        REF_TRY
        auto c = static_cast<RefCtx *>(ctx);
        if (kind == 0)
            *bytes = static_cast<uint64_t>(c->rlk.load(*c->context, reinterpret_cast<const seal_byte *>(in), size));
        else
            *bytes = static_cast<uint64_t>(c->glk.load(*c->context, reinterpret_cast<const seal_byte *>(in), size));
        REF_CATCH
    }
    // Serializable<PublicKey>::save (KeyGenerator::create_public_key() without destination): the seeded form, c_1 as its seed
    int ref_public_key_save_seeded(void *ctx, uint8_t *out, uint64_t cap, uint64_t *bytes)
    {
        REF_TRY
        auto c = static_cast<RefCtx *>(ctx);
        *bytes = static_cast<uint64_t>(c->keygen->create_public_key().save(reinterpret_cast<seal_byte *>(out), cap, compr_mode_type::none));
        REF_CATCH
    }
    // PublicKey::load of a stream (seeded or full) -> the key's words [2][L][N]
    int ref_public_key_load_words(void *ctx, const uint8_t *in, uint64_t size, uint64_t *out_words)
    {
        REF_TRY
        auto c = static_cast<RefCtx *>(ctx);
        PublicKey pk;
        pk.load(*c->context, reinterpret_cast<const seal_byte *>(in), size);
        const Ciphertext &d = pk.data();
        std::memcpy(out_words, d.data(), d.size() * d.coeff_modulus_size() * d.poly_modulus_degree() * sizeof(uint64_t));
        REF_CATCH
    }
    // PublicKey::save: a key-level ciphertext in the Ciphertext wire format
    int ref_public_key_save(void *ctx, uint8_t *out, uint64_t cap, uint64_t *bytes)
    {
        REF_TRY
        auto c = static_cast<RefCtx *>(ctx);
        if (!c->have_pk)
        {
            c->keygen->create_public_key(c->pk);
            c->have_pk = true;
        }
        *bytes = static_cast<uint64_t>(c->pk.save(reinterpret_cast<seal_byte *>(out), cap, compr_mode_type::none));
        REF_CATCH
    }
    // ---- container surface (tests/container_cases.py): per-level constants of SEALContext::ContextData, the qualifiers, the stream
    // of EncryptionParameters::save, and Ciphertext::reserve / resize bookkeeping
    // which: 0 total_coeff_modulus, 1 coeff_div_plain_modulus (operands), 2 plain_upper_half_increment, 3 upper_half_threshold,
    //        4 upper_half_increment.  *count = 0 when the reference did not compute it for these parameters.
    int ref_ctx_data_words(void *ctx, uint64_t chain_index, int which, uint64_t *out, uint64_t *count)
    {
        REF_TRY
        auto c = static_cast<RefCtx *>(ctx);
        auto l = c->level(chain_index);
        if (!l)
            return 1;
        const size_t k = l->parms().coeff_modulus().size();
        *count = 0;
        const uint64_t *src = nullptr;
        std::vector<uint64_t> tmp;
        switch (which)
        {
        case 0:
            src = l->total_coeff_modulus();
            break;
        case 1:
            if (l->coeff_div_plain_modulus())
            {
                for (size_t i = 0; i < k; i++)
                    tmp.push_back(l->coeff_div_plain_modulus()[i].operand);
                src = tmp.data();
            }
            break;
        case 2:
            src = l->plain_upper_half_increment();
            break;
        case 3:
            src = l->upper_half_threshold();
            break;
        case 4:
            src = l->upper_half_increment();
            break;
        default:
            return 1;
        }
        if (src)
        {
            *count = k;
            for (size_t i = 0; i < k; i++)
                out[i] = src[i];
        }
        REF_CATCH
    }
    // out[0..7] = total_coeff_modulus_bit_count, using_fft, using_ntt, using_batching, using_fast_plain_lift,
    //             using_descending_modulus_chain, sec_level, parameters_set; *puht = plain_upper_half_threshold
    int ref_ctx_qualifiers(void *ctx, uint64_t chain_index, int *out, uint64_t *puht)
    {
        REF_TRY
        auto c = static_cast<RefCtx *>(ctx);
        auto l = c->level(chain_index);
        if (!l)
            return 1;
        const auto q = l->qualifiers();
        out[0] = l->total_coeff_modulus_bit_count();
        out[1] = q.using_fft;
        out[2] = q.using_ntt;
        out[3] = q.using_batching;
        out[4] = q.using_fast_plain_lift;
        out[5] = q.using_descending_modulus_chain;
        out[6] = static_cast<int>(q.sec_level);
        out[7] = q.parameters_set();
        *puht = l->plain_upper_half_threshold();
        REF_CATCH
    }
    int ref_parms_save(void *ctx, uint64_t chain_index, int mode, uint8_t *out, uint64_t cap, uint64_t *bytes)
    {
        REF_TRY
        auto c = static_cast<RefCtx *>(ctx);
        auto l = c->level(chain_index);
        if (!l)
            return 1;
        *bytes = static_cast<uint64_t>(l->parms().save(reinterpret_cast<seal_byte *>(out), cap, static_cast<compr_mode_type>(mode)));
        REF_CATCH
    }
    // EncryptionParameters::load of a stream: scheme, degree, plain modulus and the primes (count in / out)
    int ref_parms_load(const uint8_t *in, uint64_t size, int *scheme, uint64_t *degree, uint64_t *plain, uint64_t *primes, uint64_t *count)
    {
        REF_TRY
        EncryptionParameters p;
        p.load(reinterpret_cast<const seal_byte *>(in), size);
        *scheme = static_cast<int>(p.scheme());
        *degree = p.poly_modulus_degree();
        *plain = p.plain_modulus().value();
        if (p.coeff_modulus().size() > *count)
            return 1;
        *count = p.coeff_modulus().size();
        for (size_t i = 0; i < p.coeff_modulus().size(); i++)
            primes[i] = p.coeff_modulus()[i].value();
        REF_CATCH
    }
    // op: 0 reserve(context, parms_id of chain_index, n)  1 reserve(n)  2 resize(context, parms_id of chain_index, n)  3 resize(n)  4 release()
    // out[0..3] = size, size_capacity, coeff_modulus_size, poly_modulus_degree afterwards
    int ref_ct_container_op(void *ctx, void *ct, int op, uint64_t chain_index, uint64_t n, uint64_t *out)
    {
        REF_TRY
        auto c = static_cast<RefCtx *>(ctx);
        Ciphertext &x = CT(ct);
        auto l = c->level(chain_index);
        switch (op)
        {
        case 0:
            if (!l)
                return 1;
            x.reserve(*c->context, l->parms_id(), n);
            break;
        case 1:
            x.reserve(n);
            break;
        case 2:
            if (!l)
                return 1;
            x.resize(*c->context, l->parms_id(), n);
            break;
        case 3:
            x.resize(n);
            break;
        case 4:
            x.release();
            break;
        default:
            return 1;
        }
        out[0] = x.size();
        out[1] = x.size_capacity();
        out[2] = x.coeff_modulus_size();
        out[3] = x.poly_modulus_degree();
        REF_CATCH
    }

    // number of key slots (KSwitchKeys::data().size()) and which are populated
    int ref_key_slots(void *ctx, int kind, uint64_t *slots)
    {
        REF_TRY
        auto c = static_cast<RefCtx *>(ctx);
        const KSwitchKeys &k = kind == 0 ? static_cast<const KSwitchKeys &>(c->rlk) : static_cast<const KSwitchKeys &>(c->glk);
        *slots = k.data().size();
        REF_CATCH
    }

    // ---- CPU baseline: time the reference Evaluator on the host cores ---------------------
    // Synthetic size-2 ciphertexts at the first data level, every RNS component uniform in
    // [0, q_i) from mt19937_64(0x5EA1 + index) (restating BMEnv::randomize_ct_*, native/bench/bench.h:195-270).
    // pipeline: 0 = CKKS multiply + relinearize + rescale_to_next (north-star),
    //           1 = BFV multiply + relinearize + mod_switch_to_next,
    //           2 = rotate_vector(1) (ckks) / rotate_rows(1) (bfv) [+ rescale for ckks],
    //           3 = forward+inverse NTT of all comps of one size-2 ciphertext.
    // Each of `threads` threads runs one untimed warm-up pass (pool first-touch, bench.cpp:28-33),
    // then `reps` timed passes on its own ciphertexts.  Returns seconds for the timed region
    // (max over threads) in *seconds; processed ciphertexts = threads * reps.
    int ref_time_pipeline(void *ctx, int pipeline, int threads, int reps, double *seconds)
    {
        REF_TRY
        auto c = static_cast<RefCtx *>(ctx);
        auto &context = *c->context;
        auto first = context.first_context_data();
        auto &mods = first->parms().coeff_modulus();
        size_t n = first->parms().poly_modulus_degree();
        size_t k = mods.size();
        bool ckks = c->scheme == scheme_type::ckks;
        double scale = ckks ? std::pow(2.0, mods.back().bit_count() / 2 - 1) : 1.0;
        auto make_ct = [&](uint64_t seed) {
            Ciphertext ct;
            ct.resize(context, first->parms_id(), 2);
            std::mt19937_64 rng(seed);
            for (size_t p = 0; p < 2; p++)
                for (size_t i = 0; i < k; i++)
                {
                    std::uniform_int_distribution<uint64_t> dist(0, mods[i].value() - 1);
                    uint64_t *d = ct.data(p) + i * n;
                    for (size_t j = 0; j < n; j++)
                        d[j] = dist(rng);
                }
            ct.is_ntt_form() = ckks;
            ct.scale() = scale;
            return ct;
        };
        // All workers build their inputs and run one untimed pass, meet at a start barrier, and the wall clock runs from
        // the barrier's release to the last worker's finish.  Each worker draws from its own MemoryPoolHandle::New()
        // (thread-local pools are allowed), so the threads do not serialise on the global pool's lock.
        std::vector<int> errs(threads, 0);
        std::mutex mu;
        std::condition_variable cv;
        int ready = 0;
        bool go = false;
        std::chrono::steady_clock::time_point t_start, t_end;
        std::vector<std::chrono::steady_clock::time_point> ends(threads);
        auto worker = [&](int tid) {
            try
            {
                MemoryPoolHandle pool = MemoryPoolHandle::New();
                Ciphertext a0 = make_ct(0x5EA1 + 2 * tid), b0 = make_ct(0x5EA1 + 2 * tid + 1);
                Ciphertext a(a0, pool), b(b0, pool), w(pool);
                auto pass = [&]() {
                    w = a;
                    switch (pipeline)
                    {
                    case 0:
                        c->evaluator->multiply_inplace(w, b, pool);
                        c->evaluator->relinearize_inplace(w, c->rlk, pool);
                        c->evaluator->rescale_to_next_inplace(w, pool);
                        break;
                    case 1:
                        c->evaluator->multiply_inplace(w, b, pool);
                        c->evaluator->relinearize_inplace(w, c->rlk, pool);
                        c->evaluator->mod_switch_to_next_inplace(w, pool);
                        break;
                    case 2:
                        if (ckks)
                        {
                            c->evaluator->rotate_vector_inplace(w, 1, c->glk, pool);
                            c->evaluator->rescale_to_next_inplace(w, pool);
                        }
                        else
                            c->evaluator->rotate_rows_inplace(w, 1, c->glk, pool);
                        break;
                    case 4:
                    {
                        // a chained program (CKKS): multiply, relinearize, rescale, rotate level after level on the same
                        // object, the second operand following by mod_switch; the result is looked at once at the end
                        Ciphertext bb = b;
                        while (w.coeff_modulus_size() > 2)
                        {
                            c->evaluator->multiply_inplace(w, bb, pool);
                            c->evaluator->relinearize_inplace(w, c->rlk, pool);
                            c->evaluator->rescale_to_next_inplace(w, pool);
                            c->evaluator->rotate_vector_inplace(w, 1, c->glk, pool);
                            c->evaluator->mod_switch_to_next_inplace(bb, pool);
                            w.scale() = scale; // synthetic words: keep the bookkeeping value in range level after level
                            bb.scale() = scale;
                        }
                        volatile std::uint64_t sink = w.data()[0] + w.data(1)[n - 1];
                        (void)sink;
                        break;
                    }
                    default:
                        if (ckks)
                        {
                            c->evaluator->transform_from_ntt_inplace(w);
                            c->evaluator->transform_to_ntt_inplace(w);
                        }
                        else
                        {
                            c->evaluator->transform_to_ntt_inplace(w);
                            c->evaluator->transform_from_ntt_inplace(w);
                        }
                    }
                };
                pass(); // warm-up (fills this worker's pool)
                {
                    std::unique_lock<std::mutex> lk(mu);
                    if (++ready == threads)
                    {
                        t_start = std::chrono::steady_clock::now();
                        go = true;
                        cv.notify_all();
                    }
                    else
                        cv.wait(lk, [&] { return go; });
                }
                for (int r = 0; r < reps; r++)
                    pass();
                ends[tid] = std::chrono::steady_clock::now();
            }
            catch (...)
            {
                errs[tid] = 1;
                std::unique_lock<std::mutex> lk(mu);
                if (++ready == threads && !go)
                {
                    t_start = std::chrono::steady_clock::now();
                    go = true;
                    cv.notify_all();
                }
            }
        };
        std::vector<std::thread> workers;
        for (int t = 0; t < threads; t++)
            workers.emplace_back(worker, t);
        for (auto &t : workers)
            t.join();
        double mx = 0;
        for (int t = 0; t < threads; t++)
        {
            if (errs[t])
                return 4;
            mx = std::max(mx, std::chrono::duration<double>(ends[t] - t_start).count());
        }
        *seconds = mx;
        REF_CATCH
    }
}
