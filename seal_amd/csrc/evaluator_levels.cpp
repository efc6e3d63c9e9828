// Evaluator, part 3: modulus switching / rescaling and the Galois automorphisms
#include "evaluator_common.h"
#include <atomic>

namespace sealhip
{
    // ---- modulus switching (evaluator.cpp:1201-1647)
    void Evaluator::mod_switch_scale_to_next(Ciphertext &e) const
    {
        StreamScope pool_scope(stream_); // pool traffic of this call is ordered on the evaluator's stream whoever calls (pool.h)
        const Scheme scheme = context_.scheme();
        if (scheme == Scheme::bfv && e.is_ntt_form())
            throw std::invalid_argument("BFV encrypted cannot be in NTT form");
        if (scheme == Scheme::ckks && !e.is_ntt_form())
            throw std::invalid_argument("CKKS encrypted must be in NTT form");
        if (scheme == Scheme::bgv && !e.is_ntt_form())
            throw std::invalid_argument("BGV encrypted must be in NTT form");
        const Level &lvl = *e.level();
        const Level *next = context_.next_level(lvl);
        double destination_scale = 1.0;
        if (scheme == Scheme::ckks)
        {
            if (!scale_within_bounds(e.scale(), lvl))
                throw std::invalid_argument("scale out of bounds");
            destination_scale = e.scale() / static_cast<double>(context_.coeff_modulus()[lvl.K - 1]);
            if (!scale_within_bounds(destination_scale, *next))
                throw std::invalid_argument("scale out of bounds");
        }
        const unsigned K = lvl.K;
        const size_t N = context_.n();
        if (scheme == Scheme::ckks && e.lazy_ && e.lazy_->owner == this && e.size() == 2 && K >= 2 &&
            ntt2_supports(context_.log_n()))
        {
            // the key switch that produced e left its mod-down undone (LazyTail): both rounding divisions in one pass
            const LazyTail t = detach_tail(e);
            try
            {
                switch_key_finish_rescale(e, t.acc, next, destination_scale, t.with_addend);
            }
            catch (...)
            {
                // the folded pass works in place on e's planes: after a failure they are neither the old nor the new ciphertext.
                // Leave an EMPTY object behind (size 0: every later use is rejected) rather than words that look valid
                DevicePool::global().free_words(t.acc, stream_);
                e.release();
                throw;
            }
            DevicePool::global().free_words(t.acc, stream_);
            return;
        }
        if (scheme == Scheme::bfv && e.lazy_ && e.lazy_->owner == this && e.size() == 2 && K >= 2)
        {
            // BFV: the key switch that produced e left its mod-down undone: one element-wise pass does it and this division
            const LazyTail t = detach_tail(e);
            try
            {
                switch_key_finish_modswitch_bfv(e, t.acc, next);
            }
            catch (...)
            {
                // (e's planes are untouched by the folded pass - it writes a new slab - but the sums are gone: leave an EMPTY object)
                DevicePool::global().free_words(t.acc, stream_);
                e.release();
                throw;
            }
            DevicePool::global().free_words(t.acc, stream_);
            return;
        }
        const size_t items = e.size() * e.batch();
        const unsigned n_log = (unsigned)context_.log_n();
        const ModDesc *mods = context_.dev_mods();
        size_t words = items * (K - 1) * N;
        uint64_t *out = DevicePool::global().alloc_words(words);
        try
        {
            if (scheme == Scheme::bfv)
            {
                ck(k_bfv_modswitch(mods, lvl.dev, e.data(), out, n_log, items, stream_), "bfv modswitch");
            }
            else if (scheme == Scheme::bgv)
            {
                // mod_t_and_divide_q_last_ntt_inplace (rns.cpp:1193-1236)
                const NttTables &tb = context_.ntt_tables();
                uint64_t *last = e.data() + (size_t)(K - 1) * N;
                ck(ntt_inverse(tb, plain_batch(last, (size_t)K * N, 1, (unsigned)items, K - 1), 0, stream_), "bgv modswitch intt last");
                Scratch delta(words);
                ck(k_bgv_delta(mods, host::make_mod(context_.plain_modulus()), lvl.dev.inv_q_last_mod_t, lvl.dev.q_last_mod_q, last,
                               (size_t)K * N, delta.p, n_log, K - 1, items, stream_),
                   "bgv modswitch delta");
                bgv_correct_and_combine(delta, e.data(), (size_t)K * N, lvl.dev.inv_q_last_mod_q, K - 1, items, out, nullptr,
                                        (size_t)(K - 1) * N, 1);
            }
            else
            {
                // divide_and_round_q_last_ntt_inplace (rns.cpp:830-901)
                const NttTables &tb = context_.ntt_tables();
                uint64_t *last = e.data() + (size_t)(K - 1) * N;
                ck(ntt_inverse(tb, plain_batch(last, (size_t)K * N, 1, (unsigned)items, K - 1), 0, stream_), "rescale intt last");
                Scratch tt(ntt2_supports(context_.log_n()) ? 1 : words);
                NttBatch b{};
                b.data = tt.p;
                b.outer_stride = (size_t)(K - 1) * N;
                b.ncomp = K - 1;
                b.nouter = (unsigned)items;
                b.comp_prime = nullptr;
                b.prime_first = 0;
                b.src = last;
                b.src_outer_stride = (size_t)K * N;
                b.src_ncomp = 1;
                b.src_mode = 2;
                b.src_half = lvl.dev.half_q_last;
                b.src_q = lvl.dev.q_last;
                b.src_fix = lvl.dev.round_fix;
                if (ntt2_supports(context_.log_n()))
                {
                    b.data = nullptr;
                    b.epi = 1;
                    b.epi_a = e.data();
                    b.epi_a_stride = (size_t)K * N;
                    b.epi_mul = lvl.dev.inv_q_last_mod_q;
                    b.epi_out0 = out;
                    b.epi_out1 = nullptr;
                    b.epi_out_stride = (size_t)(K - 1) * N;
                    ck(ntt_forward(tb, b, 1, stream_), "rescale ntt correction + combine");
                }
                else
                {
                    ck(ntt_forward(tb, b, 1, stream_), "rescale ntt correction");
                    ck(k_rescale_combine(mods, lvl.dev.inv_q_last_mod_q, e.data(), tt.p, out, n_log, K, items, stream_), "rescale combine");
                }
            }
        }
        catch (...)
        {
            DevicePool::global().free_words(out);
            throw;
        }
        size_t size = e.size();
        e.adopt(next, size, out, words);
        if (scheme == Scheme::ckks)
            e.scale() = destination_scale;
        else if (scheme == Scheme::bgv)
            // evaluator.cpp:1286-1292
            e.correction_factor() = host::mulmod(e.correction_factor(), lvl.dev.inv_q_last_mod_t, context_.plain_modulus());
    }

    void Evaluator::mod_switch_drop_to_next(Ciphertext &e) const
    {
        StreamScope pool_scope(stream_); // pool traffic of this call is ordered on the evaluator's stream whoever calls (pool.h)
        const Scheme scheme = context_.scheme();
        if (scheme == Scheme::bfv && e.is_ntt_form())
            throw std::invalid_argument("BFV encrypted cannot be in NTT form");
        if (scheme == Scheme::ckks && !e.is_ntt_form())
            throw std::invalid_argument("CKKS encrypted must be in NTT form");
        if (scheme == Scheme::bgv && !e.is_ntt_form())
            throw std::invalid_argument("BGV encrypted must be in NTT form");
        const Level &lvl = *e.level();
        const Level *next = context_.next_level(lvl);
        if (!scale_within_bounds(e.scale(), *next))
            throw std::invalid_argument("scale out of bounds");
        const unsigned K = lvl.K;
        const size_t items = e.size() * e.batch();
        size_t words = items * (K - 1) * context_.n();
        uint64_t *out = DevicePool::global().alloc_words(words);
        ck(k_drop_last(e.data(), out, (unsigned)context_.log_n(), K, items, stream_), "drop last");
        size_t size = e.size();
        e.adopt(next, size, out, words);
    }

    void Evaluator::mod_switch_to_next_inplace(Ciphertext &e) const
    {
        StreamScope pool_scope(stream_); // pool traffic of this call is ordered on the evaluator's stream whoever calls (pool.h)
        check_valid(e, "encrypted");
        if (e.level() == &context_.last_level())
            throw std::invalid_argument("end of modulus switching chain reached");
        switch (context_.scheme())
        {
        case Scheme::bfv:
            mod_switch_scale_to_next(e);
            break;
        case Scheme::ckks:
            mod_switch_drop_to_next(e);
            break;
        case Scheme::bgv:
            mod_switch_scale_to_next(e);
            break;
        default:
            throw std::invalid_argument("unsupported scheme");
        }
        throw_if_transparent(e);
    }

    void Evaluator::mod_switch_to_inplace(Ciphertext &e, const uint64_t *parms_id) const
    {
        StreamScope pool_scope(stream_); // pool traffic of this call is ordered on the evaluator's stream whoever calls (pool.h)
        const Level *target = context_.level_by_parms_id(parms_id);
        if (&e.context() != &context_ || !e.level())
            throw std::invalid_argument("encrypted is not valid for encryption parameters");
        if (!target)
            throw std::invalid_argument("parms_id is not valid for encryption parameters");
        if (e.level()->chain_index < target->chain_index)
            throw std::invalid_argument("cannot switch to higher level modulus");
        while (e.level() != target)
            mod_switch_to_next_inplace(e);
    }

    void Evaluator::rescale_to_next_inplace(Ciphertext &e) const
    {
        StreamScope pool_scope(stream_); // pool traffic of this call is ordered on the evaluator's stream whoever calls (pool.h)
        check_valid(e, "encrypted");
        if (e.level() == &context_.last_level())
            throw std::invalid_argument("end of modulus switching chain reached");
        switch (context_.scheme())
        {
        case Scheme::bfv:
        case Scheme::bgv:
            throw std::invalid_argument("unsupported operation for scheme type");
        case Scheme::ckks:
            mod_switch_scale_to_next(e);
            break;
        default:
            throw std::invalid_argument("unsupported scheme");
        }
        throw_if_transparent(e);
    }

    void Evaluator::rescale_to_inplace(Ciphertext &e, const uint64_t *parms_id) const
    {
        StreamScope pool_scope(stream_); // pool traffic of this call is ordered on the evaluator's stream whoever calls (pool.h)
        check_valid(e, "encrypted");
        const Level *target = context_.level_by_parms_id(parms_id);
        if (!target)
            throw std::invalid_argument("parms_id is not valid for encryption parameters");
        if (e.level()->chain_index < target->chain_index)
            throw std::invalid_argument("cannot switch to higher level modulus");
        if (context_.scheme() != Scheme::ckks)
            throw std::invalid_argument("unsupported operation for scheme type");
        while (e.level() != target)
            mod_switch_scale_to_next(e);
        throw_if_transparent(e);
    }

    void Evaluator::mod_reduce_to_next_inplace(Ciphertext &e) const
    {
        StreamScope pool_scope(stream_); // pool traffic of this call is ordered on the evaluator's stream whoever calls (pool.h)
        check_valid(e, "encrypted");
        if (e.level() == &context_.last_level())
            throw std::invalid_argument("end of modulus switching chain reached");
        mod_switch_drop_to_next(e);
        throw_if_transparent(e);
    }

    // evaluator.cpp:1625-1647
    void Evaluator::mod_reduce_to_inplace(Ciphertext &e, const uint64_t *parms_id) const
    {
        StreamScope pool_scope(stream_); // pool traffic of this call is ordered on the evaluator's stream whoever calls (pool.h)
        if (&e.context() != &context_ || !e.level())
            throw std::invalid_argument("encrypted is not valid for encryption parameters");
        const Level *target = context_.level_by_parms_id(parms_id);
        if (!target)
            throw std::invalid_argument("parms_id is not valid for encryption parameters");
        if (e.level()->chain_index < target->chain_index)
            throw std::invalid_argument("cannot switch to higher level modulus");
        while (e.level() != target)
            mod_reduce_to_next_inplace(e);
    }

    // ---- Galois automorphisms and rotations (evaluator.cpp:2384-2559, evaluator.h:1072-1375)
    namespace
    {
        std::atomic<uint64_t> g_galois_gathered{ 0 }, g_galois_permuted{ 0 };
    }
    void galois_path_stats(uint64_t &gathered, uint64_t &permuted)
    {
        gathered = g_galois_gathered.load();
        permuted = g_galois_permuted.load();
    }
    void Evaluator::apply_galois_inplace(Ciphertext &e, uint32_t galois_elt, const KSwitchKeys &galois_keys) const
    {
        apply_galois(e, galois_elt, galois_keys, e);
    }
    // evaluator.h:1072-1087 (apply_galois): destination = encrypted; apply_galois_inplace(destination).  The permuted polynomials go
    // into a new slab either way, so the out-of-place form reads `encrypted` where it lies: no copy of the operand
    void Evaluator::apply_galois(const Ciphertext &e, uint32_t galois_elt, const KSwitchKeys &galois_keys, Ciphertext &dest) const
    {
        StreamScope pool_scope(stream_); // pool traffic of this call is ordered on the evaluator's stream whoever calls (pool.h)
        if (&dest != &e && (&dest.context() != &context_ || dest.batch() != e.batch()))
        {
            dest = e; // a destination of another shape: the reference's two steps
            return apply_galois(dest, galois_elt, galois_keys, dest);
        }
        check_valid(e, "encrypted");
        if (galois_keys.context() != &context_)
            throw std::invalid_argument("galois_keys is not valid for encryption parameters");
        const Level &lvl = *e.level();
        const size_t N = context_.n();
        uint64_t m = 2 * (uint64_t)N;
        if (!(galois_elt & 1) || galois_elt >= m)
        {
            // has_key() throws invalid_argument for an even element before this check in the
            // reference (galoiskeys.h:48-57); either way the class is invalid_argument
            throw std::invalid_argument("Galois element is not valid");
        }
        if (!galois_keys.has_key(galois_index(galois_elt)))
            throw std::invalid_argument("Galois key not present");
        if (e.size() != 2)
            throw std::invalid_argument("encrypted size must be 2");
        const Scheme scheme = context_.scheme();
        if (scheme == Scheme::bfv && e.is_ntt_form())
            throw std::invalid_argument("BFV encrypted cannot be in NTT form");
        if (scheme == Scheme::ckks && !e.is_ntt_form())
            throw std::invalid_argument("CKKS encrypted must be in NTT form");
        if (scheme == Scheme::bgv && !e.is_ntt_form())
            throw std::invalid_argument("BGV encrypted must be in NTT form");

        PlaneGeom g{ (unsigned)context_.log_n(), lvl.K, (unsigned)e.batch() };
        const int ntt_form = scheme == Scheme::bfv ? 0 : 1;
        const size_t key_index = galois_index(galois_elt);
        static const bool gather_ok = !shl_ab_getenv("SEALHIP_GALOIS_KERNELS"); // development builds: the permutation kernels always (A/B)
        if (gather_ok && scheme == Scheme::ckks && ks_folds(galois_keys, key_index, lvl.K) && ks_split(e, galois_keys, key_index) == 1)
        {
            // Round 6: batches whose key switch runs un-split with the addend folded into its sums - neither pi(c0) nor pi(c1) is ever
            // stored: the key switch's own kernels read c0 and c1 through the automorphism's index map (the opening inverse transform
            // and the diagonal terms: c1; ks2's epilogue: c0), and the result's polynomials are written by the tail alone.
            g_galois_gathered++;
            const size_t words = 2 * g.words();
            const uint64_t *c0 = e.plane(0), *c1 = e.plane(1); // (completes whatever is pending on the operand)
            uint64_t *out = DevicePool::global().alloc_words(words, stream_);
            uint64_t *old = nullptr; // in place: the operand's slab has to outlive the kernels that read it
            if (&dest == &e)
                old = dest.exchange_slab(&lvl, 2, out, words);
            else
            {
                dest.is_ntt_form() = e.is_ntt_form();
                dest.scale() = e.scale();
                dest.correction_factor() = e.correction_factor();
                dest.adopt(&lvl, 2, out, words);
            }
            try
            {
                switch_key_inplace(dest, c1, galois_keys, key_index, true, galois_elt, c0);
                throw_if_transparent(dest);
            }
            catch (...)
            {
                if (&dest != &e)
                    dest.release(); // (as below: no half-built destination with valid-looking metadata)
                else
                {
                    // in place: the operand's words are still in `old` - give them back to the object, as the failure left nothing else
                    dest.drop_lazy();
                    DevicePool::global().free_words(dest.exchange_slab(&lvl, 2, old, words), stream_);
                }
                throw;
            }
            DevicePool::global().free_words(old, stream_);
            return;
        }
        // pi(c0) goes straight into the result slab, pi(c1) into scratch as the key-switch target, c1 starts at zero
        g_galois_permuted++;
        Scratch perm(g.words());
        const size_t words = 2 * g.words();
        uint64_t *out = DevicePool::global().alloc_words(words);
        try
        {
            ck(k_apply_galois(context_.dev_mods(), e.plane(0), out, galois_elt, ntt_form, g, 1, stream_), "apply_galois c0");
            ck(k_apply_galois(context_.dev_mods(), e.plane(1), perm.p, galois_elt, ntt_form, g, 1, stream_), "apply_galois c1");
            // (c1 = 0 is left unwritten: switch_key_inplace zeroes it only on the paths that read it, round 6)
        }
        catch (...)
        {
            DevicePool::global().free_words(out);
            throw;
        }
        if (&dest != &e)
        {
            dest.is_ntt_form() = e.is_ntt_form();
            dest.scale() = e.scale();
            dest.correction_factor() = e.correction_factor();
        }
        dest.adopt(&lvl, 2, out, words);
        try
        {
            switch_key_inplace(dest, perm.p, galois_keys, key_index, true);
            throw_if_transparent(dest);
        }
        catch (...)
        {
            // a separate destination must not keep the half-built (pi(c0), 0) with valid-looking metadata: it is left
            // EMPTY - the reference leaves a copy of the operand there (evaluator.h:1130-1135: destination = encrypted first),
            // which this form never makes; in place the object is the caller's operand and stays what the failure left of it,
            // as in the reference
            if (&dest != &e)
                dest.release();
            throw;
        }
    }

    // out-of-place forms (evaluator.h:1130-1315: destination = encrypted; *_inplace(destination)): with the exact key present the
    // operand is read where it lies; the NAF fallback copies first like the reference
    void Evaluator::rotate_internal(const Ciphertext &e, int steps, const KSwitchKeys &galois_keys, Ciphertext &dest) const
    {
        StreamScope pool_scope(stream_); // pool traffic of this call is ordered on the evaluator's stream whoever calls (pool.h)
        if (&dest == &e)
            return rotate_internal(dest, steps, galois_keys);
        if (&e.context() != &context_ || !e.level())
            throw std::invalid_argument("encrypted is not valid for encryption parameters");
        if (!context_.using_batching())
            throw std::logic_error("encryption parameters do not support batching");
        if (galois_keys.context() != &context_)
            throw std::invalid_argument("galois_keys is not valid for encryption parameters");
        const uint32_t elt = steps ? galois_elt_from_step(steps) : 0;
        if (steps && galois_keys.has_key(galois_index(elt)))
            return apply_galois(e, elt, galois_keys, dest);
        dest = e;
        rotate_internal(dest, steps, galois_keys);
    }
    void Evaluator::rotate_rows(const Ciphertext &e, int steps, const KSwitchKeys &gk, Ciphertext &dest) const
    {
        if (context_.scheme() != Scheme::bfv && context_.scheme() != Scheme::bgv)
            throw std::logic_error("unsupported scheme");
        rotate_internal(e, steps, gk, dest);
    }
    void Evaluator::rotate_vector(const Ciphertext &e, int steps, const KSwitchKeys &gk, Ciphertext &dest) const
    {
        if (context_.scheme() != Scheme::ckks)
            throw std::logic_error("unsupported scheme");
        rotate_internal(e, steps, gk, dest);
    }
    void Evaluator::rotate_columns(const Ciphertext &e, const KSwitchKeys &gk, Ciphertext &dest) const
    {
        if (context_.scheme() != Scheme::bfv && context_.scheme() != Scheme::bgv)
            throw std::logic_error("unsupported scheme");
        if (&e.context() != &context_ || !e.level())
            throw std::invalid_argument("encrypted is not valid for encryption parameters");
        if (!context_.using_batching())
            throw std::logic_error("encryption parameters do not support batching");
        apply_galois(e, galois_elt_from_step(0), gk, dest);
    }
    void Evaluator::complex_conjugate(const Ciphertext &e, const KSwitchKeys &gk, Ciphertext &dest) const
    {
        if (context_.scheme() != Scheme::ckks)
            throw std::logic_error("unsupported scheme");
        if (&e.context() != &context_ || !e.level())
            throw std::invalid_argument("encrypted is not valid for encryption parameters");
        if (!context_.using_batching())
            throw std::logic_error("encryption parameters do not support batching");
        apply_galois(e, galois_elt_from_step(0), gk, dest);
    }
    void Evaluator::rotate_internal(Ciphertext &e, int steps, const KSwitchKeys &galois_keys) const
    {
        StreamScope pool_scope(stream_); // pool traffic of this call is ordered on the evaluator's stream whoever calls (pool.h)
        if (&e.context() != &context_ || !e.level())
            throw std::invalid_argument("encrypted is not valid for encryption parameters");
        if (!context_.using_batching())
            throw std::logic_error("encryption parameters do not support batching");
        if (galois_keys.context() != &context_)
            throw std::invalid_argument("galois_keys is not valid for encryption parameters");
        if (steps == 0)
            return;
        size_t coeff_count = context_.n();
        uint32_t elt = galois_elt_from_step(steps);
        if (galois_keys.has_key(galois_index(elt)))
        {
            apply_galois_inplace(e, elt, galois_keys);
        }
        else
        {
            std::vector<int> naf_steps = naf(steps);
            if (naf_steps.size() == 1)
                throw std::invalid_argument("Galois key not present");
            for (int step : naf_steps)
                if ((size_t)std::abs(step) != (coeff_count >> 1))
                    rotate_internal(e, step, galois_keys);
        }
    }
    void Evaluator::conjugate_internal(Ciphertext &e, const KSwitchKeys &galois_keys) const
    {
        StreamScope pool_scope(stream_); // pool traffic of this call is ordered on the evaluator's stream whoever calls (pool.h)
        if (&e.context() != &context_ || !e.level())
            throw std::invalid_argument("encrypted is not valid for encryption parameters");
        if (!context_.using_batching())
            throw std::logic_error("encryption parameters do not support batching");
        apply_galois_inplace(e, galois_elt_from_step(0), galois_keys);
    }
    void Evaluator::rotate_rows_inplace(Ciphertext &e, int steps, const KSwitchKeys &gk) const
    {
        StreamScope pool_scope(stream_); // pool traffic of this call is ordered on the evaluator's stream whoever calls (pool.h)
        if (context_.scheme() != Scheme::bfv && context_.scheme() != Scheme::bgv)
            throw std::logic_error("unsupported scheme");
        rotate_internal(e, steps, gk);
    }
    void Evaluator::rotate_columns_inplace(Ciphertext &e, const KSwitchKeys &gk) const
    {
        StreamScope pool_scope(stream_); // pool traffic of this call is ordered on the evaluator's stream whoever calls (pool.h)
        if (context_.scheme() != Scheme::bfv && context_.scheme() != Scheme::bgv)
            throw std::logic_error("unsupported scheme");
        conjugate_internal(e, gk);
    }
    void Evaluator::rotate_vector_inplace(Ciphertext &e, int steps, const KSwitchKeys &gk) const
    {
        StreamScope pool_scope(stream_); // pool traffic of this call is ordered on the evaluator's stream whoever calls (pool.h)
        if (context_.scheme() != Scheme::ckks)
            throw std::logic_error("unsupported scheme");
        rotate_internal(e, steps, gk);
    }
    void Evaluator::complex_conjugate_inplace(Ciphertext &e, const KSwitchKeys &gk) const
    {
        StreamScope pool_scope(stream_); // pool traffic of this call is ordered on the evaluator's stream whoever calls (pool.h)
        if (context_.scheme() != Scheme::ckks)
            throw std::logic_error("unsupported scheme");
        conjugate_internal(e, gk);
    }
} // namespace sealhip
