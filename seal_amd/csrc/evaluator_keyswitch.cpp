// Evaluator, part 2: relinearize, switch_key_inplace in its two halves, digit-parallel key switching
#include "evaluator_common.h"
#include <map>
#include <memory>

namespace sealhip
{
    namespace
    {
        // ---- Chunked key switching on forked lanes (round 5; VERDICT r4 "missing" 4).
        // The fused key switch is two kernels per arithmetic class: ks1t writes the K(K+1) half-transformed digits of every
        // ciphertext (126 MB per C5 ciphertext), ks2 reads them back and multiplies them into the key.  Run over the whole batch
        // the intermediate of the WHOLE batch is resident (32 GB at batch 256, 258 GB at batch 2048) where the reference's
        // switch_key_inplace holds O(K N) per ciphertext (evaluator.cpp:2561-2867).  So the batch is cut into chunks that go
        // round-robin to `lanes` streams forked from / joined to the evaluator's stream, each lane with its own chunk-sized
        // intermediate (and the inverse transform of its chunk's target): the scratch is lanes x chunk items whatever the batch.
        // What the lanes buy is the launch tails of the chunks (one lane: -2.1 %; three: +0.7 % +- 0.7 against one launch over the
        // batch) - NOT an overlap of the two passes: both need the vector ALU, side by side each runs slower by what the other
        // takes, staggered or in lockstep (profiles/r05_ks_chunked.txt).
        struct KsPlan
        {
            unsigned chunk; // items per chunk
            unsigned lanes; // streams the chunks are dealt to (1 = everything on the evaluator's stream)
            unsigned chunks(unsigned batch) const { return (batch + chunk - 1) / chunk; }
        };
        unsigned env_unsigned(const char *name, unsigned fallback)
        {
            const char *v = std::getenv(name);
            return v && *v ? (unsigned)std::strtoul(v, nullptr, 10) : fallback;
        }
        // words_per_item = (K + 1) K N; wgs_per_item = pass-2 workgroups of one ciphertext ((K + 1) tiles)
        KsPlan ks_plan(unsigned batch, size_t words_per_item, size_t wgs_per_item, bool may_chunk, bool may_lane)
        {
            KsPlan p{ batch, 1 };
            if (!may_chunk || batch < 2)
                return p;
            // a chunk must still fill the chip several times over (pass 2 keeps 512 workgroups resident; the kernels reach
            // their rate from 18 - 64 C5 ciphertexts on, profiles/r03_ks_batch_sweep.txt): >= 8192 pass-2 workgroups
            unsigned chunk = (unsigned)((8192 + wgs_per_item - 1) / (wgs_per_item ? wgs_per_item : 1));
            chunk = env_unsigned("SEALHIP_KS_CHUNK", chunk < 8 ? 8 : chunk);
            unsigned lanes = env_unsigned("SEALHIP_KS_LANES", 3);
            if (lanes < 1 || !may_lane)
                lanes = 1;
            if (lanes > 4)
                lanes = 4;
            if (chunk == 0 || 2 * chunk > batch + chunk / 2) // fewer than ~1.5 chunks: not worth cutting
                return p;
            // scratch cap (default 16 GiB): lanes x chunk items of the intermediate
            const double cap = (double)env_unsigned("SEALHIP_KS_SCRATCH_CAP_MIB", 16384) * 1048576.0;
            while (chunk > 1 && (double)lanes * chunk * words_per_item * 8.0 > cap)
            {
                if (lanes > 2)
                    lanes--;
                else
                    chunk = (chunk + 1) / 2;
            }
            p.chunk = chunk;
            p.lanes = lanes;
            if (p.chunks(batch) < p.lanes)
                p.lanes = p.chunks(batch);
            return p;
        }
        // the side streams of the lanes (lane 0 is the evaluator's own stream) and the events that fork / join them
        struct KsLanes
        {
            static constexpr unsigned kSide = 3;
            hipStream_t stream[kSide] = { nullptr, nullptr, nullptr };
            hipEvent_t join[kSide] = { nullptr, nullptr, nullptr };
            hipEvent_t fork = nullptr;
            unsigned made = 0;
            int device = -1; // streams belong to the device that was current when they were made
            KsLanes() = default;
            KsLanes(const KsLanes &) = delete;
            KsLanes &operator=(const KsLanes &) = delete;
            ~KsLanes()
            {
                //  a lane set owns its streams and events: short-lived worker threads must not leak one set each.  Best effort - at
                // thread or process exit the runtime may already be gone, and errors are of no use here.
                for (unsigned i = 0; i < kSide; i++)
                {
                    if (stream[i])
                        (void)hipStreamDestroy(stream[i]);
                    if (join[i])
                        (void)hipEventDestroy(join[i]);
                }
                if (fork)
                    (void)hipEventDestroy(fork);
            }
            bool ensure(unsigned side)
            {
                if (!fork && hipEventCreateWithFlags(&fork, hipEventDisableTiming) != hipSuccess)
                    return false;
                while (made < side && made < kSide)
                {
                    // stream and event of a lane are made together: a half-made lane is taken down again, not leaked
                    hipStream_t st = nullptr;
                    hipEvent_t ev = nullptr;
                    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess)
                        return false;
                    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess)
                    {
                        (void)hipStreamDestroy(st);
                        return false;
                    }
                    stream[made] = st;
                    join[made] = ev;
                    made++;
                }
                return made >= side;
            }
        };
        // one lane set per (host thread, device): a thread that alternates between GPUs keeps a set on each instead of dropping and
        // re-making them at every switch; one process per GPU - the model - has exactly one
        KsLanes *ks_lanes()
        {
            static thread_local std::map<int, std::unique_ptr<KsLanes>> sets;
            int dev = -1;
            if (hipGetDevice(&dev) != hipSuccess)
                return nullptr;
            std::unique_ptr<KsLanes> &slot = sets[dev];
            if (!slot)
            {
                slot.reset(new KsLanes());
                slot->device = dev;
            }
            return slot.get();
        }
        std::atomic<uint64_t> g_ks_chunked_calls{ 0 }, g_ks_chunks{ 0 }, g_ks_scratch_words_max{ 0 };
    } // namespace
    // (SealHip_KsChunkStats: how often the key switch ran in chunks, how many chunks, the largest intermediate it held since the last query)
    void ks_chunk_stats(uint64_t *calls, uint64_t *chunks, uint64_t *scratch_words_max)
    {
        if (calls)
            *calls = g_ks_chunked_calls.load();
        if (chunks)
            *chunks = g_ks_chunks.load();
        if (scratch_words_max)
            *scratch_words_max = g_ks_scratch_words_max.exchange(0); // ... since the previous query
    }

    // ---- relinearize (evaluator.cpp:1144-1199)
    void Evaluator::relinearize_inplace(Ciphertext &e, const KSwitchKeys &relin_keys) const
    {
        StreamScope pool_scope(stream_); // pool traffic of this call is ordered on the evaluator's stream whoever calls (pool.h)
        if (&e.context() != &context_ || !e.level())
            throw std::invalid_argument("encrypted is not valid for encryption parameters");
        if (relin_keys.context() != &context_)
            throw std::invalid_argument("relin_keys is not valid for encryption parameters");
        size_t size = e.size();
        const size_t destination_size = 2;
        if (destination_size > size)
            throw std::invalid_argument("destination_size must be at least 2 and less than or equal to current count");
        if (relin_keys.size() < size - 2)
            throw std::invalid_argument("not enough relinearization keys");
        if (destination_size == size)
            return;
        if (size == 3 && e.has_lazy_product() && relinearize_from_product(e, relin_keys))
            return;
        size_t relins_needed = size - destination_size;
        // the reference passes the LAST polynomial as the target of every step (evaluator.cpp:1180-1188)
        for (size_t I = 0; I < relins_needed; I++)
            switch_key_inplace(e, e.plane(size - 1), relin_keys, relin_index(size - 1 - I));
        e.resize(e.level(), destination_size, stream_);
        throw_if_transparent(e);
    }

    // relinearize_inplace of a product that was never stored (LazyProduct): the same key switch with the product formed inside it.
    // Only the configuration the large-batch headline runs: the fused path with the full digit range in one group, the addend folded
    // into the sums, the tail deferred.  Anything else returns false and the caller takes the ordinary path (which forms the product).
    void lazy_product_count_fused();
    bool Evaluator::relinearize_from_product(Ciphertext &e, const KSwitchKeys &relin_keys) const
    {
        static const bool lazy_ok = !std::getenv("SEALHIP_KS_EAGER_TAIL"), fold_ok = !shl_ab_getenv("SEALHIP_KS_NO_FOLD");
        const size_t key_index = relin_index(2);
        const LazyProduct *pending = e.lazy_prod_;
        if (!pending || pending->owner != this || capturing_ || !lazy_ok || !fold_ok || context_.scheme() != Scheme::ckks || !e.is_ntt_form() ||
            !ntt2_supports(context_.log_n()) || !context_.using_keyswitching())
            return false;
        const unsigned K = e.level()->K;
        if (K < 2 || key_index >= relin_keys.slots() || !relin_keys.has_key(key_index) || !relin_keys.key(key_index).register_order)
            return false;
        {
            // the rule of switch_key_inplace: the digit loop cut into groups (small batches) has a reduce pass that wants the product's words
            const size_t wgs = e.batch() * (size_t)(K + 1) * (context_.n() >> 12);
            unsigned split = (unsigned)(2048 / (wgs ? wgs : 1));
            if (const char *f = std::getenv("SEALHIP_KS_SPLIT"))
                split = (unsigned)std::atoi(f);
            if (split > 1)
                return false;
        }
        const KSwitchKeys::Key &key = relin_keys.key(key_index);
        if (key.digit0 != 0 || key.digits < K)
            return false;
        const LazyProduct p = detach_product(e); // from here on e is an ordinary size-3 ciphertext whose words are not there yet
        lazy_product_count_fused();
        uint64_t *target = e.data_ + 2 * e.plane_words(); // (names the polynomial; neither read nor written: x1 y1 is formed from the operands where it is used)
        StreamScope pool_scope(stream_);
        try
        {
            Scratch acc(switch_key_acc_words(e));
            switch_key_partial(e, target, relin_keys, key_index, 0, K, acc.p, 1, true, &p);
            defer_tail(e, acc.release(), true);
        }
        catch (...)
        {
            // nothing of e has been written except, possibly, its third polynomial: form the product the ordinary way so that e is what
            // multiply() promised, then let the error travel
            try
            {
                PlaneGeom g{ (unsigned)context_.log_n(), K, (unsigned)e.batch() };
                (void)k_ckks_multiply_2x2(context_.dev_mods(), context_.ntt_tables().fpd, nullptr, p.xw(), p.yw(), e.data_, g, stream_);
            }
            catch (...)
            {
            }
            DevicePool::global().free_words(p.own, stream_);
            throw;
        }
        // an in-place product's previous slab: every kernel that reads it has been queued on (or joined to) this stream
        DevicePool::global().free_words(p.own, stream_);
        e.resize(e.level(), 2, stream_);
        throw_if_transparent(e);
        return true;
    }

    void Evaluator::relinearize_partial(Ciphertext &e, const KSwitchKeys &relin_keys, unsigned j0, unsigned j1, uint64_t *acc) const
    {
        StreamScope pool_scope(stream_); // pool traffic of this call is ordered on the evaluator's stream whoever calls (pool.h)
        if (&e.context() != &context_ || !e.level())
            throw std::invalid_argument("encrypted is not valid for encryption parameters");
        if (relin_keys.context() != &context_ && !(j0 == j1 && !relin_keys.context())) // a rank without digits may hold no key
            throw std::invalid_argument("relin_keys is not valid for encryption parameters");
        if (e.size() != 3)
            throw std::invalid_argument("digit-parallel relinearization takes a size-3 ciphertext");
        switch_key_partial(e, e.plane(2), relin_keys, relin_index(2), j0, j1, acc);
    }
    void Evaluator::relinearize_finish(Ciphertext &e, uint64_t *acc, unsigned parts) const
    {
        StreamScope pool_scope(stream_); // pool traffic of this call is ordered on the evaluator's stream whoever calls (pool.h)
        if (&e.context() != &context_ || !e.level() || e.size() != 3)
            throw std::invalid_argument("encrypted is not valid for encryption parameters");
        switch_key_finish(e, acc, parts, false, true);
        e.resize(e.level(), 2, stream_);
        throw_if_transparent(e);
    }
    void Evaluator::apply_galois_partial(
        Ciphertext &e, uint32_t galois_elt, const KSwitchKeys &galois_keys, unsigned j0, unsigned j1, uint64_t *acc) const
    {
        StreamScope pool_scope(stream_); // pool traffic of this call is ordered on the evaluator's stream whoever calls (pool.h)
        check_valid(e, "encrypted");
        if (galois_keys.context() != &context_ && !(j0 == j1 && !galois_keys.context()))
            throw std::invalid_argument("galois_keys is not valid for encryption parameters");
        uint64_t m = 2 * (uint64_t)context_.n();
        if (!(galois_elt & 1) || galois_elt >= m)
            throw std::invalid_argument("Galois element is not valid");
        if (j0 < j1 && !galois_keys.has_key(galois_index(galois_elt))) // a rank without digits holds no slice of the key
            throw std::invalid_argument("Galois key not present");
        if (e.size() != 2)
            throw std::invalid_argument("encrypted size must be 2");
        const Scheme scheme = context_.scheme();
        PlaneGeom g{ (unsigned)context_.log_n(), e.level()->K, (unsigned)e.batch() };
        const int ntt_form = scheme == Scheme::bfv ? 0 : 1;
        if ((ntt_form != 0) != e.is_ntt_form())
            throw std::invalid_argument(ntt_form ? "encrypted must be in NTT form" : "BFV encrypted cannot be in NTT form");
        Scratch perm(2 * g.words()); // [pi(c0), pi(c1)]
        ck(k_apply_galois(context_.dev_mods(), e.data(), perm.p, galois_elt, ntt_form, g, 2, stream_), "apply_galois");
        ck(hipMemcpyAsync(e.plane(0), perm.p, g.words() * 8, hipMemcpyDeviceToDevice, stream_), "galois copy c0");
        ck(hipMemsetAsync(e.plane(1), 0, g.words() * 8, stream_), "galois zero c1");
        switch_key_partial(e, perm.p + g.words(), galois_keys, galois_index(galois_elt), j0, j1, acc);
    }
    void Evaluator::apply_galois_finish(Ciphertext &e, uint64_t *acc, unsigned parts) const
    {
        StreamScope pool_scope(stream_); // pool traffic of this call is ordered on the evaluator's stream whoever calls (pool.h)
        if (&e.context() != &context_ || !e.level() || e.size() != 2)
            throw std::invalid_argument("encrypted is not valid for encryption parameters");
        switch_key_finish(e, acc, parts, false, true);
        throw_if_transparent(e);
    }

    // ---- switch_key_inplace (evaluator.cpp:2561-2867), in two halves so that the decomposition digits can be
    // spread over the GPUs of a node: partial = the I/J loop restricted to the digits [j0, j1)
    // (canonical partial sums S_k[I]), finish = mod-down by the special prime and accumulation into (c0, c1).
    // Between the halves the caller may add the partial sums of several ranks (one all-reduce of 2(K+1)N words).
    size_t Evaluator::switch_key_acc_words(const Ciphertext &e) const
    {
        if (&e.context() != &context_ || !e.level())
            throw std::invalid_argument("encrypted is not valid for encryption parameters");
        return (size_t)e.batch() * 2 * (e.level()->K + 1) * context_.n();
    }

    void Evaluator::switch_key_partial(
        const Ciphertext &e, const uint64_t *target, const KSwitchKeys &keys, size_t key_index, unsigned j0, unsigned j1,
        uint64_t *acc_out, unsigned split, bool fold_addend, const LazyProduct *product, bool addend1_zero, uint32_t galois_elt,
        const uint64_t *galois_c0) const
    {
        StreamScope pool_scope(stream_); // pool traffic of this call is ordered on the evaluator's stream whoever calls (pool.h)
        check_valid(e, "encrypted");
        if (!target)
            throw std::invalid_argument("target_iter");
        if (!acc_out)
            throw std::invalid_argument("acc");
        if (!context_.using_keyswitching())
            throw std::logic_error("keyswitching is not supported by the context");
        if (keys.context() != &context_ && !(j0 == j1 && !keys.context()))
            throw std::invalid_argument("parameter mismatch");
        if (j0 == j1 && j0 <= e.level()->K)
        {
            // a rank without digits (more ranks than digits) needs no key: its partial sums are zero
            ck(hipMemsetAsync(acc_out, 0, switch_key_acc_words(e) * 8 * (split ? split : 1), stream_), "ks zero partial sums");
            return;
        }
        if (key_index >= keys.slots())
            throw std::out_of_range("kswitch_keys_index");
        const Scheme scheme = context_.scheme();
        if (scheme == Scheme::bfv && e.is_ntt_form())
            throw std::invalid_argument("BFV encrypted cannot be in NTT form");
        if (scheme == Scheme::ckks && !e.is_ntt_form())
            throw std::invalid_argument("CKKS encrypted must be in NTT form");
        if (scheme == Scheme::bgv && !e.is_ntt_form())
            throw std::invalid_argument("BGV encrypted must be in NTT form");
        if (!keys.has_key(key_index))
            throw std::invalid_argument("kswitch_keys is not valid for encryption parameters");
        const KSwitchKeys::Key &key = keys.key(key_index);
        const Level &lvl = *e.level();
        const Level &klvl = context_.key_level();
        const unsigned K = lvl.K, L = klvl.K;
        if (j1 > K || j0 > j1)
            throw std::invalid_argument("digit range");
        if (fold_addend && !(scheme == Scheme::ckks && key.register_order && j0 == 0 && j1 == K && split <= 1))
            throw std::invalid_argument("fold_addend"); // the addend may only join the complete sum, on the fused path
        if (j0 < j1 && (key.digit0 > j0 || key.digit0 + key.digits < j1))
            throw std::invalid_argument("kswitch_keys inner dimension is too small");
        if (e.size() < 2)
            throw std::invalid_argument("encrypted size must be at least 2");
        if (product && !(fold_addend && ntt2_supports(context_.log_n()) && (!product->x || (product->x->level() == e.level() && product->x->batch() == e.batch())) &&
                         (!product->y || (product->y->level() == e.level() && product->y->batch() == e.batch())) &&
                         (product->own || (product->x && product->y))))
            throw std::invalid_argument("product");
        if (galois_elt && !(fold_addend && addend1_zero && !product && galois_c0 && scheme == Scheme::ckks && ntt2_supports(context_.log_n())))
            throw std::invalid_argument("galois_elt"); // the gather lives in the two-pass inverse transform and in ks2's folded epilogue
        // (the operands' words: read once - a live operand's data() completes whatever is pending on it)
        const uint64_t *prod_xw = product ? product->xw() : nullptr, *prod_yw = product ? product->yw() : nullptr;

        const size_t N = context_.n();
        const unsigned B = (unsigned)e.batch();
        const unsigned n_log = (unsigned)context_.log_n();
        const NttTables &tb = context_.ntt_tables();
        const ModDesc *mods = context_.dev_mods();
        const uint32_t *map = ks_comp_prime(K);
        // t_target: coefficient form of every decomposition digit (evaluator.cpp:2651-2658)
        const bool ntt_target = scheme == Scheme::ckks || scheme == Scheme::bgv; // the target is in NTT form
        // BFV on the fused path: the target is in coefficient form already and the kernels only read it - no copy
        const bool read_in_place = !ntt_target && key.register_order;
        Scratch t(read_in_place ? 1 : (size_t)B * K * N);
        const uint64_t *digits = read_in_place ? target : t.p;
        // fused path, large batches: chunks dealt to lanes (above); the inverse transform of the target then runs per chunk
        // inside its lane
        const size_t ks_item_words = (size_t)(K + 1) * K * N;
        const KsPlan plan = key.register_order
                                ? ks_plan(B, ks_item_words, (size_t)(K + 1) * (N >> 12), split <= 1 && ntt2_supports(context_.log_n()), !capturing_)
                                : KsPlan{ B, 1 };
        const bool chunked = plan.chunk < B;
        const bool inverse_in_lane = chunked && !read_in_place && ntt_target;
        if (read_in_place || inverse_in_lane)
            ;
        else if (ntt_target && ntt2_supports(context_.log_n()))
        {
            // out-of-place: the two-pass engine reads the target and writes t
            NttBatch bt = plain_batch(t.p, (size_t)K * N, K, B, 0);
            bt.src = target;
            bt.src_outer_stride = (size_t)K * N;
            bt.src_galois_elt = galois_elt;
            if (product)
            {
                // the target x1 y1 is formed while it is loaded; it is not stored (ks2 forms its diagonal terms from the operands too)
                bt.prod_x = prod_xw;
                bt.prod_y = prod_yw;
                bt.prod_batch = B;
                bt.prod_outer0 = 2 * B;
            }
            ck(ntt_inverse(tb, bt, 0, stream_), "ks intt target");
        }
        else
        {
            if (product)
                throw std::invalid_argument("product");
            ck(hipMemcpyAsync(t.p, target, (size_t)B * K * N * 8, hipMemcpyDeviceToDevice, stream_), "ks copy target");
            if (ntt_target)
                ck(ntt_inverse(tb, plain_batch(t.p, (size_t)K * N, K, B, 0), 0, stream_), "ks intt target");
        }

        if (key.register_order)
        {
            // fused path (ntt2_kernels.hip): the K(K+1) raised digits go through HBM once, between
            // the two passes, and are multiplied into the key inside the second pass
            const KsTargets &kt = ks_targets(K);
            Scratch mid((size_t)plan.lanes * plan.chunk * ks_item_words);
            Scratch inv_mid(inverse_in_lane ? (size_t)plan.lanes * plan.chunk * K * N : 1);
            {
                uint64_t w = (uint64_t)plan.lanes * plan.chunk * ks_item_words, seen = g_ks_scratch_words_max.load();
                while (w > seen && !g_ks_scratch_words_max.compare_exchange_weak(seen, w))
                    ;
            }
            KsFusedArgs ka{};
            ka.t = digits;
            // the I == J shortcut of evaluator.cpp:2682-2685 (a deferred product: x1 y1 is formed where it is needed, ks2 takes it from fold_x / fold_y)
            ka.target_ntt = ntt_target && !product ? target : nullptr;
            ka.key = key.dev;
            ka.mid = mid.p;
            ka.acc = acc_out;
            ka.targets1 = kt.dev;
            ka.targets2 = kt.dev + 2 * (kt.n_int + kt.n_fp);
            ka.ntargets = kt.n_int + kt.n_fp;
            ka.n_int = kt.n_int;
            ka.K = K;
            ka.L = L;
            ka.batch = B;
            ka.j0 = j0;
            ka.j1 = j1;
            ka.key_digit0 = (unsigned)key.digit0;
            ka.parts = split ? split : 1;
            if (fold_addend && product)
            {
                ka.fold_x = prod_xw;
                ka.fold_y = prod_yw;
                ka.fold_plane = e.plane_words();
                ka.fold_pm = klvl.dev.inv_q_last_mod_q;
            }
            else if (fold_addend)
            {
                ka.fold_c0 = galois_elt ? galois_c0 : e.plane(0);
                ka.fold_c1 = addend1_zero ? nullptr : e.plane(1);
                ka.fold_pm = klvl.dev.inv_q_last_mod_q;
                ka.galois_elt = galois_elt;
            }
            if (!chunked)
                ck(ks_fused(tb, ka, stream_), "ks fused");
            else
            {
                // which pass-1 kernel: decided for the whole batch (ntt2_kernels.hip: launch_ks decides from the grid it is given)
                ka.order1 = (size_t)B * (j1 - j0) * (N >> 12) >= 4096 ? 1 : 0;
                KsLanes *lnp = ks_lanes();
                unsigned lanes = plan.lanes;
                if (lanes > 1 && (!lnp || !lnp->ensure(lanes - 1)))
                    lanes = 1;
                KsLanes dummy_lanes_for_one_lane; // (never touched when lanes == 1)
                KsLanes &ln = lnp ? *lnp : dummy_lanes_for_one_lane;
                // (the two arithmetic classes of a chunk one after the other on its lane; SEALHIP_KS_CLASS_FORK=1 in development builds
                // forks the integer class to the launcher's side stream as the unchunked path does: profiles/r05_ks_chunked.txt)
                static const bool class_fork = shl_ab_getenv("SEALHIP_KS_CLASS_FORK") != nullptr;
                ka.no_class_fork = lanes > 1 && !class_fork;
                static const bool trace = shl_ab_getenv("SEALHIP_KS_TRACE") != nullptr;
                if (trace)
                    std::fprintf(stderr, "[ks] batch %u in chunks of %u on %u lane(s)\n", B, plan.chunk, lanes);
                if (lanes > 1)
                {
                    ck(hipEventRecord(ln.fork, stream_), "ks lanes fork");
                    for (unsigned l = 1; l < lanes; l++)
                        ck(hipStreamWaitEvent(ln.stream[l - 1], ln.fork, 0), "ks lanes fork");
                }
                const size_t poly_words = (size_t)K * N; // one polynomial of one item in the [batch][K][N] planes
                unsigned c = 0;
                //  a launch that fails half-way must not leave forked lanes running into scratch blocks that go back to
                // the pool ordered on stream_ only: join whatever was forked before the exception travels on
                auto join_lanes = [&](bool nothrow) {
                    for (unsigned l = 1; l < lanes; l++)
                    {
                        const hipError_t e1 = hipEventRecord(ln.join[l - 1], ln.stream[l - 1]);
                        const hipError_t e2 = e1 == hipSuccess ? hipStreamWaitEvent(stream_, ln.join[l - 1], 0) : e1;
                        if (e2 != hipSuccess)
                        {
                            if (!nothrow)
                                ck(e2, "ks lanes join");
                            (void)hipStreamSynchronize(ln.stream[l - 1]); // the event path failed too: wait for the lane outright
                        }
                    }
                };
                try
                {
                // (The lanes run in lockstep - forked together, equal work.  A two-stage form that staggers them by construction - one
                // producer stream for inverse transform + pass 1, the evaluator's stream consuming a ring of intermediates with pass 2
                // - was measured too: 9.24 k ct/s against 9.27 - 9.40 k, profiles/r05_ks_chunked.txt.  Both passes need the vector
                // ALU; side by side each runs slower by what the other takes.)
                for (unsigned b0 = 0; b0 < B; b0 += plan.chunk, c++)
                {
                    const unsigned nb = B - b0 < plan.chunk ? B - b0 : plan.chunk;
                    const unsigned l = c % lanes;
                    hipStream_t st = l == 0 ? stream_ : ln.stream[l - 1];
                    if (inverse_in_lane)
                    {
                        NttBatch bt = plain_batch(t.p + b0 * poly_words, poly_words, K, nb, 0);
                        bt.src = target + b0 * poly_words;
                        bt.src_outer_stride = poly_words;
                        bt.src_galois_elt = galois_elt;
                        if (product)
                        {
                            bt.prod_x = prod_xw;
                            bt.prod_y = prod_yw;
                            bt.prod_batch = B;
                            bt.prod_outer0 = 2 * B + b0;
                        }
                        ck(ntt2_inverse(tb, bt, 0, inv_mid.p + (size_t)l * plan.chunk * poly_words, st), "ks intt target (chunk)");
                    }
                    KsFusedArgs kc = ka;
                    kc.batch = nb;
                    kc.t = digits + b0 * poly_words;
                    kc.target_ntt = ka.target_ntt ? ka.target_ntt + b0 * poly_words : nullptr;
                    kc.mid = mid.p + (size_t)l * plan.chunk * ks_item_words;
                    kc.acc = acc_out + (size_t)b0 * 2 * (K + 1) * N;
                    if (fold_addend && product)
                    {
                        kc.fold_x = ka.fold_x + b0 * poly_words;
                        kc.fold_y = ka.fold_y + b0 * poly_words;
                    }
                    else if (fold_addend)
                    {
                        kc.fold_c0 = ka.fold_c0 + b0 * poly_words;
                        kc.fold_c1 = ka.fold_c1 ? ka.fold_c1 + b0 * poly_words : nullptr;
                    }
                    ck(ks_fused(tb, kc, st), "ks fused (chunk)");
                }
                }
                catch (...)
                {
                    join_lanes(true);
                    throw;
                }
                join_lanes(false);
                g_ks_chunked_calls++;
                g_ks_chunks += c;
            }
        }
        else if (split > 1 || product)
            throw std::invalid_argument("in-launch digit groups and deferred products need the fused key-switch path");
        else
        {
            // u[b][I][J] = NTT_I(t_J mod q_I), I over the K data primes and the special prime
            // (evaluator.cpp:2663-2701).  The reference skips the transform when I == J in CKKS because
            // NTT_J(INTT_J(x)) = x; computing it gives the same canonical words.
            Scratch u((size_t)B * (K + 1) * K * N);
            NttBatch b{};
            b.data = u.p;
            b.outer_stride = (size_t)(K + 1) * K * N;
            b.ncomp = (K + 1) * K;
            b.nouter = B;
            b.comp_prime = map;
            b.prime_first = 0;
            b.src = digits;
            b.src_outer_stride = (size_t)K * N;
            b.src_ncomp = K;
            b.src_mode = 1;
            ck(ntt_forward(tb, b, 0, stream_), "ks ntt digits");
            // inner product with the key (evaluator.cpp:2703-2755)
            ck(k_keyswitch_mac(mods, u.p, key.dev, acc_out, n_log, K, L, B, j0, j1, (unsigned)key.digit0, stream_), "ks mac");
        }
    }

    void Evaluator::switch_key_finish(Ciphertext &e, uint64_t *acc_p, unsigned parts, bool acc_has_addend, bool may_defer) const
    {
        StreamScope pool_scope(stream_); // pool traffic of this call is ordered on the evaluator's stream whoever calls (pool.h)
        check_valid(e, "encrypted");
        if (!acc_p)
            throw std::invalid_argument("acc");
        if (!context_.using_keyswitching())
            throw std::logic_error("keyswitching is not supported by the context");
        if (e.size() < 2)
            throw std::invalid_argument("encrypted size must be at least 2");
        if (parts < 1 || parts > 8)
            throw std::invalid_argument("parts"); // 8 canonical residues below 2^60 still fit a 64-bit word
        if (acc_has_addend && !(parts == 1 && context_.scheme() == Scheme::ckks && ntt2_supports(context_.log_n())))
            throw std::invalid_argument("acc_has_addend");
        const Scheme scheme = context_.scheme();
        const Level &lvl = *e.level();
        const Level &klvl = context_.key_level();
        const unsigned K = lvl.K, L = klvl.K;
        const size_t N = context_.n();
        const unsigned B = (unsigned)e.batch();
        const unsigned n_log = (unsigned)context_.log_n();
        const NttTables &tb = context_.ntt_tables();
        const ModDesc *mods = context_.dev_mods();
        const uint32_t *map = ks_comp_prime(K);
        struct AccRef
        {
            uint64_t *p;
        } acc{ acc_p };
        // Digit-parallel callers (the sums of `parts` ranks in the caller's buffer): CKKS at the two-pass sizes takes the road of
        // switch_key_inplace - the pass that reduces the sums adds the ciphertext's words and writes a block of the pool's, the
        // mod-down stays pending (LazyTail) and a rescale on this evaluator folds both divisions (configs[4]: rotate + rescale)
        static const bool lazy_ok = !std::getenv("SEALHIP_KS_EAGER_TAIL");
        static const bool fold_ok = !shl_ab_getenv("SEALHIP_KS_NO_FOLD");
        if (may_defer && lazy_ok && fold_ok && !acc_has_addend && scheme == Scheme::ckks && ntt2_supports(context_.log_n()) && K >= 2)
        {
            const size_t words = switch_key_acc_words(e);
            uint64_t *own = DevicePool::global().alloc_words(words, stream_);
            const hipError_t err = k_keyswitch_reduce(mods, acc.p, n_log, K, L, B, stream_, 1, e.plane(0), e.plane(1),
                                                      klvl.dev.inv_q_last_mod_q, own);
            if (err != hipSuccess)
            {
                DevicePool::global().free_words(own, stream_);
                ck(err, "ks reduce partial sums + addend");
            }
            defer_tail(e, own, true);
            return;
        }
        if (parts > 1)
            ck(k_keyswitch_reduce(mods, acc.p, n_log, K, L, B, stream_), "ks reduce partial sums");

        // mod-down by the special prime P and accumulate into (c0, c1) (evaluator.cpp:2806-2864)
        const uint64_t P = context_.coeff_modulus()[L - 1];
        if (scheme == Scheme::bgv)
        {
            // evaluator.cpp:2762-2805: t_last = INTT_P(S_k[P]); delta = (-(t_last mod t) P^-1 mod t) P + t_last (mod q_i);
            // ct_k[i] += (S_k[q_i] - NTT_i(delta)) P^-1
            NttBatch bi = plain_batch(acc.p + (size_t)K * N, (size_t)(K + 1) * N, 1, 2 * B, L - 1);
            ck(ntt_inverse(tb, bi, 0, stream_), "ks intt special");
            Scratch delta((size_t)B * 2 * K * N);
            ck(k_bgv_delta(mods, host::make_mod(context_.plain_modulus()), klvl.dev.inv_q_last_mod_t, klvl.dev.q_last_mod_q,
                           acc.p + (size_t)K * N, (size_t)(K + 1) * N, delta.p, n_log, K, (size_t)2 * B, stream_),
               "ks bgv delta");
            bgv_correct_and_combine(delta, acc.p, (size_t)(K + 1) * N, klvl.dev.inv_q_last_mod_q, K, 2 * B, e.plane(0), e.plane(1),
                                    (size_t)K * N, 2);
        }
        else if (scheme == Scheme::ckks)
        {
            NttBatch bi = plain_batch(acc.p + (size_t)K * N, (size_t)(K + 1) * N, 1, 2 * B, L - 1);
            ck(ntt_inverse(tb, bi, 0, stream_), "ks intt special");
            Scratch tt(ntt2_supports(context_.log_n()) ? 1 : (size_t)B * 2 * K * N);
            NttBatch b{};
            b.data = tt.p;
            b.outer_stride = (size_t)K * N;
            b.ncomp = K;
            b.nouter = 2 * B;
            b.comp_prime = nullptr;
            b.prime_first = 0;
            b.src = acc.p + (size_t)K * N;
            b.src_outer_stride = (size_t)(K + 1) * N;
            b.src_ncomp = 1;
            b.src_mode = 2;
            b.src_half = P >> 1;
            b.src_q = P;
            b.src_fix = klvl.dev.round_fix;
            if (ntt2_supports(context_.log_n()))
            {
                // the tail is the epilogue of the transform: tt is never stored
                b.data = nullptr;
                b.epi = acc_has_addend ? 3 : 2; // 3: acc = c + S P^-1 already (switch_key_partial, fold_addend)
                b.epi_a = acc.p;
                b.epi_a_stride = (size_t)(K + 1) * N;
                b.epi_mul = klvl.dev.inv_q_last_mod_q;
                b.epi_out0 = e.plane(0);
                b.epi_out1 = e.plane(1);
                b.epi_out_stride = (size_t)K * N;
                ck(ntt_forward(tb, b, 1, stream_), "ks ntt correction + tail");
            }
            else
            {
                ck(ntt_forward(tb, b, 1, stream_), "ks ntt correction");
                ck(k_keyswitch_tail_ckks(mods, klvl.dev.inv_q_last_mod_q, e.plane(0), e.plane(1), acc.p, tt.p, n_log, K, B, stream_),
                   "ks tail");
            }
        }
        else
        {
            NttBatch bi = plain_batch(acc.p, (size_t)(K + 1) * N, K + 1, 2 * B, 0);
            bi.comp_prime = map + (size_t)(K + 1) * K;
            bi.cls_hint = ks_class_hint(K);
            ck(ntt_inverse(tb, bi, 0, stream_), "ks intt all");
            ck(k_keyswitch_tail_bfv(
                   mods, klvl.dev.inv_q_last_mod_q, klvl.dev.round_fix, P >> 1, P, e.plane(0), e.plane(1), acc.p, n_log, K, B,
                   stream_),
               "ks tail bfv");
        }
    }

    // Small batches do not fill the chip: one workgroup per (target modulus, tile, batch item) is 16 (K+1) workgroups per
    // ciphertext at N = 2^16, each looping over all K digits, and only 2 x 16 of them for the two 60-bit moduli.  Cut the
    // digit loop into `split` in-launch groups (their partial sums are added by the reduce pass below): single-ciphertext
    // latency of multiply+relinearize+rescale at C5 0.63 -> 0.40 ms (profiles/HISTORY.md section 5, "Small batches / latency").  SEALHIP_KS_SPLIT overrides (tests, A/B).
    unsigned Evaluator::ks_split(const Ciphertext &e, const KSwitchKeys &keys, size_t key_index) const
    {
        const unsigned K = e.level()->K;
        unsigned split = 1;
        if (keys.context() == &context_ && key_index < keys.slots() && keys.has_key(key_index) && keys.key(key_index).register_order)
        {
            const size_t wgs = e.batch() * (size_t)(K + 1) * (context_.n() >> 12);
            split = (unsigned)(2048 / (wgs ? wgs : 1)); // measured at C5: best split 4 / 4 / 2 / 1 at batch 1 / 2 / 4 / >= 8
            // (round 6, batch 1: split 8 / 6 / 5 / 4 / 3 / 2 = 0.338 / 0.334 / 0.330 / 0.328 / 0.337 / 0.360 ms eager, 0.315 / 0.312 / 0.300 /
            // 0.296 / 0.307 / 0.316 ms as a graph replay: the pass that adds the groups reads `split` buffers of 16 MiB)
            if (split > 4)
                split = 4;
            if (const char *f = std::getenv("SEALHIP_KS_SPLIT"))
                split = (unsigned)std::atoi(f);
            if (split > 8)
                split = 8; // eight canonical residues below 2^61 still fit a 64-bit word
            if (split > K)
                split = K;
            if (split < 1)
                split = 1;
        }
        return split;
    }
    // CKKS on the fused path: the sums leave the key switch with the ciphertext's words already added (c + S P^-1), so that the
    // tail reads one operand per component instead of two - from ks2's epilogue when the digits run as one group, from the pass
    // that adds the groups otherwise.  SEALHIP_KS_NO_FOLD=1 (development builds): the round-3 form
    bool Evaluator::ks_folds(const KSwitchKeys &keys, size_t key_index, unsigned K) const
    {
        static const bool lazy_ok = !std::getenv("SEALHIP_KS_EAGER_TAIL");
        static const bool fold_ok = !shl_ab_getenv("SEALHIP_KS_NO_FOLD");
        return fold_ok && lazy_ok && context_.scheme() == Scheme::ckks && ntt2_supports(context_.log_n()) && K >= 2 && keys.context() == &context_ &&
               key_index < keys.slots() && keys.has_key(key_index) && keys.key(key_index).register_order;
    }

    void Evaluator::switch_key_inplace(Ciphertext &e, const uint64_t *target, const KSwitchKeys &keys, size_t key_index, bool c1_zero_unwritten,
                                       uint32_t galois_elt, const uint64_t *galois_c0) const
    {
        StreamScope pool_scope(stream_); // pool traffic of this call is ordered on the evaluator's stream whoever calls (pool.h)
        if (&e.context() != &context_ || !e.level())
            throw std::invalid_argument("encrypted is not valid for encryption parameters");
        const unsigned K = e.level()->K;
        const unsigned split = ks_split(e, keys, key_index);
        // CKKS at the two-pass sizes: leave the mod-down to whoever touches the ciphertext next (LazyTail) - a rescale on this
        // evaluator then does both rounding divisions with one transform per component.  SEALHIP_KS_EAGER_TAIL=1: always now.
        // BFV at the same sizes (round 4): a mod_switch_to_next on this evaluator folds the mod-down into its own division (one
        // element-wise pass after the inverse transforms, switch_key_finish_modswitch_bfv); anything else completes it first.
        static const bool lazy_ok = !std::getenv("SEALHIP_KS_EAGER_TAIL");
        const Scheme sch = context_.scheme();
        const bool defer = lazy_ok && (sch == Scheme::ckks || sch == Scheme::bfv) && ntt2_supports(context_.log_n()) && K >= 2;
        const bool fold = ks_folds(keys, key_index, K);
        if (galois_elt && !(fold && split == 1 && c1_zero_unwritten))
            throw std::logic_error("the automorphism is read inside the key switch only on its un-split folded path");
        // c1_zero_unwritten (rotations: the ciphertext is (pi(c0), 0) and its second polynomial has not been written): with the addend
        // folded into the sums nobody reads that polynomial before the tail writes it - the zeros are neither stored nor loaded
        if (c1_zero_unwritten && !fold)
            ck(hipMemsetAsync(e.plane(1), 0, e.plane_words() * 8, stream_), "zero c1");
        const bool a1z = c1_zero_unwritten && fold;
        Scratch acc(switch_key_acc_words(e) * split);
        switch_key_partial(e, target, keys, key_index, 0, K, acc.p, split, fold && split == 1, nullptr, a1z, galois_elt, galois_c0);
        if (split > 1) // several digit groups (small batches): the pass that adds them adds the ciphertext's words too
            ck(k_keyswitch_reduce(context_.dev_mods(), acc.p, (unsigned)context_.log_n(), K, context_.key_level().K, (unsigned)e.batch(),
                                  stream_, split, fold ? e.plane(0) : nullptr, fold && !a1z ? e.plane(1) : nullptr,
                                  context_.key_level().dev.inv_q_last_mod_q),
               "ks add digit groups");
        if (defer)
            defer_tail(e, acc.release(), fold);
        else
            switch_key_finish(e, acc.p, 1);
    }

    // BFV: relinearize (or a rotation) followed by mod_switch_to_next.  acc = the key-switch sums (NTT form, K + 1 components per
    // polynomial), planes 0 and 1 of e = the addends (coefficient form).  Reference steps being folded: evaluator.cpp:2806-2864
    // (inverse transforms of the sums, mod-down by P), then rns.cpp:789-828 (divide_and_round_q_last_inplace) on the result.
    void Evaluator::switch_key_finish_modswitch_bfv(Ciphertext &e, uint64_t *acc_p, const Level *next) const
    {
        StreamScope pool_scope(stream_); // pool traffic of this call is ordered on the evaluator's stream whoever calls (pool.h)
        const Level &lvl = *e.level();
        const Level &klvl = context_.key_level();
        const unsigned K = lvl.K, L = klvl.K;
        const size_t N = context_.n();
        const unsigned B = (unsigned)e.batch();
        const NttTables &tb = context_.ntt_tables();
        const uint64_t P = context_.coeff_modulus()[L - 1];
        static const bool trace = shl_ab_getenv("SEALHIP_KS_TRACE") != nullptr;
        if (trace)
            std::fprintf(stderr, "[ks] folded tail (bfv)\n");
        g_tail_folded++;
        NttBatch bi = plain_batch(acc_p, (size_t)(K + 1) * N, K + 1, 2 * B, 0);
        bi.comp_prime = ks_comp_prime(K) + (size_t)(K + 1) * K; // components 0 .. K-1 and the special prime
        bi.cls_hint = ks_class_hint(K);
        ck(ntt_inverse(tb, bi, 0, stream_), "ks intt all");
        const size_t words = (size_t)2 * B * (K - 1) * N;
        uint64_t *out = DevicePool::global().alloc_words(words, stream_);
        const uint64_t *c0 = e.data_, *c1 = e.data_ + (size_t)B * K * N; // (no deferred tail left on e: the caller detached it)
        hipError_t err = k_keyswitch_tail_modswitch_bfv(context_.dev_mods(), klvl.dev, P >> 1, P, lvl.dev, c0, c1, acc_p, out,
                                                        (unsigned)context_.log_n(), K, B, stream_);
        if (err != hipSuccess)
        {
            DevicePool::global().free_words(out, stream_);
            ck(err, "ks tail + mod switch (bfv)");
        }
        e.adopt(next, 2, out, words);
    }

    // relinearize (or a rotation) followed by rescale_to_next: acc = the key-switch sums, planes 0 and 1 of e = the addends.
    // Reference steps being folded: evaluator.cpp:2806-2864 (mod-down by the special prime P), then rns.cpp:830-901 on the result
    // (divide_and_round_q_last_ntt_inplace); see NttTail2 (ntt_kernels.h) for the algebra.
    void Evaluator::switch_key_finish_rescale(Ciphertext &e, uint64_t *acc_p, const Level *next, double destination_scale, bool acc_has_addend) const
    {
        StreamScope pool_scope(stream_); // pool traffic of this call is ordered on the evaluator's stream whoever calls (pool.h)
        const Level &lvl = *e.level();
        const Level &klvl = context_.key_level();
        const unsigned K = lvl.K, L = klvl.K;
        const size_t N = context_.n();
        const unsigned B = (unsigned)e.batch();
        const NttTables &tb = context_.ntt_tables();
        const uint64_t P = context_.coeff_modulus()[L - 1];
        uint64_t *c0 = e.data_, *c1 = e.data_ + (size_t)B * K * N; // no deferred tail left on e: the caller detached it
        static const bool trace = shl_ab_getenv("SEALHIP_KS_TRACE") != nullptr;
        if (trace)
            std::fprintf(stderr, acc_has_addend ? "[ks] folded tail, addend in the sums\n" : "[ks] folded tail\n");
        g_tail_folded++;

        // t_P: coefficient form of the special-prime sums, in place (component K of every (item, plane) of acc)
        // plus P/2: the rounding's addend goes in here, once per coefficient (NttBatch::out_add; the maps below run in mode 3)
        NttBatch bi = plain_batch(acc_p + (size_t)K * N, (size_t)(K + 1) * N, 1, 2 * B, L - 1);
        bi.out_add = P >> 1;
        // (round 6, measured and not kept: at small batches this transform and the one of the last component below - different components,
        // 2 B x 16 workgroups each - side by side on a forked lane: batch 1 eager 0.328 -> 0.360 ms because the pool has to order the lane's
        // scratch across streams, graph replay 0.296 either way; profiles/r06_latency.txt)
        ck(ntt_inverse(tb, bi, 0, stream_), "ks intt special");

        if (acc_has_addend)
        {
            // acc's data-prime components are A = c + S P^-1 (switch_key_partial, fold_addend).  The relinearised ciphertext's last
            // component is A - NTT(v) P^-1; its coefficient form is INTT(A) - v P^-1 because the transform is linear: one inverse
            // transform in place and one element-wise pass, no forward transform of v and nothing written to e's planes
            NttBatch bw = plain_batch(acc_p + (size_t)(K - 1) * N, (size_t)(K + 1) * N, 1, 2 * B, K - 1);
            ck(ntt_inverse(tb, bw, 0, stream_), "ks intt last component");
            ck(k_ks_last_coeff(context_.dev_mods(), K - 1, klvl.dev.inv_q_last_mod_q + (K - 1), klvl.dev.round_fix + (K - 1),
                               lvl.dev.half_q_last, acc_p, (unsigned)context_.log_n(), K, (size_t)2 * B, stream_),
               "ks last component, coefficient form");
        }
        else
        {
            // the relinearised ciphertext's LAST component (the one rescale divides by), completed alone: c += (S - NTT(v)) P^-1
            NttBatch b{};
            b.data = nullptr;
            b.outer_stride = N;
            b.ncomp = 1;
            b.nouter = 2 * B;
            b.comp_prime = nullptr;
            b.prime_first = K - 1;
            b.src = acc_p + (size_t)K * N;
            b.src_outer_stride = (size_t)(K + 1) * N;
            b.src_ncomp = 1;
            b.src_mode = 3;
            b.src_half = P >> 1;
            b.src_q = P;
            b.src_fix = klvl.dev.round_fix + (K - 1);
            b.epi = 2;
            b.epi_a = acc_p + (size_t)(K - 1) * N;
            b.epi_a_stride = (size_t)(K + 1) * N;
            b.epi_mul = klvl.dev.inv_q_last_mod_q + (K - 1);
            b.epi_out0 = c0 + (size_t)(K - 1) * N;
            b.epi_out1 = c1 + (size_t)(K - 1) * N;
            b.epi_out_stride = (size_t)K * N;
            ck(ntt_forward(tb, b, 1, stream_), "ks ntt correction + tail, last component");
        }
        // t_last: its coefficient form, in place (that component is dropped by the rescale)
        if (!acc_has_addend)
        {
            NttBatch bl = plain_batch(c0 + (size_t)(K - 1) * N, (size_t)K * N, 1, 2 * B, K - 1);
            bl.out_add = lvl.dev.half_q_last;
            ck(ntt_inverse(tb, bl, 0, stream_), "rescale intt last");
        }

        // components 0 .. K-2: out = (c + S P^-1 - NTT(v P^-1 + u)) q_last^-1, one transform each
        const size_t words = (size_t)2 * B * (K - 1) * N;
        uint64_t *out = DevicePool::global().alloc_words(words);
        try
        {
            NttTail2 t2{};
            t2.src2_0 = c0 + (size_t)(K - 1) * N;
            t2.src2_1 = c1 + (size_t)(K - 1) * N;
            t2.src2_stride = (size_t)K * N;
            t2.src2_half = lvl.dev.half_q_last;
            t2.src2_q = lvl.dev.q_last;
            t2.src2_fix = lvl.dev.round_fix;
            t2.pmul = klvl.dev.inv_q_last_mod_q;
            t2.c0 = c0;
            t2.c1 = c1;
            t2.c_stride = (size_t)K * N;
            t2.halves_added = 1;
            if (acc_has_addend)
            {
                // t_last sits in acc (component K - 1 of every (item, plane)); the sums carry the addend
                t2.src2_0 = acc_p + (size_t)(K - 1) * N;
                t2.src2_1 = acc_p + (size_t)(K + 1) * N + (size_t)(K - 1) * N;
                t2.src2_stride = (size_t)2 * (K + 1) * N;
                t2.a_has_c = 1;
            }
            NttBatch b{};
            b.data = nullptr;
            b.outer_stride = (size_t)(K - 1) * N;
            b.ncomp = K - 1;
            b.nouter = 2 * B;
            b.comp_prime = nullptr;
            b.prime_first = 0;
            b.src = acc_p + (size_t)K * N;
            b.src_outer_stride = (size_t)(K + 1) * N;
            b.src_ncomp = 1;
            b.src_mode = 3;
            b.src_half = P >> 1;
            b.src_q = P;
            b.src_fix = klvl.dev.round_fix;
            b.epi = 0;
            b.epi_a = acc_p;
            b.epi_a_stride = (size_t)(K + 1) * N;
            b.epi_mul = lvl.dev.inv_q_last_mod_q;
            b.epi_out0 = out;
            b.epi_out1 = out + (size_t)B * (K - 1) * N;
            b.epi_out_stride = (size_t)(K - 1) * N;
            b.tail2 = &t2;
            ck(ntt_forward(tb, b, 1, stream_), "mod-down + rescale in one transform");
        }
        catch (...)
        {
            DevicePool::global().free_words(out);
            throw;
        }
        e.adopt(next, 2, out, words);
        e.scale() = destination_scale;
    }

    // ---- digit-parallel key switching over the ranks of a communicator (the reference loop being split:
    // evaluator.cpp:2663-2755 over the digits, 2806-2864 over the target moduli).  Everything is enqueued on stream_.
    unsigned Evaluator::switch_key_slots(const Ciphertext &e, unsigned nranks) const
    {
        if (&e.context() != &context_ || !e.level())
            throw std::invalid_argument("encrypted is not valid for encryption parameters");
        if (nranks < 1 || nranks > 8)
            throw std::invalid_argument("nranks");
        return (e.level()->K + nranks - 1) / nranks;
    }

    void Evaluator::switch_key_pack_targets(const Ciphertext &e, const uint64_t *acc, unsigned nranks, uint64_t *send, uint64_t *sp) const
    {
        StreamScope pool_scope(stream_); // pool traffic of this call is ordered on the evaluator's stream whoever calls (pool.h)
        const unsigned m = switch_key_slots(e, nranks);
        if (!acc || !send || !sp)
            throw std::invalid_argument("buffer");
        ck(k_ks_pack_targets(acc, send, sp, (unsigned)context_.log_n(), e.level()->K, nranks, m, (unsigned)e.batch(), stream_), "ks pack targets");
    }

    void Evaluator::switch_key_finish_owned(
        const Ciphertext &e, const uint64_t *recv, const uint64_t *sp, unsigned nranks, unsigned rank, uint64_t *own) const
    {
        StreamScope pool_scope(stream_); // pool traffic of this call is ordered on the evaluator's stream whoever calls (pool.h)
        const unsigned m = switch_key_slots(e, nranks);
        if (rank >= nranks)
            throw std::invalid_argument("rank");
        if (!recv || !sp || !own)
            throw std::invalid_argument("buffer");
        if (context_.scheme() != Scheme::ckks)
            throw std::logic_error("the reduce-scatter exchange is built for CKKS; BFV / BGV use the all-reduce exchange");
        if (!context_.using_keyswitching())
            throw std::logic_error("keyswitching is not supported by the context");
        const Level &lvl = *e.level();
        const Level &klvl = context_.key_level();
        const unsigned K = lvl.K, L = klvl.K, B = (unsigned)e.batch(), n_log = (unsigned)context_.log_n();
        const size_t N = context_.n();
        const NttTables &tb = context_.ntt_tables();
        const ModDesc *mods = context_.dev_mods();
        unsigned first, count;
        comm_split(K, nranks, rank, first, count);
        if (!count)
        {
            // more ranks than moduli: nothing to reduce here, the chunk this rank contributes is zero
            ck(hipMemsetAsync(own, 0, (size_t)m * B * 2 * N * 8, stream_), "ks zero own");
            return;
        }
        // a key switch over this rank's `count` moduli: sums [batch][2][count+1][N], the special prime last
        Scratch acc3((size_t)B * 2 * (count + 1) * N);
        ck(k_ks_unpack_owned(mods, recv, sp, acc3.p, n_log, L, first, count, B, stream_), "ks unpack owned");
        const uint64_t P = context_.coeff_modulus()[L - 1];
        NttBatch bi = plain_batch(acc3.p + (size_t)count * N, (size_t)(count + 1) * N, 1, 2 * B, L - 1);
        ck(ntt_inverse(tb, bi, 0, stream_), "ks intt special");
        // increments of the owned moduli, compact planes [2][batch][count][N] (the tail adds into them: start from zero)
        Scratch inc((size_t)2 * B * count * N);
        ck(hipMemsetAsync(inc.p, 0, (size_t)2 * B * count * N * 8, stream_), "ks zero increments");
        Scratch tt(ntt2_supports(context_.log_n()) ? 1 : (size_t)B * 2 * count * N);
        NttBatch b{};
        b.data = tt.p;
        b.outer_stride = (size_t)count * N;
        b.ncomp = count;
        b.nouter = 2 * B;
        b.comp_prime = nullptr;
        b.prime_first = first;
        b.src = acc3.p + (size_t)count * N;
        b.src_outer_stride = (size_t)(count + 1) * N;
        b.src_ncomp = 1;
        b.src_mode = 2;
        b.src_half = P >> 1;
        b.src_q = P;
        b.src_fix = klvl.dev.round_fix + first;
        uint64_t *inc0 = inc.p, *inc1 = inc.p + (size_t)B * count * N;
        if (ntt2_supports(context_.log_n()))
        {
            b.data = nullptr;
            b.epi = 2;
            b.epi_a = acc3.p;
            b.epi_a_stride = (size_t)(count + 1) * N;
            b.epi_mul = klvl.dev.inv_q_last_mod_q + first;
            b.epi_out0 = inc0;
            b.epi_out1 = inc1;
            b.epi_out_stride = (size_t)count * N;
            ck(ntt_forward(tb, b, 1, stream_), "ks ntt correction + tail (owned moduli)");
        }
        else
        {
            ck(ntt_forward(tb, b, 1, stream_), "ks ntt correction (owned moduli)");
            ck(k_keyswitch_tail_ckks(mods + first, klvl.dev.inv_q_last_mod_q + first, inc0, inc1, acc3.p, tt.p, n_log, count, B, stream_),
               "ks tail (owned moduli)");
        }
        ck(k_ks_pack_owned(inc.p, own, n_log, count, m, B, stream_), "ks pack owned");
    }

    void Evaluator::switch_key_add_gathered(Ciphertext &e, const uint64_t *all, unsigned nranks) const
    {
        StreamScope pool_scope(stream_); // pool traffic of this call is ordered on the evaluator's stream whoever calls (pool.h)
        const unsigned m = switch_key_slots(e, nranks);
        if (!all)
            throw std::invalid_argument("buffer");
        if (e.size() < 2)
            throw std::invalid_argument("encrypted size must be at least 2");
        ck(k_ks_add_gathered(context_.dev_mods(), e.plane(0), e.plane(1), all, (unsigned)context_.log_n(), e.level()->K, nranks, m,
                             (unsigned)e.batch(), stream_),
           "ks add gathered");
    }

    void Evaluator::switch_key_exchange_finish(Ciphertext &e, uint64_t *acc, Comm &comm, KsExchange how) const
    {
        StreamScope pool_scope(stream_); // pool traffic of this call is ordered on the evaluator's stream whoever calls (pool.h)
        const unsigned G = (unsigned)comm.size();
        const size_t words = switch_key_acc_words(e);
        if (how == KsExchange::all_reduce || context_.scheme() != Scheme::ckks)
        {
            comm.all_reduce_sum(acc, words, stream_);
            switch_key_finish(e, acc, G, false, true);
            return;
        }
        const unsigned m = switch_key_slots(e, G);
        const size_t N = context_.n(), B = e.batch();
        const size_t chunk = (size_t)m * B * 2 * N, spw = B * 2 * N;
        Scratch send((size_t)G * chunk), sp(spw), recv(chunk), own(chunk), all((size_t)G * chunk);
        switch_key_pack_targets(e, acc, G, send.p, sp.p);
        comm.reduce_scatter_sum(send.p, recv.p, chunk, stream_);
        comm.all_reduce_sum(sp.p, spw, stream_);
        switch_key_finish_owned(e, recv.p, sp.p, G, (unsigned)comm.rank(), own.p);
        comm.all_gather(own.p, all.p, chunk, stream_);
        switch_key_add_gathered(e, all.p, G);
    }

    void Evaluator::relinearize_inplace(Ciphertext &e, const KSwitchKeys &relin_keys, Comm &comm, KsExchange how) const
    {
        StreamScope pool_scope(stream_); // pool traffic of this call is ordered on the evaluator's stream whoever calls (pool.h)
        if (&e.context() != &context_ || !e.level())
            throw std::invalid_argument("encrypted is not valid for encryption parameters");
        unsigned first, count;
        comm_split(e.level()->K, (unsigned)comm.size(), (unsigned)comm.rank(), first, count);
        Scratch acc(switch_key_acc_words(e));
        relinearize_partial(e, relin_keys, first, first + count, acc.p);
        switch_key_exchange_finish(e, acc.p, comm, how);
        e.resize(e.level(), 2, stream_);
        throw_if_transparent(e);
    }

    void Evaluator::apply_galois_inplace(Ciphertext &e, uint32_t galois_elt, const KSwitchKeys &galois_keys, Comm &comm, KsExchange how) const
    {
        StreamScope pool_scope(stream_); // pool traffic of this call is ordered on the evaluator's stream whoever calls (pool.h)
        if (&e.context() != &context_ || !e.level())
            throw std::invalid_argument("encrypted is not valid for encryption parameters");
        unsigned first, count;
        comm_split(e.level()->K, (unsigned)comm.size(), (unsigned)comm.rank(), first, count);
        Scratch acc(switch_key_acc_words(e));
        apply_galois_partial(e, galois_elt, galois_keys, first, first + count, acc.p);
        switch_key_exchange_finish(e, acc.p, comm, how);
        throw_if_transparent(e);
    }

    void Evaluator::rotate_vector_inplace(Ciphertext &e, int steps, const KSwitchKeys &galois_keys, Comm &comm, KsExchange how) const
    {
        StreamScope pool_scope(stream_); // pool traffic of this call is ordered on the evaluator's stream whoever calls (pool.h)
        if (context_.scheme() != Scheme::ckks)
            throw std::logic_error("unsupported scheme");
        if (&e.context() != &context_ || !e.level())
            throw std::invalid_argument("encrypted is not valid for encryption parameters");
        if (steps == 0)
            return;
        // the digit-parallel form takes the exact key (evaluator.h:1209 with the key present); the NAF fallback of
        // rotate_internal would need every rank to hold the power-of-two keys' digits as well
        apply_galois_inplace(e, galois_elt_from_step(steps), galois_keys, comm, how);
    }

    void Evaluator::broadcast_key_digits(KSwitchKeys &keys, size_t index, uint64_t *staging, Comm &comm, int root) const
    {
        if (!staging)
            throw std::invalid_argument("staging");
        if (!context_.using_keyswitching())
            throw std::logic_error("keyswitching is not supported by the context");
        const size_t N = context_.n(), L = context_.key_level().K, K = context_.first_level().K;
        const size_t digit_words = 2 * L * N;
        comm.broadcast(staging, K * digit_words, root, stream_);
        ck(hipStreamSynchronize(stream_), "broadcast key");
        unsigned first, count;
        comm_split((unsigned)K, (unsigned)comm.size(), (unsigned)comm.rank(), first, count);
        if (count)
            keys.set_key(context_, index, count, staging + first * digit_words, true, first);
    }

    // NTT the BGV correction polynomials `delta` ([items][ncomp][N], coefficient form, canonical) and fold them
    // into the resident operand: v = (A - NTT(delta)) * mul  (mod q_i), A = a + item*a_stride + comp*N;
    //   epi 1: out0[item][comp] = v;   epi 2: ct_{item&1}[item>>1][comp] += v
    void Evaluator::bgv_correct_and_combine(
        Scratch &delta, const uint64_t *a, size_t a_stride, const ShoupOp *mul, unsigned ncomp, size_t items, uint64_t *out0,
        uint64_t *out1, size_t out_stride, int epi) const
    {
        const size_t N = context_.n();
        const unsigned n_log = (unsigned)context_.log_n();
        const NttTables &tb = context_.ntt_tables();
        const ModDesc *mods = context_.dev_mods();
        if (ntt2_supports(context_.log_n()))
        {
            NttBatch b{};
            b.data = nullptr;
            b.outer_stride = (size_t)ncomp * N;
            b.ncomp = ncomp;
            b.nouter = (unsigned)items;
            b.prime_first = 0;
            b.src = delta.p;
            b.src_outer_stride = (size_t)ncomp * N;
            b.src_ncomp = ncomp;
            b.src_mode = 0;
            b.epi = epi;
            b.epi_a = a;
            b.epi_a_stride = a_stride;
            b.epi_mul = mul;
            b.epi_out0 = out0;
            b.epi_out1 = out1;
            b.epi_out_stride = out_stride;
            ck(ntt_forward(tb, b, 1, stream_), "bgv ntt correction + combine");
            return;
        }
        ck(ntt_forward(tb, plain_batch(delta.p, (size_t)ncomp * N, ncomp, (unsigned)items, 0), 1, stream_), "bgv ntt correction");
        if (epi == 1)
        {
            if (a_stride != (size_t)(ncomp + 1) * N)
                throw std::logic_error("bgv combine layout");
            ck(k_rescale_combine(mods, mul, a, delta.p, out0, n_log, ncomp + 1, items, stream_), "bgv combine");
        }
        else
            ck(k_keyswitch_tail_ckks(mods, mul, out0, out1, a, delta.p, n_log, ncomp, (unsigned)(items / 2), stream_), "bgv ks tail");
    }

} // namespace sealhip
