// Evaluator, part 1: construction, streams and graph capture, deferred key-switch tails, negate / add / sub, transforms, plaintext operands, multiply
#include "evaluator_common.h"
#include <atomic>

namespace sealhip
{
    // ---------------------------------------------------------------- Evaluator
    Evaluator::Evaluator(const Context &context) : context_(context)
    {
        void *p = nullptr;
        ck(hipMalloc(&p, sizeof(unsigned)), "hipMalloc flag");
        d_flag_ = (unsigned *)p;
    }
    // ---- deferred key-switch tails (LazyTail, evaluator.h)
    std::atomic<uint64_t> g_tail_folded{ 0 }, g_tail_plain{ 0 }, g_tail_dropped{ 0 };
    void lazy_tail_stats(uint64_t &folded, uint64_t &plain, uint64_t &dropped)
    {
        folded = g_tail_folded.load();
        plain = g_tail_plain.load();
        dropped = g_tail_dropped.load();
    }
    void Evaluator::defer_tail(Ciphertext &e, uint64_t *acc, bool with_addend) const
    {
        StreamScope pool_scope(stream_); // pool traffic of this call is ordered on the evaluator's stream whoever calls (pool.h)
        e.lazy_ = new LazyTail{ this, acc, with_addend };
        std::lock_guard<std::mutex> lock(lazy_mu_);
        lazy_cts_.push_back(&e);
    }
    LazyTail Evaluator::detach_tail(Ciphertext &e) const
    {
        StreamScope pool_scope(stream_); // pool traffic of this call is ordered on the evaluator's stream whoever calls (pool.h)
        const LazyTail t = *e.lazy_;
        delete e.lazy_;
        e.lazy_ = nullptr;
        std::lock_guard<std::mutex> lock(lazy_mu_);
        lazy_cts_.erase(std::remove(lazy_cts_.begin(), lazy_cts_.end(), &e), lazy_cts_.end());
        return t;
    }
    void Evaluator::forget_tail(const Ciphertext &e, LazyTail t) const
    {
        StreamScope pool_scope(stream_); // pool traffic of this call is ordered on the evaluator's stream whoever calls (pool.h)
        {
            std::lock_guard<std::mutex> lock(lazy_mu_);
            lazy_cts_.erase(std::remove(lazy_cts_.begin(), lazy_cts_.end(), &e), lazy_cts_.end());
        }
        DevicePool::global().free_words(t.acc, stream_);
        g_tail_dropped++;
    }
    void Evaluator::complete_tail(Ciphertext &e, LazyTail t) const
    {
        const hipStream_t caller = DevicePool::thread_stream(); // (read before this call's own scope replaces it)
        StreamScope pool_scope(stream_); // pool traffic of this call is ordered on the evaluator's stream whoever calls (pool.h)
        {
            std::lock_guard<std::mutex> lock(lazy_mu_);
            lazy_cts_.erase(std::remove(lazy_cts_.begin(), lazy_cts_.end(), &e), lazy_cts_.end());
        }
        // the sums were produced on this evaluator's stream: the tail runs there too; a caller working on another stream
        // (another evaluator, a host copy) continues only when it is done
        static const bool trace = shl_ab_getenv("SEALHIP_KS_TRACE") != nullptr; // tests: which tail ran
        if (trace)
            std::fprintf(stderr, t.with_addend ? "[ks] plain tail, addend in the sums\n" : "[ks] plain tail\n");
        g_tail_plain++;
        try
        {
            switch_key_finish(e, t.acc, 1, t.with_addend);
        }
        catch (...)
        {
            DevicePool::global().free_words(t.acc, stream_);
            throw;
        }
        DevicePool::global().free_words(t.acc, stream_);
        if (caller != stream_ && !capturing_) // (a stream that is recording cannot be waited for: its work runs when the graph does)
            ck(hipStreamSynchronize(stream_), "deferred key-switch tail");
    }
    // ---- deferred tensor products (evaluator.h: LazyProduct)
    namespace
    {
        std::atomic<uint64_t> g_prod_fused{ 0 }, g_prod_formed{ 0 }, g_prod_dropped{ 0 };
    }
    void lazy_product_stats(uint64_t &fused, uint64_t &formed, uint64_t &dropped)
    {
        fused = g_prod_fused.load();
        formed = g_prod_formed.load();
        dropped = g_prod_dropped.load();
    }
    // the launcher's rule: CKKS at a two-pass size, a batch whose key switch runs un-split (the fused relinearisation's configuration)
    bool Evaluator::may_defer_product(const Level &lvl, size_t batch) const
    {
        const char *lazy_env = std::getenv("SEALHIP_LAZY_PRODUCT"); // (read per call: the tests switch it)
        if (lazy_env && std::atoi(lazy_env) == 0)
            return false;
        // (SEALHIP_LAZY_PRODUCT_MIN_WGS: tests reach the fused path at small batches together with SEALHIP_KS_SPLIT=1)
        const char *min_env = std::getenv("SEALHIP_LAZY_PRODUCT_MIN_WGS");
        const size_t min_wgs = min_env ? (size_t)std::atol(min_env) : 1024;
        return !capturing_ && !transparent_check_ && lvl.K >= 2 && ntt2_supports(context_.log_n()) &&
               batch * (lvl.K + 1) * (context_.n() >> 12) > min_wgs;
    }
    void Evaluator::defer_product(Ciphertext &dest, const Ciphertext *x, const Ciphertext *y, uint64_t *own) const
    {
        dest.lazy_prod_ = new LazyProduct{ this, x, y, own };
        lazy_product_link(&dest, *dest.lazy_prod_);
        std::lock_guard<std::mutex> lock(lazy_mu_);
        lazy_cts_.push_back(&dest);
    }
    // (called by Ciphertext::settle_product, which unlinks the operands and deletes the record afterwards)
    void Evaluator::complete_product(Ciphertext &dest, LazyProduct p) const
    {
        const hipStream_t caller = DevicePool::thread_stream();
        StreamScope pool_scope(stream_);
        {
            std::lock_guard<std::mutex> lock(lazy_mu_);
            lazy_cts_.erase(std::remove(lazy_cts_.begin(), lazy_cts_.end(), &dest), lazy_cts_.end());
        }
        const Level &lvl = *dest.level_;
        PlaneGeom g{ (unsigned)context_.log_n(), lvl.K, (unsigned)dest.batch() };
        g_prod_formed++;
        hipError_t err = hipSuccess;
        try
        {
            err = k_ckks_multiply_2x2(context_.dev_mods(), context_.ntt_tables().fpd, nullptr, p.xw(), p.yw(), dest.data_, g, stream_);
        }
        catch (...)
        {
            DevicePool::global().free_words(p.own, stream_);
            throw;
        }
        DevicePool::global().free_words(p.own, stream_); // (an in-place product's previous slab: read by the kernel just queued on this stream)
        ck(err, "ckks_multiply (deferred)");
        // the product is formed on this evaluator's stream; a caller on another stream - the one about to read the words or to
        // overwrite an operand - continues only when it is done
        if (caller != stream_ && !capturing_)
            ck(hipStreamSynchronize(stream_), "deferred tensor product");
    }
    // the fused relinearisation takes the record over: the destination is no longer pending, the live operands are released (an owned
    // slab is now the caller's to free)
    LazyProduct Evaluator::detach_product(Ciphertext &dest) const
    {
        const LazyProduct p = *dest.lazy_prod_;
        delete dest.lazy_prod_;
        dest.lazy_prod_ = nullptr;
        lazy_product_unlink(&dest, p);
        std::lock_guard<std::mutex> lock(lazy_mu_);
        lazy_cts_.erase(std::remove(lazy_cts_.begin(), lazy_cts_.end(), &dest), lazy_cts_.end());
        return p;
    }
    void Evaluator::forget_product(const Ciphertext &dest, LazyProduct p) const
    {
        StreamScope pool_scope(stream_);
        {
            std::lock_guard<std::mutex> lock(lazy_mu_);
            lazy_cts_.erase(std::remove(lazy_cts_.begin(), lazy_cts_.end(), &dest), lazy_cts_.end());
        }
        DevicePool::global().free_words(p.own, stream_);
        g_prod_dropped++;
    }
    void lazy_product_count_fused()
    {
        g_prod_fused++;
    }

    void Evaluator::settle_all() const
    {
        for (;;)
        {
            const Ciphertext *c = nullptr;
            {
                std::lock_guard<std::mutex> lock(lazy_mu_);
                if (lazy_cts_.empty())
                    return;
                c = lazy_cts_.back();
            }
            c->settle(); // removes it from the list
        }
    }

    Evaluator::~Evaluator()
    {
        try
        {
            settle_all();
        }
        catch (...)
        {
            // a tail could not be completed (device error): no ciphertext may keep a LazyTail whose owner is gone - the pending
            // sums are discarded, the objects stay what they were before their key switch's mod-down (their next use is still
            // memory-safe; the device error itself is what the caller sees on its next call)
            for (;;)
            {
                const Ciphertext *c = nullptr;
                {
                    std::lock_guard<std::mutex> lock(lazy_mu_);
                    if (lazy_cts_.empty())
                        break;
                    c = lazy_cts_.back();
                }
                const_cast<Ciphertext *>(c)->drop_product(); // either removes it from the list
                const_cast<Ciphertext *>(c)->drop_lazy();
            }
        }
        for (auto &kv : ks_maps_)
            (void)hipFree(kv.second);
        for (auto &kv : ks_targets_)
            (void)hipFree(kv.second.dev);
        if (d_flag_)
            (void)hipFree(d_flag_);
        if (capture_stream_)
            (void)hipStreamDestroy(capture_stream_);
        DevicePool::global().unregister_stream(capturing_ ? saved_stream_ : stream_);
    }
    void Evaluator::set_stream(hipStream_t s)
    {
        if (capturing_)
            throw std::logic_error("a capture is in progress");
        if (s == stream_)
            return;
        settle_all(); // deferred tails belong to the stream their sums were produced on
        DevicePool::global().unregister_stream(stream_);
        stream_ = s;
        DevicePool::global().register_stream(stream_);
    }
    Evaluator::Graph::~Graph()
    {
        if (exec)
            (void)hipGraphExecDestroy(exec);
        DevicePool::global().release_held(scratch);
    }
    void Evaluator::begin_capture()
    {
        if (capturing_)
            throw std::logic_error("a capture is already in progress");
        if (transparent_check_)
            throw std::logic_error("the transparent-ciphertext check reads device memory back and cannot be captured");
        settle_all(); // deferred tails hold pool blocks of their own: complete them before the recording starts
        if (!capture_stream_)
            ck(hipStreamCreateWithFlags(&capture_stream_, hipStreamNonBlocking), "capture stream");
        // drain the device: every cached pool block is idle from here on, and the recording takes its scratch only from
        // idle or fresh blocks, which then belong to the graph (pool.h)
        ck(hipDeviceSynchronize(), "device synchronize");
        saved_stream_ = stream_;
        stream_ = capture_stream_;
        // relaxed: a pool miss may still hipMalloc while recording (it touches no stream)
        hipError_t e = hipStreamBeginCapture(stream_, hipStreamCaptureModeRelaxed);
        if (e != hipSuccess)
        {
            stream_ = saved_stream_;
            ck(e, "hipStreamBeginCapture");
        }
        DevicePool::global().begin_hold();
        capturing_ = true;
    }
    Evaluator::Graph *Evaluator::end_capture()
    {
        if (!capturing_)
            throw std::logic_error("no capture in progress");
        // a tail deferred inside the recording and not consumed by it runs as the recording's last work (on the capture stream)
        try
        {
            settle_all();
        }
        catch (...)
        {
            hipGraph_t dead = nullptr;
            (void)hipStreamEndCapture(stream_, &dead);
            if (dead)
                (void)hipGraphDestroy(dead);
            stream_ = saved_stream_;
            capturing_ = false;
            DevicePool::global().release_held(DevicePool::global().end_hold());
            throw;
        }
        hipGraph_t graph = nullptr;
        hipError_t e = hipStreamEndCapture(stream_, &graph);
        stream_ = saved_stream_;
        capturing_ = false;
        std::unique_ptr<Graph> g(new Graph);
        g->scratch = DevicePool::global().end_hold();
        ck(e, "hipStreamEndCapture");
        e = hipGraphInstantiate(&g->exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        ck(e, "hipGraphInstantiate");
        return g.release();
    }
    void Evaluator::launch_graph(const Graph *graph) const
    {
        if (capturing_)
            throw std::logic_error("a capture is in progress");
        if (!graph || !graph->exec)
            throw std::invalid_argument("graph");
        ck(hipGraphLaunch(graph->exec, stream_), "hipGraphLaunch");
    }

    void Evaluator::synchronize() const
    {
        ck(hipStreamSynchronize(stream_), "stream synchronize");
    }

    size_t Evaluator::relin_index(size_t key_power)
    {
        if (key_power < 2)
            throw std::invalid_argument("key_power cannot be less than 2");
        return key_power - 2;
    }
    size_t Evaluator::galois_index(uint32_t galois_elt)
    {
        if (!(galois_elt & 1))
            throw std::invalid_argument("galois_elt is not valid");
        return (galois_elt - 1) >> 1;
    }
    uint32_t Evaluator::galois_elt_from_step(int step) const
    {
        // GaloisTool::get_elt_from_step (util/galois.cpp:53-95), generator 3
        uint32_t n = (uint32_t)context_.n();
        uint32_t m32 = n * 2;
        uint64_t m = m32;
        if (step == 0)
            return (uint32_t)(m - 1);
        bool sign = step < 0;
        uint32_t pos_step = (uint32_t)std::abs(step);
        if (pos_step >= (n >> 1))
            throw std::invalid_argument("step count too large");
        pos_step &= m32 - 1;
        int s = sign ? (int)(n >> 1) - (int)pos_step : (int)pos_step;
        uint64_t elt = 1;
        while (s--)
        {
            elt *= 3;
            elt &= m - 1;
        }
        return (uint32_t)elt;
    }

    const uint32_t *Evaluator::ks_comp_prime(unsigned K) const
    {
        std::lock_guard<std::mutex> g(cache_mu_);
        auto it = ks_maps_.find(K);
        if (it != ks_maps_.end())
            return it->second;
        // [ (I*K + J) -> prime(I) for I in 0..K ] followed by [ i -> prime(i) for i in 0..K ]
        unsigned L = context_.key_level().K;
        std::vector<uint32_t> m;
        for (unsigned I = 0; I <= K; I++)
            for (unsigned J = 0; J < K; J++)
                m.push_back(I == K ? L - 1 : I);
        for (unsigned I = 0; I <= K; I++)
            m.push_back(I == K ? L - 1 : I);
        void *p = nullptr;
        ck(hipMalloc(&p, m.size() * 4), "hipMalloc ks map");
        ck(hipMemcpy(p, m.data(), m.size() * 4, hipMemcpyHostToDevice), "upload ks map");
        ks_maps_[K] = (uint32_t *)p;
        return (uint32_t *)p;
    }

    const Evaluator::KsTargets &Evaluator::ks_targets(unsigned K) const
    {
        std::lock_guard<std::mutex> g(cache_mu_);
        auto it = ks_targets_.find(K);
        if (it != ks_targets_.end())
            return it->second;
        // target moduli of a key switch at a level with K data primes: q_0..q_{K-1} and the special
        // prime (slot I = K, pool prime and key component L-1), split by arithmetic back end
        const unsigned L = context_.key_level().K;
        std::vector<uint32_t> t1[2], t2[2];
        for (unsigned I = 0; I <= K; I++)
        {
            const unsigned prime = I == K ? L - 1 : I;
            const int fp = context_.fp_prime(prime) ? 1 : 0;
            t1[fp].push_back(I);
            t1[fp].push_back(prime);
            t2[fp].push_back(I);
            t2[fp].push_back(prime);
            t2[fp].push_back(prime);
            t2[fp].push_back(key_comp_offset_units(context_.ntt_tables(), prime)); // where the component sits inside a digit (ntt2_kernels.h)
        }
        KsTargets kt;
        kt.n_int = (unsigned)t1[0].size() / 2;
        kt.n_fp = (unsigned)t1[1].size() / 2;
        // [targets1: int..., fp...][targets2: int..., fp...]
        std::vector<uint32_t> all;
        for (int fp = 0; fp < 2; fp++)
            all.insert(all.end(), t1[fp].begin(), t1[fp].end());
        for (int fp = 0; fp < 2; fp++)
            all.insert(all.end(), t2[fp].begin(), t2[fp].end());
        void *p = nullptr;
        ck(hipMalloc(&p, all.size() * 4 + 4), "hipMalloc ks targets");
        ck(hipMemcpy(p, all.data(), all.size() * 4, hipMemcpyHostToDevice), "upload ks targets");
        kt.dev = (uint32_t *)p;
        return ks_targets_[K] = kt;
    }

    // arithmetic class of the K data primes and the special prime together (NttBatch::cls_hint for batches that name them through
    // ks_comp_prime's map): 0 all on the integer back end, 1 all on the double-precision one, -1 mixed
    int Evaluator::ks_class_hint(unsigned K) const
    {
        const unsigned L = context_.key_level().K;
        unsigned fp = context_.fp_prime(L - 1) ? 1 : 0;
        for (unsigned i = 0; i < K; i++)
            fp += context_.fp_prime(i) ? 1 : 0;
        return fp == 0 ? 0 : (fp == K + 1 ? 1 : -1);
    }

    bool Evaluator::scale_within_bounds(double scale, const Level &lvl) const
    {
        // is_scale_within_bounds (evaluator.cpp:29-48)
        int bound = 0;
        switch (context_.scheme())
        {
        case Scheme::bfv:
        case Scheme::bgv:
            bound = host::bit_count(context_.plain_modulus());
            break;
        case Scheme::ckks:
            bound = lvl.total_coeff_modulus_bit_count;
            break;
        default:
            bound = -1;
        }
        return !(!std::isnormal(scale) || scale <= 0 || (static_cast<int>(std::log2(scale)) >= bound));
    }

    void Evaluator::check_valid(const Ciphertext &ct, const char *what) const
    {
        // is_metadata_valid_for + is_buffer_valid (valcheck.cpp:81-139, 221-237)
        bool ok = &ct.context() == &context_ && ct.level() != nullptr;
        if (ok)
        {
            const Level &l = *ct.level();
            ok = l.chain_index <= context_.first_level().chain_index;
            size_t size = ct.size();
            ok = ok && !((size < 2 && size != 0) || size > 16);
            double scale = ct.scale();
            Scheme s = context_.scheme();
            if (s == Scheme::bfv || s == Scheme::bgv)
                ok = ok && scale == 1.0;
            else
                ok = ok && (std::isnormal(scale) && scale > 0);
            uint64_t cf = ct.correction_factor();
            if (s == Scheme::bgv)
                ok = ok && !(cf == 0 || cf >= context_.plain_modulus());
            else
                ok = ok && cf == 1;
            ok = ok && (ct.word_count() == 0 || ct.has_storage());
            ok = ok && ct.word_count() <= ct.capacity_words(); // is_buffer_valid (valcheck.cpp:172-198): size x K x N words are there
        }
        if (!ok)
            throw std::invalid_argument(std::string(what) + " is not valid for encryption parameters");
    }

    bool Evaluator::is_transparent(const Ciphertext &ct) const
    {
        // Ciphertext::is_transparent (ciphertext.h:451-456)
        if (!ct.word_count() || ct.size() < 2)
            return true;
        Scratch word(1); // per call: concurrent checks on different ciphertexts must not share the flag
        unsigned *d_flag = reinterpret_cast<unsigned *>(word.p);
        unsigned zero = 0;
        ck(hipMemcpyAsync(d_flag, &zero, sizeof(zero), hipMemcpyHostToDevice, stream_), "flag reset");
        ck(k_any_nonzero(ct.plane(1), (ct.size() - 1) * ct.plane_words(), d_flag, stream_), "any_nonzero");
        unsigned flag = 0;
        ck(hipMemcpyAsync(&flag, d_flag, sizeof(flag), hipMemcpyDeviceToHost, stream_), "flag read");
        ck(hipStreamSynchronize(stream_), "flag sync");
        return flag == 0;
    }
    void Evaluator::throw_if_transparent(const Ciphertext &ct) const
    {
        StreamScope pool_scope(stream_); // pool traffic of this call is ordered on the evaluator's stream whoever calls (pool.h)
        if (transparent_check_ && is_transparent(ct))
            throw std::logic_error("result ciphertext is transparent");
    }

    // ---- negate / add / sub (evaluator.cpp:130-350)
    void Evaluator::negate_inplace(Ciphertext &e) const
    {
        StreamScope pool_scope(stream_); // pool traffic of this call is ordered on the evaluator's stream whoever calls (pool.h)
        check_valid(e, "encrypted");
        PlaneGeom g{ (unsigned)context_.log_n(), e.level()->K, (unsigned)e.batch() };
        if (e.size())
            ck(k_addsub(context_.dev_mods(), e.data(), nullptr, e.data(), 2, g, (unsigned)e.size(), stream_), "negate");
        throw_if_transparent(e);
    }

    void Evaluator::add_inplace(Ciphertext &e1, const Ciphertext &e2) const
    {
        StreamScope pool_scope(stream_); // pool traffic of this call is ordered on the evaluator's stream whoever calls (pool.h)
        check_valid(e1, "encrypted1");
        check_valid(e2, "encrypted2");
        if (e1.level() != e2.level())
            throw std::invalid_argument("encrypted1 and encrypted2 parameter mismatch");
        if (e1.is_ntt_form() != e2.is_ntt_form())
            throw std::invalid_argument("NTT form mismatch");
        if (!are_close(e1.scale(), e2.scale()))
            throw std::invalid_argument("scale mismatch");
        if (e1.batch() != e2.batch())
            throw std::invalid_argument("batch mismatch");
        if (e1.correction_factor() != e2.correction_factor())
        {
            // balance the correction factors and scale both operands first (BGV, evaluator.cpp:173-192)
            uint64_t f, m1, m2;
            balance_correction_factors(e1.correction_factor(), e2.correction_factor(), context_.plain_modulus(), f, m1, m2);
            PlaneGeom gg{ (unsigned)context_.log_n(), e1.level()->K, (unsigned)e1.batch() };
            if (e1.size())
                ck(k_mul_scalar(context_.dev_mods(), e1.data(), e1.data(), m1, gg, (unsigned)e1.size(), stream_), "add: scale encrypted1");
            Ciphertext copy(e2);
            if (copy.size())
                ck(k_mul_scalar(context_.dev_mods(), copy.data(), copy.data(), m2, gg, (unsigned)copy.size(), stream_), "add: scale encrypted2");
            e1.correction_factor() = f;
            copy.correction_factor() = f;
            add_inplace(e1, copy);
            return;
        }
        size_t s1 = e1.size(), s2 = e2.size();
        size_t mx = std::max(s1, s2), mn = std::min(s1, s2);
        e1.resize(e1.level(), mx, stream_);
        PlaneGeom g{ (unsigned)context_.log_n(), e1.level()->K, (unsigned)e1.batch() };
        if (mn)
            ck(k_addsub(context_.dev_mods(), e1.data(), e2.data(), e1.data(), 0, g, (unsigned)mn, stream_), "add");
        if (s1 < s2)
            ck(hipMemcpyAsync(e1.plane(s1), e2.plane(mn), (s2 - s1) * e1.plane_words() * 8, hipMemcpyDeviceToDevice, stream_),
               "add copy tail");
        throw_if_transparent(e1);
    }

    void Evaluator::sub_inplace(Ciphertext &e1, const Ciphertext &e2) const
    {
        StreamScope pool_scope(stream_); // pool traffic of this call is ordered on the evaluator's stream whoever calls (pool.h)
        check_valid(e1, "encrypted1");
        check_valid(e2, "encrypted2");
        if (e1.level() != e2.level())
            throw std::invalid_argument("encrypted1 and encrypted2 parameter mismatch");
        if (e1.is_ntt_form() != e2.is_ntt_form())
            throw std::invalid_argument("NTT form mismatch");
        if (!are_close(e1.scale(), e2.scale()))
            throw std::invalid_argument("scale mismatch");
        if (e1.batch() != e2.batch())
            throw std::invalid_argument("batch mismatch");
        if (e1.correction_factor() != e2.correction_factor())
        {
            // evaluator.cpp:259-278
            uint64_t f, m1, m2;
            balance_correction_factors(e1.correction_factor(), e2.correction_factor(), context_.plain_modulus(), f, m1, m2);
            PlaneGeom gg{ (unsigned)context_.log_n(), e1.level()->K, (unsigned)e1.batch() };
            if (e1.size())
                ck(k_mul_scalar(context_.dev_mods(), e1.data(), e1.data(), m1, gg, (unsigned)e1.size(), stream_), "sub: scale encrypted1");
            Ciphertext copy(e2);
            if (copy.size())
                ck(k_mul_scalar(context_.dev_mods(), copy.data(), copy.data(), m2, gg, (unsigned)copy.size(), stream_), "sub: scale encrypted2");
            e1.correction_factor() = f;
            copy.correction_factor() = f;
            sub_inplace(e1, copy);
            return;
        }
        size_t s1 = e1.size(), s2 = e2.size();
        size_t mx = std::max(s1, s2), mn = std::min(s1, s2);
        e1.resize(e1.level(), mx, stream_);
        PlaneGeom g{ (unsigned)context_.log_n(), e1.level()->K, (unsigned)e1.batch() };
        if (mn)
            ck(k_addsub(context_.dev_mods(), e1.data(), e2.data(), e1.data(), 1, g, (unsigned)mn, stream_), "sub");
        if (s1 < s2)
            ck(k_addsub(context_.dev_mods(), e2.plane(mn), nullptr, e1.plane(mn), 2, g, (unsigned)(s2 - mn), stream_), "sub negate tail");
        throw_if_transparent(e1);
    }

    // ---- transforms (evaluator.cpp:2289-2382)
    void Evaluator::transform_to_ntt_inplace(Ciphertext &e) const
    {
        StreamScope pool_scope(stream_); // pool traffic of this call is ordered on the evaluator's stream whoever calls (pool.h)
        check_valid(e, "encrypted");
        if (e.is_ntt_form())
            throw std::invalid_argument("encrypted is already in NTT form");
        unsigned K = e.level()->K;
        NttBatch b = plain_batch(e.data(), (size_t)K * context_.n(), K, (unsigned)(e.size() * e.batch()), 0);
        ck(ntt_forward(context_.ntt_tables(), b, 0, stream_), "ntt_forward");
        e.is_ntt_form() = true;
        throw_if_transparent(e);
    }
    void Evaluator::transform_from_ntt_inplace(Ciphertext &e) const
    {
        StreamScope pool_scope(stream_); // pool traffic of this call is ordered on the evaluator's stream whoever calls (pool.h)
        check_valid(e, "encrypted");
        if (!e.is_ntt_form())
            throw std::invalid_argument("encrypted_ntt is not in NTT form");
        unsigned K = e.level()->K;
        NttBatch b = plain_batch(e.data(), (size_t)K * context_.n(), K, (unsigned)(e.size() * e.batch()), 0);
        ck(ntt_inverse(context_.ntt_tables(), b, 0, stream_), "ntt_inverse");
        e.is_ntt_form() = false;
        throw_if_transparent(e);
    }

    // ---- plaintext operands (evaluator.cpp:1760-2287) and many-operand forms (242-261, 1649-1757)
    void Evaluator::check_valid(const Plaintext &p) const
    {
        // is_metadata_valid_for(Plaintext) + is_buffer_valid (valcheck.cpp:28-79, 209-219)
        bool ok = &p.context() == &context_;
        if (ok && p.is_ntt_form())
        {
            const Level &l = *p.level();
            ok = l.chain_index <= context_.first_level().chain_index && p.coeff_count() == (size_t)l.K * context_.n();
        }
        else if (ok)
            ok = p.coeff_count() <= context_.n();
        if (ok && context_.scheme() == Scheme::ckks)
            ok = std::isnormal(p.scale()) && p.scale() > 0;
        ok = ok && (p.coeff_count() == 0 || p.data() != nullptr);
        if (!ok)
            throw std::invalid_argument("plain is not valid for encryption parameters");
    }

    // coefficients modulo t -> [K][N] residues of the centred lift (evaluator.cpp:2098-2125, 2243-2282), times scale_by mod t
    void Evaluator::plain_to_rns(const Plaintext &plain, const Level &lvl, uint64_t scale_by, uint64_t *out) const
    {
        StreamScope pool_scope(stream_); // pool traffic of this call is ordered on the evaluator's stream whoever calls (pool.h)
        if (context_.scheme() == Scheme::ckks)
            throw std::invalid_argument("CKKS plain must be in NTT form");
        ck(k_plain_lift(context_.dev_mods(), host::make_mod(context_.plain_modulus()), plain.data(), plain.coeff_count(), scale_by,
                        lvl.dev.plain_upper_half_threshold, lvl.dev.upper_half_inc, out, (unsigned)context_.log_n(), lvl.K, stream_),
           "plain lift");
    }

    void Evaluator::transform_to_ntt_inplace(Plaintext &plain, const uint64_t *parms_id) const
    {
        StreamScope pool_scope(stream_); // pool traffic of this call is ordered on the evaluator's stream whoever calls (pool.h)
        check_valid(plain);
        const Level *lvl = context_.level_by_parms_id(parms_id);
        if (!lvl)
            throw std::invalid_argument("parms_id is not valid for the current context");
        if (plain.is_ntt_form())
            throw std::invalid_argument("plain is already in NTT form");
        const size_t words = (size_t)lvl->K * context_.n();
        uint64_t *out = DevicePool::global().alloc_words(words);
        try
        {
            plain_to_rns(plain, *lvl, 1, out);
            ck(ntt_forward(context_.ntt_tables(), plain_batch(out, words, lvl->K, 1, 0), 0, stream_), "plain ntt");
        }
        catch (...)
        {
            DevicePool::global().free_words(out);
            throw;
        }
        plain.adopt(out, words, words);
        plain.set_level(lvl);
    }

    void Evaluator::mod_switch_to_next_inplace(Plaintext &plain) const
    {
        StreamScope pool_scope(stream_); // pool traffic of this call is ordered on the evaluator's stream whoever calls (pool.h)
        // mod_switch_drop_to_next(Plaintext) (evaluator.cpp:1369-1402): the flat [K][N] array keeps its first K-1 components
        check_valid(plain);
        if (!plain.is_ntt_form())
            throw std::invalid_argument("plain is not in NTT form");
        const Level *next = context_.next_level(*plain.level());
        if (!next)
            throw std::invalid_argument("end of modulus switching chain reached");
        if (!scale_within_bounds(plain.scale(), *next))
            throw std::invalid_argument("scale out of bounds");
        plain.adopt_count((size_t)next->K * context_.n());
        plain.set_level(next);
    }
    void Evaluator::mod_switch_to_inplace(Plaintext &plain, const uint64_t *parms_id) const
    {
        StreamScope pool_scope(stream_); // pool traffic of this call is ordered on the evaluator's stream whoever calls (pool.h)
        check_valid(plain);
        const Level *target = context_.level_by_parms_id(parms_id);
        if (!plain.is_ntt_form())
            throw std::invalid_argument("plain is not in NTT form");
        if (!target)
            throw std::invalid_argument("parms_id is not valid for encryption parameters");
        if (plain.level()->chain_index < target->chain_index)
            throw std::invalid_argument("cannot switch to higher level modulus");
        while (plain.level() != target)
            mod_switch_to_next_inplace(plain);
    }

    void Evaluator::add_plain_inplace(Ciphertext &e, const Plaintext &plain) const
    {
        StreamScope pool_scope(stream_); // pool traffic of this call is ordered on the evaluator's stream whoever calls (pool.h)
        // add_plain_inplace / sub_plain_inplace share everything but the sign
        check_valid(e, "encrypted");
        check_valid(plain);
        addsub_plain(e, plain, 0);
    }
    void Evaluator::sub_plain_inplace(Ciphertext &e, const Plaintext &plain) const
    {
        StreamScope pool_scope(stream_); // pool traffic of this call is ordered on the evaluator's stream whoever calls (pool.h)
        check_valid(e, "encrypted");
        check_valid(plain);
        addsub_plain(e, plain, 1);
    }
    void Evaluator::addsub_plain(Ciphertext &e, const Plaintext &plain, int op) const
    {
        StreamScope pool_scope(stream_); // pool traffic of this call is ordered on the evaluator's stream whoever calls (pool.h)
        const Scheme scheme = context_.scheme();
        if (scheme == Scheme::bfv)
        {
            if (e.is_ntt_form())
                throw std::invalid_argument("BFV encrypted cannot be in NTT form");
            if (plain.is_ntt_form())
                throw std::invalid_argument("BFV plain cannot be in NTT form");
        }
        else if (scheme == Scheme::ckks)
        {
            if (!e.is_ntt_form())
                throw std::invalid_argument("CKKS encrypted must be in NTT form");
            if (!plain.is_ntt_form())
                throw std::invalid_argument("CKKS plain must be in NTT form");
            if (e.level() != plain.level())
                throw std::invalid_argument("encrypted and plain parameter mismatch");
            if (!are_close(e.scale(), plain.scale()))
                throw std::invalid_argument("scale mismatch");
        }
        else
        {
            if (!e.is_ntt_form())
                throw std::invalid_argument("BGV encrypted must be in NTT form");
            if (plain.is_ntt_form())
                throw std::invalid_argument("BGV plain cannot be in NTT form");
        }
        if (e.size() < 1)
            throw std::invalid_argument("encrypted is not valid for encryption parameters");
        const Level &lvl = *e.level();
        const unsigned n_log = (unsigned)context_.log_n();
        const ModDesc *mods = context_.dev_mods();
        switch (scheme)
        {
        case Scheme::bfv:
            // multiply_add/sub_plain_with_scaling_variant (util/scalingvariant.cpp:70-175)
            ck(k_bfv_addsub_plain(mods, host::make_mod(context_.plain_modulus()), plain.data(), plain.coeff_count(), lvl.dev.q_mod_t,
                                  lvl.dev.plain_upper_half_threshold, lvl.dev.delta_mod_q, e.plane(0), op, n_log, lvl.K, e.batch(), stream_),
               "bfv add/sub plain");
            break;
        case Scheme::ckks:
            ck(k_addsub_plain(mods, e.plane(0), plain.data(), op, n_log, lvl.K, e.batch(), stream_), "ckks add/sub plain");
            break;
        case Scheme::bgv:
        {
            // plain * correction_factor mod t, lifted and transformed at the ciphertext's level (evaluator.cpp:1836-1847)
            Scratch tmp((size_t)lvl.K * context_.n());
            plain_to_rns(plain, lvl, e.correction_factor(), tmp.p);
            ck(ntt_forward(context_.ntt_tables(), plain_batch(tmp.p, (size_t)lvl.K * context_.n(), lvl.K, 1, 0), 0, stream_), "plain ntt");
            ck(k_addsub_plain(mods, e.plane(0), tmp.p, op, n_log, lvl.K, e.batch(), stream_), "bgv add/sub plain");
            break;
        }
        default:
            throw std::invalid_argument("unsupported scheme");
        }
        throw_if_transparent(e);
    }

    void Evaluator::multiply_plain_ntt(Ciphertext &e, const uint64_t *plain_rns, const Level *plain_level, double plain_scale) const
    {
        StreamScope pool_scope(stream_); // pool traffic of this call is ordered on the evaluator's stream whoever calls (pool.h)
        // multiply_plain_ntt (evaluator.cpp:2157-2194)
        if (e.level() != plain_level)
            throw std::invalid_argument("encrypted_ntt and plain_ntt parameter mismatch");
        const Level &lvl = *e.level();
        ck(k_dyadic_plain(context_.dev_mods(), e.data(), plain_rns, e.data(), (unsigned)context_.log_n(), lvl.K, e.size() * e.batch(), stream_),
           "multiply_plain");
        e.scale() *= plain_scale;
        if (!scale_within_bounds(e.scale(), lvl))
            throw std::invalid_argument("scale out of bounds");
    }

    // The monomial shortcut of multiply_plain_normal (evaluator.cpp:2051-2095).  It is not merely faster: with the
    // "fast plain lift" (t below every q_i) the reference multiplies by the RAW coefficient even when it lies in the
    // upper half, i.e. by m instead of m - t; the ciphertext words differ from the generic path (by t * ct * x^e), so
    // the branch has to be reproduced.  Costs one 24-byte read-back per coefficient-form multiply_plain.
    bool Evaluator::mul_plain_monomial(Ciphertext &e, const Plaintext &plain) const
    {
        StreamScope pool_scope(stream_); // pool traffic of this call is ordered on the evaluator's stream whoever calls (pool.h)
        const Level &lvl = *e.level();
        Scratch stats(3);
        ck(k_plain_stats(plain.data(), plain.coeff_count(), stats.p, stream_), "plain stats");
        uint64_t st[3];
        ck(hipMemcpyAsync(st, stats.p, 24, hipMemcpyDeviceToHost, stream_), "plain stats read");
        ck(hipStreamSynchronize(stream_), "plain stats sync");
        if (st[0] != 1)
            return false;
        const size_t mono_exponent = (size_t)st[1] - 1;
        const uint64_t c = st[2], t = context_.plain_modulus();
        const std::vector<uint64_t> &q = context_.coeff_modulus();
        bool fast_lift = true; // qualifiers.using_fast_plain_lift: t smaller than every prime of this level
        for (unsigned i = 0; i < lvl.K; i++)
            fast_lift = fast_lift && t < q[i];
        std::vector<uint64_t> sc(lvl.K);
        for (unsigned i = 0; i < lvl.K; i++)
        {
            sc[i] = c % q[i];
            if (c >= lvl.dev.plain_upper_half_threshold && !fast_lift)
                sc[i] = (sc[i] + (q[i] - t % q[i]) % q[i]) % q[i]; // (c + Q - t) mod q_i
        }
        Scratch dsc(lvl.K);
        ck(hipMemcpyAsync(dsc.p, sc.data(), lvl.K * 8, hipMemcpyHostToDevice, stream_), "mono scalars");
        const size_t words = e.word_count();
        uint64_t *out = DevicePool::global().alloc_words(words);
        ck(k_negacyclic_mul_mono(context_.dev_mods(), e.data(), out, dsc.p, mono_exponent, (unsigned)context_.log_n(), lvl.K,
                                 e.size() * e.batch(), stream_),
           "multiply_plain monomial");
        ck(hipStreamSynchronize(stream_), "mono sync"); // sc lives on this stack frame
        const size_t size = e.size();
        e.adopt(&lvl, size, out, words);
        if (context_.scheme() == Scheme::ckks)
        {
            e.scale() *= plain.scale();
            if (!scale_within_bounds(e.scale(), lvl))
                throw std::invalid_argument("scale out of bounds");
        }
        return true;
    }

    void Evaluator::multiply_plain_inplace(Ciphertext &e, const Plaintext &plain) const
    {
        StreamScope pool_scope(stream_); // pool traffic of this call is ordered on the evaluator's stream whoever calls (pool.h)
        check_valid(e, "encrypted");
        check_valid(plain);
        const Level &lvl = *e.level();
        const size_t pw = (size_t)lvl.K * context_.n();
        if (e.is_ntt_form() && plain.is_ntt_form())
            multiply_plain_ntt(e, plain.data(), plain.level(), plain.scale());
        else if (!plain.is_ntt_form())
        {
            // multiply_plain_normal (evaluator.cpp:2021-2155) and the "encrypted in NTT form, plain not" branch (2006-2011):
            // lift the plaintext at the ciphertext's level, transform it, and multiply in the NTT domain.  The reference's
            // monomial shortcut computes the same product, so the canonical result is identical.
            const bool ct_ntt = e.is_ntt_form();
            const unsigned items = (unsigned)(e.size() * e.batch());
            if (!ct_ntt && mul_plain_monomial(e, plain))
            {
                throw_if_transparent(e);
                return;
            }
            Scratch tmp(pw);
            plain_to_rns(plain, lvl, 1, tmp.p);
            ck(ntt_forward(context_.ntt_tables(), plain_batch(tmp.p, pw, lvl.K, 1, 0), 0, stream_), "plain ntt");
            if (!ct_ntt)
                ck(ntt_forward(context_.ntt_tables(), plain_batch(e.data(), pw, lvl.K, items, 0), 1, stream_), "multiply_plain ntt");
            if (ct_ntt)
                multiply_plain_ntt(e, tmp.p, &lvl, plain.scale());
            else
            {
                ck(k_dyadic_plain(context_.dev_mods(), e.data(), tmp.p, e.data(), (unsigned)context_.log_n(), lvl.K, items, stream_),
                   "multiply_plain");
                ck(ntt_inverse(context_.ntt_tables(), plain_batch(e.data(), pw, lvl.K, items, 0), 0, stream_), "multiply_plain intt");
                if (context_.scheme() == Scheme::ckks)
                {
                    e.scale() *= plain.scale();
                    if (!scale_within_bounds(e.scale(), lvl))
                        throw std::invalid_argument("scale out of bounds");
                }
            }
        }
        else
        {
            // encrypted not in NTT form, plain in NTT form (evaluator.cpp:2012-2017)
            transform_to_ntt_inplace(e);
            multiply_plain_ntt(e, plain.data(), plain.level(), plain.scale());
            transform_from_ntt_inplace(e);
        }
        throw_if_transparent(e);
    }

    void Evaluator::add_many(const std::vector<const Ciphertext *> &encrypteds, Ciphertext &destination) const
    {
        StreamScope pool_scope(stream_); // pool traffic of this call is ordered on the evaluator's stream whoever calls (pool.h)
        if (encrypteds.empty())
            throw std::invalid_argument("encrypteds cannot be empty");
        for (const Ciphertext *c : encrypteds)
            if (!c || c == &destination)
                throw std::invalid_argument("encrypteds must be different from destination");
        destination = *encrypteds[0];
        for (size_t i = 1; i < encrypteds.size(); i++)
            add_inplace(destination, *encrypteds[i]);
    }

    void Evaluator::multiply_many(const std::vector<const Ciphertext *> &encrypteds, const KSwitchKeys &relin_keys, Ciphertext &destination) const
    {
        StreamScope pool_scope(stream_); // pool traffic of this call is ordered on the evaluator's stream whoever calls (pool.h)
        // the product tree of evaluator.cpp:1649-1723, every product relinearized
        if (encrypteds.empty())
            throw std::invalid_argument("encrypteds vector must not be empty");
        for (const Ciphertext *c : encrypteds)
            if (!c || c == &destination)
                throw std::invalid_argument("encrypteds must be different from destination");
        if (&encrypteds[0]->context() != &context_ || !encrypteds[0]->level())
            throw std::invalid_argument("encrypteds is not valid for encryption parameters");
        if (context_.scheme() != Scheme::bfv && context_.scheme() != Scheme::bgv)
            throw std::logic_error("unsupported scheme");
        if (encrypteds.size() == 1)
        {
            destination = *encrypteds[0];
            return;
        }
        std::vector<std::unique_ptr<Ciphertext>> owned;
        std::vector<const Ciphertext *> prod;
        for (size_t i = 0; i + 1 < encrypteds.size(); i += 2)
        {
            std::unique_ptr<Ciphertext> temp(new Ciphertext(context_, encrypteds[i]->batch()));
            if (encrypteds[i]->data() == encrypteds[i + 1]->data())
            {
                *temp = *encrypteds[i];
                square_inplace(*temp);
            }
            else
                multiply(*encrypteds[i], *encrypteds[i + 1], *temp);
            relinearize_inplace(*temp, relin_keys);
            prod.push_back(temp.get());
            owned.push_back(std::move(temp));
        }
        if (encrypteds.size() & 1)
            prod.push_back(encrypteds.back());
        for (size_t i = 0; i + 1 < prod.size(); i += 2)
        {
            std::unique_ptr<Ciphertext> temp(new Ciphertext(context_, prod[i]->batch()));
            multiply(*prod[i], *prod[i + 1], *temp);
            relinearize_inplace(*temp, relin_keys);
            prod.push_back(temp.get());
            owned.push_back(std::move(temp));
        }
        destination = *prod.back();
    }

    void Evaluator::exponentiate_inplace(Ciphertext &e, uint64_t exponent, const KSwitchKeys &relin_keys) const
    {
        StreamScope pool_scope(stream_); // pool traffic of this call is ordered on the evaluator's stream whoever calls (pool.h)
        if (&e.context() != &context_ || !e.level())
            throw std::invalid_argument("encrypted is not valid for encryption parameters");
        if (relin_keys.context() != &context_)
            throw std::invalid_argument("relin_keys is not valid for encryption parameters");
        if (exponent == 0)
            throw std::invalid_argument("exponent cannot be 0");
        if (exponent == 1)
            return;
        // multiply_many over `exponent` copies (evaluator.cpp:1725-1757); the copies share one buffer here, so the
        // first tree level squares it, exactly as the reference does for identical operands
        Ciphertext base(e);
        std::vector<const Ciphertext *> v((size_t)exponent, &base);
        multiply_many(v, relin_keys, e);
    }

    // ---- multiply (evaluator.cpp:352-708)
    void Evaluator::multiply_inplace(Ciphertext &e1, const Ciphertext &e2) const
    {
        StreamScope pool_scope(stream_); // pool traffic of this call is ordered on the evaluator's stream whoever calls (pool.h)
        check_valid(e1, "encrypted1");
        check_valid(e2, "encrypted2");
        if (e1.level() != e2.level())
            throw std::invalid_argument("encrypted1 and encrypted2 parameter mismatch");
        if (e1.batch() != e2.batch())
            throw std::invalid_argument("batch mismatch");
        switch (context_.scheme())
        {
        case Scheme::bfv:
            bfv_multiply(e1, e2);
            break;
        case Scheme::ckks:
            ckks_multiply(e1, e2);
            break;
        case Scheme::bgv:
            bgv_multiply(e1, e2);
            break;
        default:
            throw std::invalid_argument("unsupported scheme");
        }
        throw_if_transparent(e1);
    }
    void Evaluator::multiply(const Ciphertext &e1, const Ciphertext &e2, Ciphertext &dest) const
    {
        StreamScope pool_scope(stream_); // pool traffic of this call is ordered on the evaluator's stream whoever calls (pool.h)
        // evaluator.h:239-247: destination = encrypted1; multiply_inplace(destination, encrypted2).
        // Device-resident fast path: the common size-2 x size-2 CKKS product writes the three result
        // polynomials straight into `dest` instead of copying encrypted1 first.
        if (&dest == &e1)
            return multiply_inplace(dest, e2);
        if (&dest == &e2)
            return multiply_inplace(dest, e1);
        if (context_.scheme() == Scheme::ckks && e1.size() == 2 && e2.size() == 2 && &dest.context() == &context_ &&
            dest.batch() == e1.batch())
        {
            check_valid(e1, "encrypted1");
            check_valid(e2, "encrypted2");
            if (e1.level() != e2.level())
                throw std::invalid_argument("encrypted1 and encrypted2 parameter mismatch");
            if (e1.batch() != e2.batch())
                throw std::invalid_argument("batch mismatch");
            if (!(e1.is_ntt_form() && e2.is_ntt_form()))
                throw std::invalid_argument("encrypted1 or encrypted2 must be in NTT form");
            const Level &lvl = *e1.level();
            PlaneGeom g{ (unsigned)context_.log_n(), lvl.K, (unsigned)e1.batch() };
            // operands first: whatever is pending on them (a key-switch tail, a product of their own) is settled before their words are
            // read - now or by the deferred product
            const uint64_t *xw = e1.data(), *yw = e2.data();
            dest.reshape_uninitialized(&lvl, 3);
            // Round 6: batches whose key switch runs un-split at a two-pass size - the product is not formed now (LazyProduct): a
            // relinearize_inplace that follows forms it inside its own kernels, anything else that needs the words forms it first
            const bool defer = may_defer_product(lvl, e1.batch());
            // (development builds, bound only: SEALHIP_AB_SKIP_TENSOR leaves the product unwritten after two real calls - what fusing the
            // tensor product into its consumers could save at most, profiles/r06_lazy_product.txt)
            static const bool skip_tensor = shl_ab_getenv("SEALHIP_AB_SKIP_TENSOR") != nullptr;
            static std::atomic<unsigned> tensor_calls{ 0 };
            if (defer)
                defer_product(dest, &e1, &e2, nullptr);
            else if (!skip_tensor || tensor_calls.fetch_add(1) < 2)
                ck(k_ckks_multiply_2x2(context_.dev_mods(), context_.ntt_tables().fpd, nullptr, xw, yw, dest.data(), g, stream_), "ckks_multiply");
            dest.is_ntt_form() = true;
            dest.correction_factor() = 1;
            dest.scale() = e1.scale() * e2.scale();
            if (!scale_within_bounds(dest.scale(), lvl))
                throw std::invalid_argument("scale out of bounds");
            throw_if_transparent(dest);
            return;
        }
        if (context_.scheme() == Scheme::bfv && &dest.context() == &context_ && dest.batch() == e1.batch())
        {
            // BFV: the same checks as multiply_inplace, then the product straight into `dest` (no copy of encrypted1: 6.8 GB per
            // step at BASELINE configs[3])
            check_valid(e1, "encrypted1");
            check_valid(e2, "encrypted2");
            if (e1.level() != e2.level())
                throw std::invalid_argument("encrypted1 and encrypted2 parameter mismatch");
            if (e1.batch() != e2.batch())
                throw std::invalid_argument("batch mismatch");
            bfv_multiply_to(e1, e2, dest);
            dest.is_ntt_form() = false;
            dest.scale() = e1.scale();
            dest.correction_factor() = e1.correction_factor();
            throw_if_transparent(dest);
            return;
        }
        dest = e1;
        if (&e1 == &e2)
            multiply_inplace(dest, dest);
        else
            multiply_inplace(dest, e2);
    }

    void Evaluator::square_inplace(Ciphertext &e) const
    {
        StreamScope pool_scope(stream_); // pool traffic of this call is ordered on the evaluator's stream whoever calls (pool.h)
        // ckks_square / bfv_square compute (c0^2, 2 c0 c1, c1^2) = the product of e with itself
        // (evaluator.cpp:878-1142); canonical results coincide with multiply(e, e).
        check_valid(e, "encrypted");
        multiply_inplace(e, e);
    }

    // (c0, c1) x (d0, d1) -> three polynomials in e1, NTT form (evaluator.cpp:626-707).  When e1's slab has room the product is
    // formed in place (every thread reads its four words before it writes its three); otherwise it goes straight into a new
    // slab that e1 adopts - no copy of the two old polynomials, no zeroing of the third.
    void Evaluator::tensor_2x2(Ciphertext &e1, const Ciphertext &e2, const Level &lvl, const PlaneGeom &g) const
    {
        StreamScope pool_scope(stream_); // pool traffic of this call is ordered on the evaluator's stream whoever calls (pool.h)
        const size_t words = 3 * g.words();
        const uint64_t *x = e1.data(), *y = (&e1 == &e2) ? x : e2.data(); // (completes a pending key-switch tail)
        if (e1.capacity_words() >= words)
        {
            e1.reshape_uninitialized(&lvl, 3); // enough room: the words stay where they are
            ck(k_ckks_multiply_2x2(context_.dev_mods(), context_.ntt_tables().fpd, nullptr, x, y, e1.data(), g, stream_), "multiply 2x2");
        }
        else
        {
            uint64_t *out = DevicePool::global().alloc_words(words, stream_);
            ck(k_ckks_multiply_2x2(context_.dev_mods(), context_.ntt_tables().fpd, nullptr, x, y, out, g, stream_), "multiply 2x2");
            e1.adopt(&lvl, 3, out, words);
        }
    }

    void Evaluator::ckks_multiply(Ciphertext &e1, const Ciphertext &e2) const
    {
        StreamScope pool_scope(stream_); // pool traffic of this call is ordered on the evaluator's stream whoever calls (pool.h)
        if (!(e1.is_ntt_form() && e2.is_ntt_form()))
            throw std::invalid_argument("encrypted1 or encrypted2 must be in NTT form");
        const Level &lvl = *e1.level();
        size_t s1 = e1.size(), s2 = e2.size();
        if (s1 < 2 || s2 < 2)
            throw std::invalid_argument("encrypted size must be at least 2");
        size_t dest = s1 + s2 - 1;
        if (dest > 16)
            throw std::logic_error("invalid parameters");
        const bool self = (&e1 == &e2);
        double new_scale = e1.scale() * e2.scale();
        PlaneGeom g{ (unsigned)context_.log_n(), lvl.K, (unsigned)e1.batch() };
        if (dest == 3 && may_defer_product(lvl, e1.batch()))
        {
            // Round 6 (LazyProduct, in place): e1's slab - its two polynomials as they are now, whatever was pending on them completed -
            // goes to the record, e1 gets a fresh slab of three whose words are pending.  Nothing is copied and nothing is computed here.
            (void)e1.data();
            if (!self)
                (void)e2.data();
            const size_t words = 3 * g.words();
            uint64_t *fresh = DevicePool::global().alloc_words(words, stream_);
            uint64_t *old = e1.exchange_slab(&lvl, 3, fresh, words);
            defer_product(e1, nullptr, self ? nullptr : &e2, old);
        }
        else if (dest == 3)
            tensor_2x2(e1, self ? e1 : e2, lvl, g);
        else
        {
            size_t words = dest * g.words();
            uint64_t *out = DevicePool::global().alloc_words(words);
            ck(k_multiply_general(context_.dev_mods(), nullptr, e1.data(), (unsigned)s1, e2.data(), (unsigned)s2, out, g, stream_),
               "ckks_multiply general");
            e1.adopt(&lvl, dest, out, words);
        }
        e1.scale() = new_scale;
        if (!scale_within_bounds(e1.scale(), lvl))
            throw std::invalid_argument("scale out of bounds");
    }

    // bgv_multiply / bgv_square (evaluator.cpp:710-841, 1079-1142): the tensor product of ckks_multiply on
    // NTT-form operands; the scale is untouched and the correction factors multiply modulo t.
    void Evaluator::bgv_multiply(Ciphertext &e1, const Ciphertext &e2) const
    {
        StreamScope pool_scope(stream_); // pool traffic of this call is ordered on the evaluator's stream whoever calls (pool.h)
        if (!(e1.is_ntt_form() && e2.is_ntt_form()))
            throw std::invalid_argument("encrypted1 or encrypted2 must be in NTT form");
        const Level &lvl = *e1.level();
        size_t s1 = e1.size(), s2 = e2.size();
        if (s1 < 2 || s2 < 2)
            throw std::invalid_argument("encrypted size must be at least 2");
        size_t dest = s1 + s2 - 1;
        if (dest > 16)
            throw std::logic_error("invalid parameters");
        const bool self = (&e1 == &e2);
        const uint64_t cf = host::mulmod(e1.correction_factor(), e2.correction_factor(), context_.plain_modulus());
        PlaneGeom g{ (unsigned)context_.log_n(), lvl.K, (unsigned)e1.batch() };
        if (dest == 3)
            tensor_2x2(e1, self ? e1 : e2, lvl, g);
        else
        {
            size_t words = dest * g.words();
            uint64_t *out = DevicePool::global().alloc_words(words);
            ck(k_multiply_general(context_.dev_mods(), nullptr, e1.data(), (unsigned)s1, e2.data(), (unsigned)s2, out, g, stream_),
               "bgv_multiply general");
            e1.adopt(&lvl, dest, out, words);
        }
        e1.correction_factor() = cf;
    }

    void Evaluator::bfv_multiply(Ciphertext &e1, const Ciphertext &e2) const
    {
        bfv_multiply_to(e1, e2, e1);
    }
    // the product of e1 and e2 into dest (dest may be e1): BEHZ reads its operands through the first transform pass and builds the
    // result in a new slab, so an out-of-place product needs no copy of e1 (evaluator.h:239-247 copies it first)
    void Evaluator::bfv_multiply_to(const Ciphertext &e1, const Ciphertext &e2, Ciphertext &dst) const
    {
        StreamScope pool_scope(stream_); // pool traffic of this call is ordered on the evaluator's stream whoever calls (pool.h)
        if (e1.is_ntt_form() || e2.is_ntt_form())
            throw std::invalid_argument("encrypted1 or encrypted2 cannot be in NTT form");
        const Level &lvl = *e1.level();
        const LevelDev &lv = lvl.dev;
        const unsigned K = lvl.K, nBsk = lv.nBsk;
        const size_t N = context_.n(), B = e1.batch();
        const unsigned n_log = (unsigned)context_.log_n();
        size_t s1 = e1.size(), s2 = e2.size();
        if (s1 < 2 || s2 < 2)
            throw std::invalid_argument("encrypted size must be at least 2");
        size_t dest = s1 + s2 - 1;
        if (dest > 16)
            throw std::logic_error("invalid parameters");
        const bool self = (&e1 == &e2);
        const NttTables &tb = context_.ntt_tables();
        const ModDesc *mods = context_.dev_mods();

        // steps (1)-(3): lift each input to q U Bsk and transform (evaluator.cpp:456-489)
        auto lift = [&](const Ciphertext &x, Scratch &xq, Scratch &xb) {
            size_t items = x.size() * B;
            // out of place: the first pass reads the ciphertext itself (NttBatch::src, mode 0) - no copy of the input
            NttBatch qb = plain_batch(xq.p, (size_t)K * N, K, (unsigned)items, 0);
            qb.src = x.data();
            qb.src_outer_stride = (size_t)K * N;
            qb.src_ncomp = K;
            qb.src_mode = 0;
            ck(ntt_forward(tb, qb, 0, stream_), "bfv ntt q");
            ck(k_behz_lift(mods, lv, x.data(), xb.p, n_log, items, stream_), "behz lift");
            NttBatch bb = plain_batch(xb.p, (size_t)nBsk * N, nBsk, (unsigned)items, 0);
            bb.comp_prime = lv.bsk_prime;
            bb.cls_hint = 0; // the auxiliary base is 61-bit primes: integer back end, single-class kernels (ntt_kernels.h)
            ck(ntt_forward(tb, bb, 0, stream_), "bfv ntt Bsk");
        };
        Scratch x_q(s1 * B * K * N), x_b(s1 * B * nBsk * N);
        lift(e1, x_q, x_b);
        std::unique_ptr<Scratch> y_q, y_b;
        if (!self)
        {
            y_q.reset(new Scratch(s2 * B * K * N));
            y_b.reset(new Scratch(s2 * B * nBsk * N));
            lift(e2, *y_q, *y_b);
        }
        const uint64_t *yq = self ? x_q.p : y_q->p;
        const uint64_t *yb = self ? x_b.p : y_b->p;

        // step (4): dyadic ciphertext product in both bases (evaluator.cpp:497-541)
        Scratch d_q(dest * B * K * N), d_b(dest * B * nBsk * N);
        PlaneGeom gq{ n_log, K, (unsigned)B }, gb{ n_log, nBsk, (unsigned)B };
        // Two-pass sizes, 2 x 2: steps (4) and (5) are ONE pair of passes per base - the inverse transform's first pass forms the
        // product from the four operand polynomials as it loads them (NttBatch::prod_x), so the product is never stored in NTT form
        // and read back: 7 plane crossings instead of 13 for the pair.  SEALHIP_BFV_NO_PRODUCT_SOURCE=1 (development builds): separate
        static const bool product_source_ok = !shl_ab_getenv("SEALHIP_BFV_NO_PRODUCT_SOURCE");
        const bool product_source = product_source_ok && s1 == 2 && s2 == 2 && ntt2_supports(context_.log_n());
        if (product_source)
        {
            NttBatch iq = plain_batch(d_q.p, (size_t)K * N, K, (unsigned)(3 * B), 0);
            iq.prod_x = x_q.p;
            iq.prod_y = yq;
            iq.prod_batch = (unsigned)B;
            iq.src_outer_stride = (size_t)K * N;
            ck(ntt_inverse(tb, iq, 0, stream_), "bfv tensor + intt q");
            NttBatch ibp = plain_batch(d_b.p, (size_t)nBsk * N, nBsk, (unsigned)(3 * B), 0);
            ibp.comp_prime = lv.bsk_prime;
            ibp.cls_hint = 0;
            ibp.prod_x = x_b.p;
            ibp.prod_y = yb;
            ibp.prod_batch = (unsigned)B;
            ibp.src_outer_stride = (size_t)nBsk * N;
            ck(ntt_inverse(tb, ibp, 0, stream_), "bfv tensor + intt Bsk");
        }
        else if (s1 == 2 && s2 == 2)
        {
            // the common product: four loads, three reductions per coefficient (the general kernel: eight and four); the transforms
            // around it leave canonical words, primes below 2^50 take the double-precision products, the 61-bit auxiliary base does not
            ck(k_ckks_multiply_2x2(mods, tb.fpd, nullptr, x_q.p, yq, d_q.p, gq, stream_), "bfv tensor q");
            ck(k_ckks_multiply_2x2(mods, tb.fpd, lv.bsk_prime, x_b.p, yb, d_b.p, gb, stream_), "bfv tensor Bsk");
        }
        else
        {
            ck(k_multiply_general(mods, nullptr, x_q.p, (unsigned)s1, yq, (unsigned)s2, d_q.p, gq, stream_), "bfv tensor q");
            ck(k_multiply_general(mods, lv.bsk_prime, x_b.p, (unsigned)s1, yb, (unsigned)s2, d_b.p, gb, stream_), "bfv tensor Bsk");
        }

        // step (5): back to coefficient form
        if (!product_source)
        {
            ck(ntt_inverse(tb, plain_batch(d_q.p, (size_t)K * N, K, (unsigned)(dest * B), 0), 0, stream_), "bfv intt q");
            NttBatch ib = plain_batch(d_b.p, (size_t)nBsk * N, nBsk, (unsigned)(dest * B), 0);
            ib.comp_prime = lv.bsk_prime;
            ib.cls_hint = 0;
            ck(ntt_inverse(tb, ib, 0, stream_), "bfv intt Bsk");
        }

        // steps (6)-(8)
        size_t words = dest * B * K * N;
        uint64_t *out = DevicePool::global().alloc_words(words);
        ck(k_behz_floor_sk(mods, lv, d_q.p, d_b.p, out, n_log, dest * B, stream_), "behz floor_sk");
        dst.adopt(&lvl, dest, out, words);
    }

} // namespace sealhip
