#include "evaluator.h"
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <limits>

namespace sealhip
{
    namespace
    {
        void ck(hipError_t e, const char *what)
        {
            if (e != hipSuccess)
                throw std::runtime_error(std::string("HIP failure in ") + what + ": " + hipGetErrorString(e));
        }

        // util::are_close<double> (util/common.h:574-578)
        bool are_close(double a, double b)
        {
            double scale_factor = std::max({ std::fabs(a), std::fabs(b), 1.0 });
            return std::fabs(a - b) < std::numeric_limits<double>::epsilon() * scale_factor;
        }

        // util::naf (util/numth.h:22-42)
        std::vector<int> naf(int value)
        {
            std::vector<int> res;
            bool sign = value < 0;
            value = std::abs(value);
            for (int i = 0; value; i++)
            {
                int zi = (value & 1) ? 2 - (value & 3) : 0;
                value = (value - zi) >> 1;
                if (zi)
                    res.push_back((sign ? -zi : zi) * (1 << i));
            }
            return res;
        }

        // balance_correction_factors (evaluator.cpp:50-117).  Two BGV operands carry correction factors c1, c2 (units mod t);
        // before they can be added both are scaled to a common factor f = e1*c1 = e2*c2 (mod t), and the scalars should be
        // small as centred residues because they multiply the noise.  With rho = c2 / c1 (mod t) the admissible pairs are
        // exactly the lattice points e1 = rho * e2 (mod t), and the short ones appear among the remainders of Euclid's
        // algorithm on (t, rho): every step yields r = s * rho (mod t).  The walk starts from (rho, 1) and a later step
        // replaces the choice only when its centred 1-norm is STRICTLY smaller and r is a unit - the reference's tie rule,
        // which decides the result words and is therefore kept.
        void balance_correction_factors(uint64_t factor1, uint64_t factor2, uint64_t t, uint64_t &f, uint64_t &e1, uint64_t &e2)
        {
            const uint64_t c1 = factor1 % t, c2 = factor2 % t;
            if (c1 == 0 || std::__gcd(c1, t) != 1)
                throw std::logic_error("invalid correction factor1");
            const uint64_t rho = host::mulmod(host::invmod(c1, t), c2, t);
            const auto residue = [t](int64_t v) { // v mod t in [0, t)
                const uint64_t m = static_cast<uint64_t>(v < 0 ? -v : v) % t;
                return (v < 0 && m) ? t - m : m;
            };
            const auto centred_abs = [t](uint64_t x) { // |x| as the centred representative of x mod t
                return static_cast<int64_t>(x > t / 2 ? t - x : x);
            };
            struct Row
            {
                int64_t r, s; // r = s * rho (mod t)
            };
            Row above{ static_cast<int64_t>(t), 0 }, here{ static_cast<int64_t>(rho), 1 };
            e1 = rho;
            e2 = 1;
            int64_t best = centred_abs(e1) + centred_abs(e2);
            while (here.r != 0)
            {
                const int64_t quot = above.r / here.r;
                const Row below{ above.r - quot * here.r, above.s - quot * here.s };
                above = here;
                here = below;
                const uint64_t r = residue(here.r), sc = residue(here.s);
                if (r == 0 || std::__gcd(r, t) != 1)
                    continue;
                const int64_t norm = centred_abs(r) + centred_abs(sc);
                if (norm < best)
                {
                    best = norm;
                    e1 = r;
                    e2 = sc;
                }
            }
            f = host::mulmod(e1, c1, t);
        }

        NttBatch plain_batch(uint64_t *data, size_t outer_stride, unsigned ncomp, unsigned nouter, unsigned prime_first)
        {
            NttBatch b{};
            b.data = data;
            b.outer_stride = outer_stride;
            b.ncomp = ncomp;
            b.nouter = nouter;
            b.comp_prime = nullptr;
            b.prime_first = prime_first;
            b.src = nullptr;
            return b;
        }
    } // namespace

    // ---------------------------------------------------------------- Ciphertext
    Ciphertext::~Ciphertext()
    {
        release();
    }
    namespace
    {
        // settle() runs from const accessors, and the reference lets several threads read one ciphertext at a time (evaluator.h:
        // "concurrent calls on different destinations are safe"): exactly one of them may take the pending tail, the others wait
        // until it has been launched.  One mutex per ciphertext would grow every object; a small table keyed by address does.
        std::mutex g_settle_mu[64];
        inline std::mutex &settle_mutex(const void *p)
        {
            return g_settle_mu[(reinterpret_cast<uintptr_t>(p) >> 6) & 63];
        }
        thread_local const Ciphertext *tl_settling = nullptr; // the tail's own kernels read the words through data() / plane()
    } // namespace
    void Ciphertext::settle() const
    {
        if (!__atomic_load_n(&lazy_, __ATOMIC_ACQUIRE) || tl_settling == this)
            return;
        std::lock_guard<std::mutex> lock(settle_mutex(this));
        struct Marker
        {
            const Ciphertext *saved;
            explicit Marker(const Ciphertext *c) : saved(tl_settling) { tl_settling = c; }
            ~Marker() { tl_settling = saved; }
        } marker(this);
        LazyTail *pending = lazy_;
        if (!pending)
            return; // another thread completed it while this one waited
        const LazyTail t = *pending;
        // the words are valid once complete_tail() returns; only then may a reader that skips the lock see "nothing pending"
        try
        {
            t.owner->complete_tail(const_cast<Ciphertext &>(*this), t);
        }
        catch (...)
        {
            __atomic_store_n(&lazy_, (LazyTail *)nullptr, __ATOMIC_RELEASE);
            delete pending;
            throw;
        }
        __atomic_store_n(&lazy_, (LazyTail *)nullptr, __ATOMIC_RELEASE);
        delete pending;
    }
    void Ciphertext::drop_lazy()
    {
        if (!lazy_)
            return;
        const LazyTail t = *lazy_;
        delete lazy_;
        lazy_ = nullptr;
        t.owner->forget_tail(*this, t);
    }
    void Ciphertext::release()
    {
        drop_lazy();
        DevicePool::global().free_words(data_);
        data_ = nullptr;
        capacity_words_ = 0;
        size_ = 0;
        level_ = nullptr;
    }
    Ciphertext::Ciphertext(const Ciphertext &o) : ctx_(o.ctx_), batch_(o.batch_)
    {
        *this = o;
    }
    Ciphertext &Ciphertext::operator=(const Ciphertext &o)
    {
        if (this == &o)
            return *this;
        o.settle();  // the source's words are read below
        drop_lazy(); // this object's words are replaced
        if (ctx_ != o.ctx_ || batch_ != o.batch_)
        {
            release();
            ctx_ = o.ctx_;
            batch_ = o.batch_;
        }
        size_t words = o.word_count();
        if (capacity_words_ < words)
        {
            DevicePool::global().free_words(data_);
            data_ = DevicePool::global().alloc_words(words);
            capacity_words_ = words;
        }
        level_ = o.level_;
        size_ = o.size_;
        is_ntt_form_ = o.is_ntt_form_;
        scale_ = o.scale_;
        correction_factor_ = o.correction_factor_;
        if (words)
            ck(hipMemcpyAsync(data_, o.data_, words * 8, hipMemcpyDeviceToDevice, DevicePool::thread_stream()), "Ciphertext copy");
        return *this;
    }
    void Ciphertext::resize(const Level *level, size_t size, hipStream_t stream)
    {
        if (!level)
            throw std::invalid_argument("parms_id is not valid for encryption parameters");
        if ((size < 2 && size != 0) || size > 16) // SEAL_CIPHERTEXT_SIZE_MIN/MAX (defines.h)
            throw std::invalid_argument("invalid size");
        size_t pw = batch_ * level->K * ctx_->n();
        size_t need = size * pw;
        bool same_level = (level == level_);
        // a deferred key-switch tail works on the first two polynomials in place: dropping trailing ones at the same level
        // (relinearize: 3 -> 2) leaves it alone, anything else completes it first
        if (lazy_ && !(same_level && size >= 2 && size <= size_))
            settle();
        size_t keep = same_level ? std::min(size_, size) * pw : 0;
        if (need > capacity_words_)
        {
            uint64_t *nd = DevicePool::global().alloc_words(need);
            if (keep)
                ck(hipMemcpyAsync(nd, data_, keep * 8, hipMemcpyDeviceToDevice, stream), "Ciphertext resize copy");
            DevicePool::global().free_words(data_);
            data_ = nd;
            capacity_words_ = need;
        }
        if (need > keep)
            ck(hipMemsetAsync(data_ + keep, 0, (need - keep) * 8, stream), "Ciphertext resize zero");
        level_ = level;
        size_ = size;
    }
    void Ciphertext::reshape_uninitialized(const Level *level, size_t size)
    {
        drop_lazy();
        size_t need = size * batch_ * level->K * ctx_->n();
        if (need > capacity_words_)
        {
            DevicePool::global().free_words(data_);
            data_ = DevicePool::global().alloc_words(need);
            capacity_words_ = need;
        }
        level_ = level;
        size_ = size;
    }
    void Ciphertext::adopt(const Level *level, size_t size, uint64_t *slab, size_t capacity_words)
    {
        drop_lazy();
        DevicePool::global().free_words(data_);
        data_ = slab;
        capacity_words_ = capacity_words;
        level_ = level;
        size_ = size;
    }

    // ---------------------------------------------------------------- Plaintext
    Plaintext::~Plaintext()
    {
        DevicePool::global().free_words(data_);
    }
    Plaintext::Plaintext(const Plaintext &o) : ctx_(o.ctx_)
    {
        *this = o;
    }
    Plaintext &Plaintext::operator=(const Plaintext &o)
    {
        if (this == &o)
            return *this;
        ctx_ = o.ctx_;
        if (capacity_words_ < o.coeff_count_)
        {
            DevicePool::global().free_words(data_);
            data_ = DevicePool::global().alloc_words(o.coeff_count_);
            capacity_words_ = o.coeff_count_;
        }
        coeff_count_ = o.coeff_count_;
        level_ = o.level_;
        scale_ = o.scale_;
        if (coeff_count_)
            ck(hipMemcpyAsync(data_, o.data_, coeff_count_ * 8, hipMemcpyDeviceToDevice, DevicePool::thread_stream()), "Plaintext copy");
        return *this;
    }
    void Plaintext::resize(size_t coeff_count, hipStream_t stream)
    {
        if (level_)
            throw std::logic_error("cannot resize an NTT transformed Plaintext"); // plaintext.h:274-277
        if (coeff_count > capacity_words_)
        {
            uint64_t *nd = DevicePool::global().alloc_words(coeff_count);
            if (coeff_count_)
                ck(hipMemcpyAsync(nd, data_, coeff_count_ * 8, hipMemcpyDeviceToDevice, stream), "Plaintext resize copy");
            DevicePool::global().free_words(data_);
            data_ = nd;
            capacity_words_ = coeff_count;
        }
        if (coeff_count > coeff_count_)
            ck(hipMemsetAsync(data_ + coeff_count_, 0, (coeff_count - coeff_count_) * 8, stream), "Plaintext resize zero");
        coeff_count_ = coeff_count;
    }
    void Plaintext::set(const uint64_t *words, size_t count, bool from_device)
    {
        level_ = nullptr;
        coeff_count_ = 0;
        resize(count, nullptr);
        if (count && from_device)
            ck(hipMemcpy(data_, words, count * 8, hipMemcpyDeviceToDevice), "Plaintext set");
        else if (count)
        {
            ck(hipDeviceSynchronize(), "Plaintext set");
            copy_h2d(data_, words, count * 8);
        }
    }
    void Plaintext::adopt(uint64_t *slab, size_t count, size_t capacity_words)
    {
        DevicePool::global().free_words(data_);
        data_ = slab;
        coeff_count_ = count;
        capacity_words_ = capacity_words;
    }

    // ---------------------------------------------------------------- KSwitchKeys
    KSwitchKeys::~KSwitchKeys()
    {
        for (auto &k : keys_)
            if (k.dev)
                (void)hipFree(k.dev);
    }
    void KSwitchKeys::clear()
    {
        for (auto &k : keys_)
            if (k.dev)
                (void)hipFree(k.dev); // synchronises with the device: no queued key switch still reads it
        keys_.clear();
    }
    size_t KSwitchKeys::size() const
    {
        size_t c = 0;
        for (auto &k : keys_)
            c += k.dev != nullptr;
        return c;
    }
    void KSwitchKeys::set_key(const Context &ctx, size_t index, size_t digits, const uint64_t *words, bool from_device, size_t digit0)
    {
        if (!words)
            throw std::invalid_argument("empty key");
        const size_t bytes = digits * 2 * ctx.key_level().K * ctx.n() * 8;
        set_key_with(
            ctx, index, digits,
            [&](uint64_t *dst) {
                if (from_device)
                    ck(hipMemcpy(dst, words, bytes, hipMemcpyDeviceToDevice), "upload key");
                else
                    copy_h2d(dst, words, bytes);
            },
            digit0);
    }
    void KSwitchKeys::set_key_with(const Context &ctx, size_t index, size_t digits, const std::function<void(uint64_t *)> &upload, size_t digit0)
    {
        if (!ctx.using_keyswitching())
            throw std::logic_error("keyswitching is not supported by the context");
        if (ctx_ && ctx_ != &ctx)
            throw std::invalid_argument("kswitch_keys belongs to another context");
        if (digits == 0)
            throw std::invalid_argument("empty key");
        ctx_ = &ctx;
        if (index >= keys_.size())
            keys_.resize(index + 1);
        size_t L = ctx.key_level().K;
        size_t bytes = digits * 2 * L * ctx.n() * 8;
        if (keys_[index].dev)
            (void)hipFree(keys_[index].dev);
        void *p = nullptr;
        const bool reorder = ntt2_supports(ctx.log_n()) && !shl_ab_getenv("SEALHIP_OLD_KS");
        // register order carries a second plane: the Shoup quotients of the integer back end's components
        const size_t plane_words = bytes / 8;
        ck(hipMalloc(&p, reorder ? key_register_order_words(ctx.log_n(), (unsigned)L, digits * 2) * 8 : bytes), "hipMalloc key");
        if (reorder)
        {
            // upload to a staging block, then lay the key out for the fused kernel
            Scratch stage(bytes / 8);
            upload(stage.p);
            ck(key_to_register_order(ctx.ntt_tables(), stage.p, (uint64_t *)p, (unsigned)L, digits * 2, nullptr), "key layout");
            ck(hipDeviceSynchronize(), "key layout sync");
        }
        else
            upload((uint64_t *)p);
        keys_[index].dev = (uint64_t *)p;
        keys_[index].digits = digits;
        keys_[index].digit0 = digit0;
        keys_[index].register_order = reorder;
        keys_[index].quot_off = reorder ? plane_words : 0;
    }

    // ---------------------------------------------------------------- Evaluator
    Evaluator::Evaluator(const Context &context) : context_(context)
    {
        void *p = nullptr;
        ck(hipMalloc(&p, sizeof(unsigned)), "hipMalloc flag");
        d_flag_ = (unsigned *)p;
    }
    // ---- deferred key-switch tails (LazyTail, evaluator.h)
    namespace
    {
        std::atomic<uint64_t> g_tail_folded{ 0 }, g_tail_plain{ 0 }, g_tail_dropped{ 0 };
    }
    void lazy_tail_stats(uint64_t &folded, uint64_t &plain, uint64_t &dropped)
    {
        folded = g_tail_folded.load();
        plain = g_tail_plain.load();
        dropped = g_tail_dropped.load();
    }
    void Evaluator::defer_tail(Ciphertext &e, uint64_t *acc) const
    {
        e.lazy_ = new LazyTail{ this, acc };
        std::lock_guard<std::mutex> lock(lazy_mu_);
        lazy_cts_.push_back(&e);
    }
    LazyTail Evaluator::detach_tail(Ciphertext &e) const
    {
        const LazyTail t = *e.lazy_;
        delete e.lazy_;
        e.lazy_ = nullptr;
        std::lock_guard<std::mutex> lock(lazy_mu_);
        lazy_cts_.erase(std::remove(lazy_cts_.begin(), lazy_cts_.end(), &e), lazy_cts_.end());
        return t;
    }
    void Evaluator::forget_tail(const Ciphertext &e, LazyTail t) const
    {
        {
            std::lock_guard<std::mutex> lock(lazy_mu_);
            lazy_cts_.erase(std::remove(lazy_cts_.begin(), lazy_cts_.end(), &e), lazy_cts_.end());
        }
        DevicePool::global().free_words(t.acc, stream_);
        g_tail_dropped++;
    }
    void Evaluator::complete_tail(Ciphertext &e, LazyTail t) const
    {
        {
            std::lock_guard<std::mutex> lock(lazy_mu_);
            lazy_cts_.erase(std::remove(lazy_cts_.begin(), lazy_cts_.end(), &e), lazy_cts_.end());
        }
        // the sums were produced on this evaluator's stream: the tail runs there too; a caller working on another stream
        // (another evaluator, a host copy) continues only when it is done
        const hipStream_t caller = DevicePool::thread_stream();
        static const bool trace = shl_ab_getenv("SEALHIP_KS_TRACE") != nullptr; // tests: which tail ran
        if (trace)
            std::fprintf(stderr, "[ks] plain tail\n");
        g_tail_plain++;
        try
        {
            StreamScope scope(stream_);
            switch_key_finish(e, t.acc, 1);
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
        if (transparent_check_ && is_transparent(ct))
            throw std::logic_error("result ciphertext is transparent");
    }

    // ---- negate / add / sub (evaluator.cpp:130-350)
    void Evaluator::negate_inplace(Ciphertext &e) const
    {
        check_valid(e, "encrypted");
        PlaneGeom g{ (unsigned)context_.log_n(), e.level()->K, (unsigned)e.batch() };
        if (e.size())
            ck(k_addsub(context_.dev_mods(), e.data(), nullptr, e.data(), 2, g, (unsigned)e.size(), stream_), "negate");
        throw_if_transparent(e);
    }

    void Evaluator::add_inplace(Ciphertext &e1, const Ciphertext &e2) const
    {
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
        if (context_.scheme() == Scheme::ckks)
            throw std::invalid_argument("CKKS plain must be in NTT form");
        ck(k_plain_lift(context_.dev_mods(), host::make_mod(context_.plain_modulus()), plain.data(), plain.coeff_count(), scale_by,
                        lvl.dev.plain_upper_half_threshold, lvl.dev.upper_half_inc, out, (unsigned)context_.log_n(), lvl.K, stream_),
           "plain lift");
    }

    void Evaluator::transform_to_ntt_inplace(Plaintext &plain, const uint64_t *parms_id) const
    {
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
        // add_plain_inplace / sub_plain_inplace share everything but the sign
        check_valid(e, "encrypted");
        check_valid(plain);
        addsub_plain(e, plain, 0);
    }
    void Evaluator::sub_plain_inplace(Ciphertext &e, const Plaintext &plain) const
    {
        check_valid(e, "encrypted");
        check_valid(plain);
        addsub_plain(e, plain, 1);
    }
    void Evaluator::addsub_plain(Ciphertext &e, const Plaintext &plain, int op) const
    {
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
            dest.reshape_uninitialized(&lvl, 3);
            ck(k_ckks_multiply_2x2(context_.dev_mods(), nullptr, e1.data(), e2.data(), dest.data(), g, stream_), "ckks_multiply");
            dest.is_ntt_form() = true;
            dest.correction_factor() = 1;
            dest.scale() = e1.scale() * e2.scale();
            if (!scale_within_bounds(dest.scale(), lvl))
                throw std::invalid_argument("scale out of bounds");
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
        // ckks_square / bfv_square compute (c0^2, 2 c0 c1, c1^2) = the product of e with itself
        // (evaluator.cpp:878-1142); canonical results coincide with multiply(e, e).
        check_valid(e, "encrypted");
        multiply_inplace(e, e);
    }

    void Evaluator::ckks_multiply(Ciphertext &e1, const Ciphertext &e2) const
    {
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
        if (dest == 3)
        {
            e1.resize(&lvl, 3, stream_);
            ck(k_ckks_multiply_2x2(context_.dev_mods(), nullptr, e1.data(), self ? e1.data() : e2.data(), e1.data(), g, stream_),
               "ckks_multiply");
        }
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
        {
            e1.resize(&lvl, 3, stream_);
            ck(k_ckks_multiply_2x2(context_.dev_mods(), nullptr, e1.data(), self ? e1.data() : e2.data(), e1.data(), g, stream_),
               "bgv_multiply");
        }
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
        ck(k_multiply_general(mods, nullptr, x_q.p, (unsigned)s1, yq, (unsigned)s2, d_q.p, gq, stream_), "bfv tensor q");
        ck(k_multiply_general(mods, lv.bsk_prime, x_b.p, (unsigned)s1, yb, (unsigned)s2, d_b.p, gb, stream_), "bfv tensor Bsk");

        // step (5): back to coefficient form
        ck(ntt_inverse(tb, plain_batch(d_q.p, (size_t)K * N, K, (unsigned)(dest * B), 0), 0, stream_), "bfv intt q");
        NttBatch ib = plain_batch(d_b.p, (size_t)nBsk * N, nBsk, (unsigned)(dest * B), 0);
        ib.comp_prime = lv.bsk_prime;
        ck(ntt_inverse(tb, ib, 0, stream_), "bfv intt Bsk");

        // steps (6)-(8)
        size_t words = dest * B * K * N;
        uint64_t *out = DevicePool::global().alloc_words(words);
        ck(k_behz_floor_sk(mods, lv, d_q.p, d_b.p, out, n_log, dest * B, stream_), "behz floor_sk");
        e1.adopt(&lvl, dest, out, words);
    }

    // ---- relinearize (evaluator.cpp:1144-1199)
    void Evaluator::relinearize_inplace(Ciphertext &e, const KSwitchKeys &relin_keys) const
    {
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
        size_t relins_needed = size - destination_size;
        // the reference passes the LAST polynomial as the target of every step (evaluator.cpp:1180-1188)
        for (size_t I = 0; I < relins_needed; I++)
            switch_key_inplace(e, e.plane(size - 1), relin_keys, relin_index(size - 1 - I));
        e.resize(e.level(), destination_size, stream_);
        throw_if_transparent(e);
    }

    void Evaluator::relinearize_partial(Ciphertext &e, const KSwitchKeys &relin_keys, unsigned j0, unsigned j1, uint64_t *acc) const
    {
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
        if (&e.context() != &context_ || !e.level() || e.size() != 3)
            throw std::invalid_argument("encrypted is not valid for encryption parameters");
        switch_key_finish(e, acc, parts);
        e.resize(e.level(), 2, stream_);
        throw_if_transparent(e);
    }
    void Evaluator::apply_galois_partial(
        Ciphertext &e, uint32_t galois_elt, const KSwitchKeys &galois_keys, unsigned j0, unsigned j1, uint64_t *acc) const
    {
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
        if (&e.context() != &context_ || !e.level() || e.size() != 2)
            throw std::invalid_argument("encrypted is not valid for encryption parameters");
        switch_key_finish(e, acc, parts);
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
        uint64_t *acc_out, unsigned split) const
    {
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
        if (j0 < j1 && (key.digit0 > j0 || key.digit0 + key.digits < j1))
            throw std::invalid_argument("kswitch_keys inner dimension is too small");
        if (e.size() < 2)
            throw std::invalid_argument("encrypted size must be at least 2");

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
        if (read_in_place)
            ;
        else if (ntt_target && ntt2_supports(context_.log_n()))
        {
            // out-of-place: the two-pass engine reads the target and writes t
            NttBatch bt = plain_batch(t.p, (size_t)K * N, K, B, 0);
            bt.src = target;
            bt.src_outer_stride = (size_t)K * N;
            ck(ntt_inverse(tb, bt, 0, stream_), "ks intt target");
        }
        else
        {
            ck(hipMemcpyAsync(t.p, target, (size_t)B * K * N * 8, hipMemcpyDeviceToDevice, stream_), "ks copy target");
            if (ntt_target)
                ck(ntt_inverse(tb, plain_batch(t.p, (size_t)K * N, K, B, 0), 0, stream_), "ks intt target");
        }

        if (key.register_order)
        {
            // fused path (ntt2_kernels.hip): the K(K+1) raised digits go through HBM once, between
            // the two passes, and are multiplied into the key inside the second pass
            const KsTargets &kt = ks_targets(K);
            Scratch mid((size_t)B * (K + 1) * K * N);
            KsFusedArgs ka{};
            ka.t = digits;
            ka.target_ntt = ntt_target ? target : nullptr; // the I == J shortcut of evaluator.cpp:2682-2685
            ka.key = key.dev;
            ka.key_quot_off = key.quot_off;
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
            ck(ks_fused(tb, ka, stream_), "ks fused");
        }
        else if (split > 1)
            throw std::invalid_argument("in-launch digit groups need the fused key-switch path");
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

    void Evaluator::switch_key_finish(Ciphertext &e, uint64_t *acc_p, unsigned parts) const
    {
        check_valid(e, "encrypted");
        if (!acc_p)
            throw std::invalid_argument("acc");
        if (!context_.using_keyswitching())
            throw std::logic_error("keyswitching is not supported by the context");
        if (e.size() < 2)
            throw std::invalid_argument("encrypted size must be at least 2");
        if (parts < 1 || parts > 8)
            throw std::invalid_argument("parts"); // 8 canonical residues below 2^60 still fit a 64-bit word
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
                b.epi = 2;
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
            ck(ntt_inverse(tb, bi, 0, stream_), "ks intt all");
            ck(k_keyswitch_tail_bfv(
                   mods, klvl.dev.inv_q_last_mod_q, klvl.dev.round_fix, P >> 1, P, e.plane(0), e.plane(1), acc.p, n_log, K, B,
                   stream_),
               "ks tail bfv");
        }
    }

    void Evaluator::switch_key_inplace(Ciphertext &e, const uint64_t *target, const KSwitchKeys &keys, size_t key_index) const
    {
        if (&e.context() != &context_ || !e.level())
            throw std::invalid_argument("encrypted is not valid for encryption parameters");
        // Small batches do not fill the chip: one workgroup per (target modulus, tile, batch item) is 16 (K+1) workgroups per
        // ciphertext at N = 2^16, each looping over all K digits, and only 2 x 16 of them for the two 60-bit moduli.  Cut the
        // digit loop into `split` in-launch groups (their partial sums are added by the reduce pass below): single-ciphertext
        // latency of multiply+relinearize+rescale at C5 0.63 -> 0.40 ms.  SEALHIP_KS_SPLIT overrides (tests, A/B).
        const unsigned K = e.level()->K;
        unsigned split = 1;
        if (keys.context() == &context_ && key_index < keys.slots() && keys.has_key(key_index) && keys.key(key_index).register_order)
        {
            const size_t wgs = e.batch() * (size_t)(K + 1) * (context_.n() >> 12);
            split = (unsigned)(2048 / (wgs ? wgs : 1)); // measured at C5: best split 8 / 4 / 2 / 1 at batch 1 / 2 / 4 / >= 8
            if (const char *f = std::getenv("SEALHIP_KS_SPLIT"))
                split = (unsigned)std::atoi(f);
            if (split > 8)
                split = 8; // eight canonical residues below 2^61 still fit a 64-bit word
            if (split > K)
                split = K;
            if (split < 1)
                split = 1;
        }
        Scratch acc(switch_key_acc_words(e) * split);
        switch_key_partial(e, target, keys, key_index, 0, K, acc.p, split);
        if (split > 1)
            ck(k_keyswitch_reduce(context_.dev_mods(), acc.p, (unsigned)context_.log_n(), K, context_.key_level().K, (unsigned)e.batch(),
                                  stream_, split),
               "ks add digit groups");
        // CKKS at the two-pass sizes: leave the mod-down to whoever touches the ciphertext next (LazyTail) - a rescale on this
        // evaluator then does both rounding divisions with one transform per component.  SEALHIP_KS_EAGER_TAIL=1: always now.
        static const bool lazy_ok = !std::getenv("SEALHIP_KS_EAGER_TAIL");
        if (lazy_ok && context_.scheme() == Scheme::ckks && ntt2_supports(context_.log_n()) && K >= 2)
            defer_tail(e, acc.release());
        else
            switch_key_finish(e, acc.p, 1);
    }

    // relinearize (or a rotation) followed by rescale_to_next: acc = the key-switch sums, planes 0 and 1 of e = the addends.
    // Reference steps being folded: evaluator.cpp:2806-2864 (mod-down by the special prime P), then rns.cpp:830-901 on the result
    // (divide_and_round_q_last_ntt_inplace); see NttTail2 (ntt_kernels.h) for the algebra.
    void Evaluator::switch_key_finish_rescale(Ciphertext &e, uint64_t *acc_p, const Level *next, double destination_scale) const
    {
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
            std::fprintf(stderr, "[ks] folded tail\n");
        g_tail_folded++;

        // t_P: coefficient form of the special-prime sums, in place (component K of every (item, plane) of acc)
        // plus P/2: the rounding's addend goes in here, once per coefficient (NttBatch::out_add; the maps below run in mode 3)
        NttBatch bi = plain_batch(acc_p + (size_t)K * N, (size_t)(K + 1) * N, 1, 2 * B, L - 1);
        bi.out_add = P >> 1;
        ck(ntt_inverse(tb, bi, 0, stream_), "ks intt special");

        // the relinearised ciphertext's LAST component (the one rescale divides by), completed alone: c += (S - NTT(v)) P^-1
        {
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
        const unsigned m = switch_key_slots(e, nranks);
        if (!acc || !send || !sp)
            throw std::invalid_argument("buffer");
        ck(k_ks_pack_targets(acc, send, sp, (unsigned)context_.log_n(), e.level()->K, nranks, m, (unsigned)e.batch(), stream_), "ks pack targets");
    }

    void Evaluator::switch_key_finish_owned(
        const Ciphertext &e, const uint64_t *recv, const uint64_t *sp, unsigned nranks, unsigned rank, uint64_t *own) const
    {
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
        const unsigned G = (unsigned)comm.size();
        const size_t words = switch_key_acc_words(e);
        if (how == KsExchange::all_reduce || context_.scheme() != Scheme::ckks)
        {
            comm.all_reduce_sum(acc, words, stream_);
            switch_key_finish(e, acc, G);
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

    // ---- modulus switching (evaluator.cpp:1201-1647)
    void Evaluator::mod_switch_scale_to_next(Ciphertext &e) const
    {
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
                switch_key_finish_rescale(e, t.acc, next, destination_scale);
            }
            catch (...)
            {
                DevicePool::global().free_words(t.acc, stream_);
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
        check_valid(e, "encrypted");
        if (e.level() == &context_.last_level())
            throw std::invalid_argument("end of modulus switching chain reached");
        mod_switch_drop_to_next(e);
        throw_if_transparent(e);
    }

    // evaluator.cpp:1625-1647
    void Evaluator::mod_reduce_to_inplace(Ciphertext &e, const uint64_t *parms_id) const
    {
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
    void Evaluator::apply_galois_inplace(Ciphertext &e, uint32_t galois_elt, const KSwitchKeys &galois_keys) const
    {
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
        // pi(c0) goes straight into the result slab, pi(c1) into scratch as the key-switch target, c1 starts at zero
        Scratch perm(g.words());
        const size_t words = 2 * g.words();
        uint64_t *out = DevicePool::global().alloc_words(words);
        try
        {
            ck(k_apply_galois(context_.dev_mods(), e.plane(0), out, galois_elt, ntt_form, g, 1, stream_), "apply_galois c0");
            ck(k_apply_galois(context_.dev_mods(), e.plane(1), perm.p, galois_elt, ntt_form, g, 1, stream_), "apply_galois c1");
            ck(hipMemsetAsync(out + g.words(), 0, g.words() * 8, stream_), "galois zero c1");
        }
        catch (...)
        {
            DevicePool::global().free_words(out);
            throw;
        }
        e.adopt(&lvl, 2, out, words);
        switch_key_inplace(e, perm.p, galois_keys, galois_index(galois_elt));
        throw_if_transparent(e);
    }

    void Evaluator::rotate_internal(Ciphertext &e, int steps, const KSwitchKeys &galois_keys) const
    {
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
        if (&e.context() != &context_ || !e.level())
            throw std::invalid_argument("encrypted is not valid for encryption parameters");
        if (!context_.using_batching())
            throw std::logic_error("encryption parameters do not support batching");
        apply_galois_inplace(e, galois_elt_from_step(0), galois_keys);
    }
    void Evaluator::rotate_rows_inplace(Ciphertext &e, int steps, const KSwitchKeys &gk) const
    {
        if (context_.scheme() != Scheme::bfv && context_.scheme() != Scheme::bgv)
            throw std::logic_error("unsupported scheme");
        rotate_internal(e, steps, gk);
    }
    void Evaluator::rotate_columns_inplace(Ciphertext &e, const KSwitchKeys &gk) const
    {
        if (context_.scheme() != Scheme::bfv && context_.scheme() != Scheme::bgv)
            throw std::logic_error("unsupported scheme");
        conjugate_internal(e, gk);
    }
    void Evaluator::rotate_vector_inplace(Ciphertext &e, int steps, const KSwitchKeys &gk) const
    {
        if (context_.scheme() != Scheme::ckks)
            throw std::logic_error("unsupported scheme");
        rotate_internal(e, steps, gk);
    }
    void Evaluator::complex_conjugate_inplace(Ciphertext &e, const KSwitchKeys &gk) const
    {
        if (context_.scheme() != Scheme::ckks)
            throw std::logic_error("unsupported scheme");
        conjugate_internal(e, gk);
    }
} // namespace sealhip
