// DevicePool: stream-aware caching allocator for HBM scratch and slabs (see pool.h).
#include "pool.h"
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <stdexcept>
#include <string>

namespace sealhip
{
    namespace
    {
        struct ThreadState
        {
            hipStream_t stream = nullptr;
            bool scoped = false;
            // graph-capture hold (begin_hold .. end_hold on this thread)
            bool holding = false;
            std::multimap<size_t, uint64_t *> hold_free;
            std::vector<uint64_t *> hold_all;
        };
        ThreadState &ts()
        {
            static thread_local ThreadState s;
            return s;
        }
#ifdef SEALHIP_POOL_EXACT
        // sanitizer builds (tools/asan_emu.sh): a block is exactly the words that were asked for and a cached block serves only
        // requests of exactly its size, so that the heap's red zones sit directly behind the last word a kernel may touch - the
        // 256 KiB rounding below would swallow an overrun (a prefetch one tile past the end) without a report
        constexpr size_t kGran = 8;
        size_t round_bytes(size_t words)
        {
            return words ? words * 8 : kGran;
        }
        bool fits(size_t have, size_t want)
        {
            return have == want;
        }
#else
        constexpr size_t kGran = size_t(256) << 10;
        size_t round_bytes(size_t words)
        {
            size_t bytes = (words * 8 + kGran - 1) / kGran * kGran;
            return bytes ? bytes : kGran;
        }
        bool fits(size_t have, size_t want)
        {
            return have <= want + want / 4 + kGran;
        }
#endif
    } // namespace

    StreamScope::StreamScope(hipStream_t s) : prev_(ts().stream), prev_set_(ts().scoped)
    {
        ts().stream = s;
        ts().scoped = true;
    }
    StreamScope::~StreamScope()
    {
        ts().stream = prev_;
        ts().scoped = prev_set_;
    }

    DevicePool &DevicePool::global()
    {
        // never destroyed: device objects may be released during static destruction (a host library that owns handles and is
        // torn down after this one), when the bookkeeping must still be there and the HIP runtime is not asked for anything
        static DevicePool *pool = new DevicePool;
        return *pool;
    }
    hipStream_t DevicePool::thread_stream()
    {
        return ts().stream;
    }
    bool DevicePool::thread_has_scope()
    {
        return ts().scoped;
    }

    void DevicePool::register_stream(hipStream_t s)
    {
        if (!s)
            return;
        std::lock_guard<std::mutex> g(mu_);
        streams_.insert(s);
    }
    void DevicePool::unregister_stream(hipStream_t s)
    {
        if (!s)
            return;
        std::lock_guard<std::mutex> g(mu_);
        // blocks tagged with a stream that leaves the registry: its owner is about to destroy it (or stops using it);
        // whatever it queued is ordered by one event now, while the handle is still valid
        for (auto &kv : free_)
            if (kv.second.tag == Tag::stream && kv.second.stream == s)
            {
                (void)hipStreamSynchronize(s);
                kv.second.tag = Tag::idle;
            }
        streams_.erase(s);
        auto it = events_.find(s);
        if (it != events_.end())
        {
            (void)hipEventDestroy(it->second);
            events_.erase(it);
        }
    }

    // dst waits for everything queued on src so far
    void DevicePool::order_after(hipStream_t src, hipStream_t dst)
    {
        if (src == dst)
            return;
        auto it = events_.find(src);
        if (it == events_.end())
        {
            hipEvent_t ev = nullptr;
            if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess)
            {
                (void)hipStreamSynchronize(src); // no event to be had: fall back to a host wait
                return;
            }
            it = events_.emplace(src, ev).first;
        }
        if (hipEventRecord(it->second, src) != hipSuccess || hipStreamWaitEvent(dst, it->second, 0) != hipSuccess)
        {
            (void)hipGetLastError();
            (void)hipDeviceSynchronize(); // a stale stream handle: nothing finer is available
        }
        waits_++;
    }

    void DevicePool::make_usable(const Block &b, hipStream_t user)
    {
        switch (b.tag)
        {
        case Tag::idle:
            return;
        case Tag::stream:
            order_after(b.stream, user);
            return;
        case Tag::unknown:
            order_after(nullptr, user);
            for (hipStream_t s : streams_)
                order_after(s, user);
            return;
        }
    }

    uint64_t *DevicePool::alloc_words(size_t words, hipStream_t stream)
    {
        const size_t bytes = round_bytes(words);
        ThreadState &t = ts();
        if (t.holding)
        {
            // recording a graph: recycle among the recorded operations, otherwise idle or fresh blocks only (no event may be
            // recorded on or waited for by a capturing stream)
            auto h = t.hold_free.lower_bound(bytes);
            if (h != t.hold_free.end() && fits(h->first, bytes))
            {
                uint64_t *p = h->second;
                std::lock_guard<std::mutex> g(mu_);
                live_[p] = Live{ h->first, true };
                t.hold_free.erase(h);
                return p;
            }
        }
        {
            std::lock_guard<std::mutex> g(mu_);
            for (auto it = free_.lower_bound(bytes); it != free_.end() && fits(it->first, bytes); ++it)
            {
                if (t.holding && it->second.tag != Tag::idle)
                    continue;
                Block b = it->second;
                make_usable(b, stream);
                live_[b.p] = Live{ it->first, t.holding };
                free_.erase(it);
                if (t.holding)
                    t.hold_all.push_back(b.p);
                return b.p;
            }
        }
        void *p = nullptr;
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess)
        {
            (void)hipGetLastError();
            if (!t.holding)
            {
                // drop the cache and retry once (hipFree synchronises the device: no tag needs honouring)
                release_all();
                e = hipMalloc(&p, bytes);
            }
            if (e != hipSuccess)
            {
                (void)hipGetLastError();
                throw std::bad_alloc();
            }
        }
#ifdef SEALHIP_AB_SWITCHES
        // development builds only (tools/quick/step_placement_probe.py): where the runtime put each block
        if (std::getenv("SEALHIP_POOL_TRACE"))
            std::fprintf(stderr, "[pool] hipMalloc %zu bytes at %p\n", bytes, p);
#endif
        std::lock_guard<std::mutex> g(mu_);
        held_ += bytes;
        live_[static_cast<uint64_t *>(p)] = Live{ bytes, t.holding };
        if (t.holding)
            t.hold_all.push_back(static_cast<uint64_t *>(p));
        return static_cast<uint64_t *>(p);
    }

    void DevicePool::free_words(uint64_t *p)
    {
        if (!p)
            return;
        ThreadState &t = ts();
        if (t.scoped)
        {
            free_words(p, t.stream);
            return;
        }
        std::lock_guard<std::mutex> g(mu_);
        auto it = live_.find(p);
        if (it == live_.end())
            return;
        if (t.holding)
            t.hold_free.emplace(it->second.bytes, p); // recorded nodes may reference it: it stays with the graph (see below)
        else
            free_.emplace(it->second.bytes, Block{ p, Tag::unknown, nullptr });
        live_.erase(it);
    }

    void DevicePool::free_words(uint64_t *p, hipStream_t stream)
    {
        if (!p)
            return;
        ThreadState &t = ts();
        std::lock_guard<std::mutex> g(mu_);
        auto it = live_.find(p);
        if (it == live_.end())
            return;
        // While this thread records a graph EVERY block it frees stays with the recording - also one that was allocated before
        // the recording began (a destination whose old slab is replaced inside it): recorded nodes hold its address, so it must
        // not be handed to another object before the graph is destroyed, and no pool block may ever be tagged with the
        // capturing stream (another thread taking it would record an event on / wait for a stream that is capturing).
        if (t.holding)
            t.hold_free.emplace(it->second.bytes, p);
        else
            free_.emplace(it->second.bytes, Block{ p, Tag::stream, stream });
        live_.erase(it);
    }

    void DevicePool::begin_hold()
    {
        ThreadState &t = ts();
        t.holding = true;
        t.hold_free.clear();
        t.hold_all.clear();
        // the caller has drained the device: nothing queued anywhere touches a cached block
        std::lock_guard<std::mutex> g(mu_);
        for (auto &kv : free_)
            kv.second.tag = Tag::idle;
    }

    std::vector<uint64_t *> DevicePool::end_hold()
    {
        ThreadState &t = ts();
        std::vector<uint64_t *> graph_owned;
        std::lock_guard<std::mutex> g(mu_);
        // scratch that was recycled inside the recording belongs to the graph from now on; blocks still live are owned by
        // the objects that hold them (destinations resized during the recording) and go back to normal bookkeeping
        for (auto &kv : t.hold_free)
        {
            graph_owned.push_back(kv.second);
            graph_sizes_[kv.second] = kv.first;
        }
        for (uint64_t *p : t.hold_all)
        {
            auto it = live_.find(p);
            if (it != live_.end())
                it->second.held = false;
        }
        t.hold_free.clear();
        t.hold_all.clear();
        t.holding = false;
        return graph_owned;
    }

    void DevicePool::release_held(const std::vector<uint64_t *> &blocks)
    {
        std::lock_guard<std::mutex> g(mu_);
        for (uint64_t *p : blocks)
        {
            auto it = graph_sizes_.find(p);
            if (it == graph_sizes_.end())
                continue;
            free_.emplace(it->second, Block{ p, Tag::unknown, nullptr }); // replays may have run on any stream
            graph_sizes_.erase(it);
        }
    }

    void DevicePool::release_all()
    {
        std::lock_guard<std::mutex> g(mu_);
        for (auto &kv : free_)
        {
            (void)hipFree(kv.second.p); // synchronises with the device
            held_ -= kv.first;
        }
        free_.clear();
    }

    DevicePool::~DevicePool()
    {
        // process teardown: the HIP runtime may already be gone; leak rather than crash
    }

    // ---------------------------------------------------------------- host copies (see pool.h)
    namespace
    {
        void hip_ok(hipError_t e, const char *what)
        {
            if (e != hipSuccess)
                throw std::runtime_error(std::string("HIP failure in ") + what + ": " + hipGetErrorString(e));
        }
        struct Staging
        {
            static constexpr size_t kChunk = size_t(4) << 20;
            std::mutex mu;
            uint8_t *buf[2] = { nullptr, nullptr };
            hipEvent_t done[2] = { nullptr, nullptr };
            hipStream_t stream = nullptr;
            bool ready()
            {
                if (stream)
                    return true;
                for (int i = 0; i < 2; i++)
                    if (hipHostMalloc(reinterpret_cast<void **>(&buf[i]), kChunk, hipHostMallocDefault) != hipSuccess ||
                        hipEventCreateWithFlags(&done[i], hipEventDisableTiming) != hipSuccess)
                        return false;
                return hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) == hipSuccess;
            }
        };
        Staging &staging()
        {
            static Staging *s = new Staging; // never destroyed (see DevicePool::global)
            return *s;
        }
        std::atomic<bool> g_staged{ false };
    } // namespace

    void set_staged_host_copies(bool enabled)
    {
        g_staged.store(enabled);
    }

    void copy_h2d(void *dev, const void *host, size_t bytes)
    {
        if (!bytes)
            return;
        if (!g_staged.load(std::memory_order_relaxed))
        {
            hip_ok(hipMemcpy(dev, host, bytes, hipMemcpyHostToDevice), "H2D");
            return;
        }
        Staging &st = staging();
        std::lock_guard<std::mutex> g(st.mu);
        if (!st.ready())
            throw std::runtime_error("pinned staging buffers unavailable");
        hip_ok(hipDeviceSynchronize(), "sync"); // ordered after everything queued, as the blocking hipMemcpy is
        size_t off = 0;
        for (int i = 0; off < bytes; i ^= 1)
        {
            const size_t n = std::min(Staging::kChunk, bytes - off);
            hip_ok(hipEventSynchronize(st.done[i]), "staging wait"); // the previous transfer out of this bounce buffer
            std::memcpy(st.buf[i], static_cast<const uint8_t *>(host) + off, n);
            hip_ok(hipMemcpyAsync(static_cast<uint8_t *>(dev) + off, st.buf[i], n, hipMemcpyHostToDevice, st.stream), "H2D (staged)");
            hip_ok(hipEventRecord(st.done[i], st.stream), "staging record");
            off += n;
        }
        hip_ok(hipStreamSynchronize(st.stream), "staging drain");
    }

    void copy_d2h(void *host, const void *dev, size_t bytes)
    {
        if (!bytes)
            return;
        if (!g_staged.load(std::memory_order_relaxed))
        {
            hip_ok(hipMemcpy(host, dev, bytes, hipMemcpyDeviceToHost), "D2H");
            return;
        }
        Staging &st = staging();
        std::lock_guard<std::mutex> g(st.mu);
        if (!st.ready())
            throw std::runtime_error("pinned staging buffers unavailable");
        hip_ok(hipDeviceSynchronize(), "sync");
        // chunk k+1 travels while chunk k is copied out of its bounce buffer
        size_t issued = 0, taken = 0;
        size_t len[2] = { 0, 0 };
        int wi = 0, ri = 0;
        auto issue = [&]() {
            const size_t n = std::min(Staging::kChunk, bytes - issued);
            hip_ok(hipMemcpyAsync(st.buf[wi], static_cast<const uint8_t *>(dev) + issued, n, hipMemcpyDeviceToHost, st.stream), "D2H (staged)");
            hip_ok(hipEventRecord(st.done[wi], st.stream), "staging record");
            len[wi] = n;
            issued += n;
            wi ^= 1;
        };
        issue();
        while (taken < bytes)
        {
            if (issued < bytes)
                issue();
            hip_ok(hipEventSynchronize(st.done[ri]), "staging wait");
            std::memcpy(static_cast<uint8_t *>(host) + taken, st.buf[ri], len[ri]);
            taken += len[ri];
            ri ^= 1;
        }
    }
} // namespace sealhip
