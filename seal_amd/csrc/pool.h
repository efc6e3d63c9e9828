// Device memory pool shared by the host orchestration and the transform launchers.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>
#include <map>
#include <mutex>
#include <set>
#include <vector>

namespace sealhip
{
    // Size-bucketed caching allocator for HBM scratch and ciphertext slabs (hipMalloc/hipFree synchronise the device; the
    // reference's MemoryPool plays the same role on the host, native/src/seal/util/mempool.h).
    //
    // Stream safety.  free_words() returns a block while kernels that read or write it may still be queued, so every cached
    // block carries the stream that work was issued on:
    //   * a block handed back to the SAME stream is reused at once (stream order protects it);
    //   * a block handed to ANOTHER stream T makes T wait first (an event recorded on the tagged stream at hand-out time,
    //     which covers everything queued there before the free);
    //   * a block freed outside any StreamScope (object destructors called from arbitrary host code) is tagged "unknown":
    //     the next user waits for every stream registered with the pool and for the NULL stream.
    // The calling thread's stream is a thread-local set by StreamScope at the C-ABI boundary (Evaluator entry points use the
    // evaluator's stream); without a scope allocations are ordered on the NULL stream, which is what the Encryptor,
    // Decryptor, KeyGenerator and the encoders launch on.
    //
    // Graph capture.  Between begin_hold() and end_hold() (Evaluator::begin_capture / end_capture, same thread) scratch is
    // taken only from blocks known to be idle or freshly allocated, is recycled among the recorded operations, and is then
    // handed to the graph: the addresses are baked into the executable graph, so the blocks stay out of the pool until
    // release_held() (Graph_Destroy).
    class DevicePool
    {
    public:
        static DevicePool &global();
        static hipStream_t thread_stream();
        static bool thread_has_scope();

        uint64_t *alloc_words(size_t words) { return alloc_words(words, thread_stream()); }
        uint64_t *alloc_words(size_t words, hipStream_t stream);
        void free_words(uint64_t *p);                     // tagged with the thread's scope stream, "unknown" outside a scope
        void free_words(uint64_t *p, hipStream_t stream); // work touching p was issued on `stream`
        void register_stream(hipStream_t s);
        void unregister_stream(hipStream_t s);
        void release_all();
        size_t bytes_held() const { return held_; }
        // counters for the tests: cross-stream hand-outs that had to be ordered by an event
        size_t cross_stream_waits() const { return waits_; }

        void begin_hold(); // after the caller drained the device: every cached block is idle
        std::vector<uint64_t *> end_hold();
        void release_held(const std::vector<uint64_t *> &blocks);
        ~DevicePool();

    private:
        enum class Tag
        {
            idle,    // no queued work touches the block
            stream,  // work on `stream` may
            unknown, // work on any registered stream (or the NULL stream) may
        };
        struct Block
        {
            uint64_t *p;
            Tag tag;
            hipStream_t stream;
        };
        struct Live
        {
            size_t bytes;
            bool held;
        };
        void order_after(hipStream_t src, hipStream_t dst); // mu_ held
        void make_usable(const Block &b, hipStream_t user);  // mu_ held
        std::mutex mu_;
        std::multimap<size_t, Block> free_;
        std::map<uint64_t *, Live> live_;
        std::set<hipStream_t> streams_;
        std::map<hipStream_t, hipEvent_t> events_;
        std::map<uint64_t *, size_t> graph_sizes_; // blocks owned by executable graphs
        size_t held_ = 0;
        size_t waits_ = 0;
    };

    // Orders this thread's pool traffic on `s` for the lifetime of the object (nests; restores the previous state).
    struct StreamScope
    {
        explicit StreamScope(hipStream_t s);
        ~StreamScope();
        StreamScope(const StreamScope &) = delete;
        StreamScope &operator=(const StreamScope &) = delete;

    private:
        hipStream_t prev_;
        bool prev_set_;
    };

    // Host <-> device copies of words that live in the CALLER's pageable memory (ciphertexts, plaintexts, keys of a host
    // library).  Direct (default): hipMemcpy on the caller's buffer - the runtime pins the buffer's pages for the DMA and keeps
    // that registration cached.  Staged: through two pinned bounce buffers owned by this library, so that the caller's pages
    // are NEVER registered with the driver.  A host that changes the protection of its own buffers (the device-resident
    // drop-in, integration/seal_evaluator_hip.cpp: mprotect shadows) needs the staged form: a registered range that becomes
    // inaccessible makes the driver re-validate it on every later submission (measured 20 - 40 ms per operation,
    // profiles/r02_dropin_chain.txt).  Both are synchronous: the copy is complete on return.
    void set_staged_host_copies(bool enabled);
    void copy_h2d(void *device_dst, const void *host_src, size_t bytes);
    void copy_d2h(void *host_dst, const void *device_src, size_t bytes);

    struct Scratch
    {
        uint64_t *p = nullptr;
        explicit Scratch(size_t words) : p(DevicePool::global().alloc_words(words)) {}
        ~Scratch() { DevicePool::global().free_words(p); }
        uint64_t *release() // hand the block to another owner
        {
            uint64_t *r = p;
            p = nullptr;
            return r;
        }
        Scratch(const Scratch &) = delete;
        Scratch &operator=(const Scratch &) = delete;
    };

} // namespace sealhip
