// Device memory pool shared by the host orchestration and the transform launchers.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>
#include <map>
#include <mutex>

namespace sealhip
{
    // Size-bucketed caching allocator for HBM scratch and ciphertext slabs (hipMalloc/hipFree
    // synchronise the device; the reference's MemoryPool plays the same role on the host,
    // native/src/seal/util/mempool.h).  Blocks are reused in stream order by a single stream.
    class DevicePool
    {
    public:
        static DevicePool &global();
        uint64_t *alloc_words(size_t words);
        void free_words(uint64_t *p);
        void release_all();
        size_t bytes_held() const { return held_; }
        ~DevicePool();

    private:
        std::mutex mu_;
        std::multimap<size_t, uint64_t *> free_;
        std::map<uint64_t *, size_t> live_;
        size_t held_ = 0;
    };

    struct Scratch
    {
        uint64_t *p = nullptr;
        explicit Scratch(size_t words) : p(DevicePool::global().alloc_words(words)) {}
        ~Scratch() { DevicePool::global().free_words(p); }
        Scratch(const Scratch &) = delete;
        Scratch &operator=(const Scratch &) = delete;
    };

} // namespace sealhip
