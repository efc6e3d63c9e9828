// One rank of a communicator over the GPUs of one node: RCCL over xGMI, one process per GPU.
// The hot path has exactly one exchange step - the partial sums of a digit-parallel key switch (evaluator.cpp:
// switch_key_exchange) - plus one-time broadcasts of keys and tables; independent ciphertexts need no collective.
// librccl is loaded at run time (dlopen), so the library has no link-time dependency on it and a process that already
// carries RCCL (e.g. through torch.distributed) shares that copy.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

namespace sealhip
{
    class Comm
    {
    public:
        static constexpr size_t kIdBytes = 128; // ncclUniqueId
        // rank 0 calls unique_id() and hands the bytes to the other ranks out of band (file, socket, torch.distributed store)
        static void unique_id(uint8_t out[kIdBytes]);
        static bool rccl_available();
        // collective over all ranks (ncclCommInitRank) on the calling thread's current device.  nranks == 1 works without
        // RCCL (every collective degenerates to a copy).
        Comm(const uint8_t id[kIdBytes], int nranks, int rank);
        ~Comm();
        Comm(const Comm &) = delete;
        Comm &operator=(const Comm &) = delete;

        int rank() const { return rank_; }
        int size() const { return nranks_; }
        bool loopback() const { return comm_ == nullptr; }
        // all on `stream`, asynchronous, in place where send == recv; counts in 64-bit words
        void all_reduce_sum(uint64_t *buf, size_t words, hipStream_t stream);
        void reduce_scatter_sum(const uint64_t *send, uint64_t *recv, size_t words_per_rank, hipStream_t stream);
        void all_gather(const uint64_t *send, uint64_t *recv, size_t words_per_rank, hipStream_t stream);
        void broadcast(uint64_t *buf, size_t words, int root, hipStream_t stream);

    private:
        void *comm_ = nullptr; // ncclComm_t
        int nranks_ = 1, rank_ = 0;
    };

    // the contiguous share [first, first + count) of `total` items owned by `rank` of `world` (sizes differ by at most one):
    // the digits a rank multiplies and the target moduli it reduces
    inline void comm_split(unsigned total, unsigned world, unsigned rank, unsigned &first, unsigned &count)
    {
        const unsigned base = total / world, extra = total % world;
        first = rank * base + (rank < extra ? rank : extra);
        count = base + (rank < extra ? 1 : 0);
    }
} // namespace sealhip
