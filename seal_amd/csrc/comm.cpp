// RCCL binding of comm.h.  The few entry points used are resolved with dlsym from librccl.so.1; their prototypes are
// restated here (rccl.h: ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclAllReduce, ncclReduceScatter,
// ncclAllGather, ncclBroadcast, ncclGetErrorString; ncclUint64 = 5, ncclSum = 0) so that the emulated CPU build needs
// no RCCL header.
#include "comm.h"
#include <dlfcn.h>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <string>

namespace sealhip
{
    namespace
    {
        struct UniqueId
        {
            char internal[Comm::kIdBytes];
        };
        constexpr int kUint64 = 5, kSum = 0;
        struct Rccl
        {
            void *lib = nullptr;
            int (*GetUniqueId)(UniqueId *) = nullptr;
            int (*CommInitRank)(void **, int, UniqueId, int) = nullptr;
            int (*CommDestroy)(void *) = nullptr;
            int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
            int (*ReduceScatter)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
            int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
            int (*Broadcast)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
            const char *(*GetErrorString)(int) = nullptr;
            bool ok = false;
        };
        Rccl &rccl()
        {
            static Rccl r;
            static std::once_flag once;
            std::call_once(once, [] {
                if (std::getenv("SEALHIP_COMM_NO_RCCL")) // CPU tests (emulated kernels, no device): single-rank loopback only
                    return;
                for (const char *name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" })
                    if ((r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL)))
                        break;
                if (!r.lib)
                    return;
                auto sym = [&](const char *n) { return dlsym(r.lib, n); };
                r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
                r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
                r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
                r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
                r.ReduceScatter = reinterpret_cast<decltype(r.ReduceScatter)>(sym("ncclReduceScatter"));
                r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
                r.Broadcast = reinterpret_cast<decltype(r.Broadcast)>(sym("ncclBroadcast"));
                r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
                r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce && r.ReduceScatter && r.AllGather && r.Broadcast;
            });
            return r;
        }
        void ok(int rc, const char *what)
        {
            if (rc == 0)
                return;
            const Rccl &r = rccl();
            throw std::runtime_error(std::string("RCCL failure in ") + what + ": " + (r.GetErrorString ? r.GetErrorString(rc) : "error"));
        }
        void hip_ok(hipError_t e, const char *what)
        {
            if (e != hipSuccess)
                throw std::runtime_error(std::string("HIP failure in ") + what + ": " + hipGetErrorString(e));
        }
    } // namespace

    bool Comm::rccl_available()
    {
        return rccl().ok;
    }

    void Comm::unique_id(uint8_t out[kIdBytes])
    {
        std::memset(out, 0, kIdBytes);
        if (!rccl().ok)
            return; // a single-rank loopback communicator needs no id
        UniqueId id;
        ok(rccl().GetUniqueId(&id), "ncclGetUniqueId");
        std::memcpy(out, id.internal, kIdBytes);
    }

    Comm::Comm(const uint8_t id[kIdBytes], int nranks, int rank) : nranks_(nranks), rank_(rank)
    {
        if (nranks < 1 || rank < 0 || rank >= nranks)
            throw std::invalid_argument("rank outside the communicator");
        if (nranks > 8)
            throw std::invalid_argument("at most 8 ranks: the partial sums of 8 residues below 2^61 still fit a 64-bit word");
        if (!rccl().ok)
        {
            if (nranks == 1)
                return; // loopback
            throw std::runtime_error("librccl.so.1 could not be loaded: multi-GPU exchange unavailable");
        }
        if (!id)
            throw std::invalid_argument("unique id");
        UniqueId u;
        std::memcpy(u.internal, id, kIdBytes);
        ok(rccl().CommInitRank(&comm_, nranks, u, rank), "ncclCommInitRank");
    }

    Comm::~Comm()
    {
        if (comm_)
            (void)rccl().CommDestroy(comm_);
    }

    void Comm::all_reduce_sum(uint64_t *buf, size_t words, hipStream_t stream)
    {
        if (!words)
            return;
        if (comm_)
            ok(rccl().AllReduce(buf, buf, words, kUint64, kSum, comm_, stream), "ncclAllReduce");
    }

    void Comm::reduce_scatter_sum(const uint64_t *send, uint64_t *recv, size_t words_per_rank, hipStream_t stream)
    {
        if (!words_per_rank)
            return;
        if (comm_)
            ok(rccl().ReduceScatter(send, recv, words_per_rank, kUint64, kSum, comm_, stream), "ncclReduceScatter");
        else if (send != recv)
            hip_ok(hipMemcpyAsync(recv, send, words_per_rank * 8, hipMemcpyDeviceToDevice, stream), "loopback reduce-scatter");
    }

    void Comm::all_gather(const uint64_t *send, uint64_t *recv, size_t words_per_rank, hipStream_t stream)
    {
        if (!words_per_rank)
            return;
        if (comm_)
            ok(rccl().AllGather(send, recv, words_per_rank, kUint64, comm_, stream), "ncclAllGather");
        else if (send != recv)
            hip_ok(hipMemcpyAsync(recv, send, words_per_rank * 8, hipMemcpyDeviceToDevice, stream), "loopback all-gather");
    }

    void Comm::broadcast(uint64_t *buf, size_t words, int root, hipStream_t stream)
    {
        if (root < 0 || root >= nranks_)
            throw std::invalid_argument("broadcast root");
        if (comm_ && words)
            ok(rccl().Broadcast(buf, buf, words, kUint64, root, comm_, stream), "ncclBroadcast");
    }
} // namespace sealhip
