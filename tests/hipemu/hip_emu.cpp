// TEST INFRASTRUCTURE ONLY — runtime of the fiber-based kernel emulator declared in
// tests/hipemu/include/hip/hip_runtime.h (see the header for scope and caveats).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <ucontext.h>
#include <mutex>
#include <vector>

uint3_emu threadIdx, blockIdx;
dim3 blockDim, gridDim;

namespace hipemu
{
    namespace
    {
        constexpr size_t kStack = 256 * 1024;
        constexpr int kMaxThreads = 1024;
        constexpr size_t kLds = 160 * 1024;

        struct Fiber
        {
            ucontext_t ctx;
            char *stack = nullptr;
            bool done = false;
            bool at_barrier = false;
            bool at_shfl = false;
            uint3_emu tid;
            int linear;
        };

        alignas(16) char g_lds[kLds];
        std::vector<Fiber> g_fibers;
        ucontext_t g_sched;
        int g_cur = -1;
        const std::function<void()> *g_body = nullptr;

        // wave mailboxes: [wave][generation parity][lane]
        uint64_t g_mail[kMaxThreads / 64 + 1][2][64];
        int g_arrived[kMaxThreads / 64 + 1];
        int g_gen[kMaxThreads / 64 + 1];
        int g_wave_size[kMaxThreads / 64 + 1];

        void trampoline()
        {
            (*g_body)();
            g_fibers[g_cur].done = true;
            swapcontext(&g_fibers[g_cur].ctx, &g_sched);
        }

        void yield()
        {
            int me = g_cur;
            swapcontext(&g_fibers[me].ctx, &g_sched);
            // resumed: scheduler restored the thread coordinates
        }
    } // namespace

    void *dyn_lds()
    {
        // static __shared__ arrays live in their own storage; dynamic LDS gets the whole buffer
        return g_lds;
    }

    int lane_id()
    {
        return g_fibers[g_cur].linear & 63;
    }

    void syncthreads()
    {
        g_fibers[g_cur].at_barrier = true;
        yield();
    }

    uint64_t shfl_exchange(uint64_t value, int src_lane)
    {
        Fiber &f = g_fibers[g_cur];
        int wave = f.linear >> 6, lane = f.linear & 63;
        int gen = g_gen[wave];
        g_mail[wave][gen & 1][lane] = value;
        g_arrived[wave]++;
        f.at_shfl = true;
        while (g_gen[wave] == gen) // released by the scheduler once the whole wave arrived
            yield();
        f.at_shfl = false;
        return g_mail[wave][gen & 1][src_lane];
    }

    // the fibers and their scheduler are process-wide state: launches of several host threads (the threaded tests of the
    // library's own locking) run one after the other
    static std::mutex g_launch_mu;

    void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()> &body)
    {
        std::lock_guard<std::mutex> one_at_a_time(g_launch_mu);
        int nthreads = (int)(block.x * block.y * block.z);
        if (nthreads > kMaxThreads || shmem > kLds)
        {
            std::fprintf(stderr, "hipemu: launch exceeds limits (%d threads, %zu LDS)\n", nthreads, shmem);
            std::abort();
        }
        if ((int)g_fibers.size() < nthreads)
        {
            size_t old = g_fibers.size();
            g_fibers.resize(nthreads);
            for (size_t i = old; i < g_fibers.size(); i++)
                g_fibers[i].stack = (char *)std::malloc(kStack);
        }
        gridDim = grid;
        blockDim = block;
        g_body = &body;
        for (unsigned bz = 0; bz < grid.z; bz++)
            for (unsigned by = 0; by < grid.y; by++)
                for (unsigned bx = 0; bx < grid.x; bx++)
                {
                    int nwaves = (nthreads + 63) / 64;
                    for (int w = 0; w < nwaves; w++)
                    {
                        g_arrived[w] = 0;
                        g_gen[w] = 0;
                        g_wave_size[w] = (w == nwaves - 1) ? nthreads - 64 * w : 64;
                    }
                    for (int t = 0; t < nthreads; t++)
                    {
                        Fiber &f = g_fibers[t];
                        f.done = f.at_barrier = f.at_shfl = false;
                        f.linear = t;
                        f.tid.x = t % block.x;
                        f.tid.y = (t / block.x) % block.y;
                        f.tid.z = t / (block.x * block.y);
                        getcontext(&f.ctx);
                        f.ctx.uc_stack.ss_sp = f.stack;
                        f.ctx.uc_stack.ss_size = kStack;
                        f.ctx.uc_link = &g_sched;
                        makecontext(&f.ctx, trampoline, 0);
                    }
                    int live = nthreads;
                    while (live > 0)
                    {
                        int waiting = 0;
                        for (int t = 0; t < nthreads; t++)
                        {
                            Fiber &f = g_fibers[t];
                            if (f.done)
                                continue;
                            if (f.at_barrier)
                            {
                                waiting++;
                                continue;
                            }
                            blockIdx.x = bx;
                            blockIdx.y = by;
                            blockIdx.z = bz;
                            threadIdx = f.tid;
                            g_cur = t;
                            swapcontext(&g_sched, &f.ctx);
                            if (f.done)
                                live--;
                            else if (f.at_barrier)
                                waiting++;
                            // release a wave-level shuffle rendezvous when complete
                            int w = t >> 6;
                            if (g_arrived[w] == g_wave_size[w])
                            {
                                g_arrived[w] = 0;
                                g_gen[w]++;
                            }
                        }
                        if (live > 0 && waiting == live)
                            for (int t = 0; t < nthreads; t++)
                                g_fibers[t].at_barrier = false;
                    }
                }
        g_body = nullptr;
    }
} // namespace hipemu
