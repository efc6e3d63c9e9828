// An abort inside libsealhip.so (or inside the HIP runtime under it) names itself.
//
// Two ways a process that uses the library can end in abort(): an exception that escapes where none may (a destructor, a
// host worker thread) reaches std::terminate, and the ROCm runtime calls abort() itself when the device reports a memory
// access fault or a queue error.  Either way the host sees "Aborted" and nothing else if its stderr is captured (pytest
// keeps fd 2 in a temporary file that dies with the process: the one unexplained abort of round 3 left no trace).
// Both hooks are OPT-IN (round 5, ADVICE r4: process-wide handlers belong to the host program; loading a library must not
// replace them): SealHip_InstallAbortTrace(path) - or SEALHIP_ABORT_TRACE=<path> in the environment when the library is loaded -
//   * installs a terminate handler that prints the exception's message (nothing else: no HIP call - the runtime may be torn
//     down by then) and chains to the handler that was there before;
//   * catches SIGABRT and appends the call stack of the aborting thread to `path` (a ROCm fault handler on the stack tells a
//     device fault from a C++ one), then lets the abort proceed with the DEFAULT disposition (not a saved, possibly stale one).
#include "capi_common.h"
#include <csignal>
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <execinfo.h>
#include <fcntl.h>
#include <unistd.h>
#include <atomic>

namespace
{
    std::terminate_handler g_previous_terminate = nullptr;
    // two path buffers and the index of the live one: a handler on another thread reads a complete path whichever it sees
    char g_trace_paths[2][512] = { { 0 }, { 0 } };
    std::atomic<int> g_trace_live{ 0 };
    std::atomic<bool> g_abort_installed{ false };

    void put(int fd, const char *s)
    {
        size_t n = 0;
        while (s[n])
            n++;
        ssize_t r = ::write(fd, s, n);
        (void)r;
    }
    void put_dec(int fd, long v)
    {
        char b[24];
        int i = 23;
        b[i] = 0;
        if (v == 0)
            b[--i] = '0';
        while (v > 0 && i > 0)
        {
            b[--i] = char('0' + v % 10);
            v /= 10;
        }
        put(fd, b + i);
    }

    void on_terminate()
    {
        std::fprintf(stderr, "sealhip: std::terminate called");
        if (std::exception_ptr e = std::current_exception())
        {
            try
            {
                std::rethrow_exception(e);
            }
            catch (const std::exception &ex)
            {
                std::fprintf(stderr, " with an uncaught exception: %s", ex.what());
            }
            catch (...)
            {
                std::fprintf(stderr, " with an uncaught exception that is not a std::exception");
            }
        }
        std::fprintf(stderr, "\n");
        std::fflush(stderr);
        if (g_previous_terminate)
            g_previous_terminate();
        std::abort();
    }

    // async-signal-safe apart from backtrace()'s first-call initialisation, which install() forces ahead of time
    void on_abort(int sig)
    {
        const char *path = g_trace_paths[g_trace_live.load(std::memory_order_acquire)];
        int fd = path[0] ? ::open(path, O_WRONLY | O_CREAT | O_APPEND, 0644) : 2;
        if (fd < 0)
            fd = 2;
        put(fd, "sealhip: SIGABRT in process ");
        put_dec(fd, (long)::getpid());
        put(fd, "; call stack of the aborting thread (a ROCm fault / queue-error handler here means the device reported the error):\n");
        void *frames[64];
        const int n = ::backtrace(frames, 64);
        ::backtrace_symbols_fd(frames, n, fd);
        put(fd, "sealhip: end of call stack\n");
        if (fd != 2)
            ::close(fd);
        // let the abort proceed: default disposition (a disposition saved at install time may be stale - the host may have
        // installed its own handler since)
        ::signal(sig, SIG_DFL);
        ::raise(sig);
    }

    struct AtLoad
    {
        AtLoad()
        {
            if (const char *p = std::getenv("SEALHIP_ABORT_TRACE"))
                if (*p)
                    SealHip_InstallAbortTrace(p);
        }
    } g_at_load;
} // namespace

extern "C"
{
    SHL_FUNC SealHip_InstallAbortTrace(const char *path)
    {
        IfNullRet(path, SHL_E_POINTER);
        size_t n = std::strlen(path);
        if (n == 0 || n >= sizeof(g_trace_paths[0]))
            return SHL_E_INVALIDARG;
        const int spare = 1 - g_trace_live.load(std::memory_order_relaxed);
        std::memcpy(g_trace_paths[spare], path, n + 1);
        g_trace_live.store(spare, std::memory_order_release);
        if (g_abort_installed.exchange(true))
            return SHL_S_OK; // only the path changes
        void *warm[4];
        (void)::backtrace(warm, 4); // loads libgcc now, not inside the handler
        struct sigaction sa;
        std::memset(&sa, 0, sizeof(sa));
        sa.sa_handler = on_abort;
        sigemptyset(&sa.sa_mask);
        sa.sa_flags = SA_NODEFER;
        if (::sigaction(SIGABRT, &sa, nullptr) != 0)
        {
            // nothing was installed: a later call starts from scratch (the terminate handler used to stay behind here, and
            // the next call then chained on_terminate to itself)
            g_abort_installed = false;
            return SHL_E_UNEXPECTED;
        }
        // the terminate handler goes in LAST and exactly once per process, so that g_previous_terminate can never be on_terminate itself
        static std::atomic<bool> terminate_installed{ false };
        if (!terminate_installed.exchange(true))
            g_previous_terminate = std::set_terminate(on_terminate);
        return SHL_S_OK;
    }
}
