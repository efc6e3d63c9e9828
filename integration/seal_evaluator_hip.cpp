// seal::Evaluator implemented on libsealhip.so — the header-compatible link flavour of INTEGRATION.md §2.
//
// This file is what a maintainer of the reference adds NEXT TO native/src/seal/evaluator.cpp and compiles INSTEAD of it:
// it defines the out-of-line members of seal::Evaluator (native/src/seal/evaluator.h:79-1387) against the reference's own,
// unmodified headers and forwards every operation to the C ABI of include/sealhip.h.  Programs written against
// seal::Evaluator / seal::Ciphertext / seal::RelinKeys / seal::GaloisKeys link unchanged.
//
// Shape of a call: validate what the reference validates before touching data (so the same exception class is thrown for
// the same input), bring the operands to the device, run the device operation, update the host object's bookkeeping.
// Device state lives in a process-wide registry keyed by SEALContext::key_parms_id() because the class layout is fixed by
// the header (its only member is the context, evaluator.h:1385).  Key-switching keys are uploaded once per key object.
//
// DEVICE-RESIDENT CIPHERTEXTS.  A seal::Ciphertext is a host object whose words are reached through inline
// accessors, so the only way to keep a chain of operations on the device without touching the headers is to make the HOST
// BUFFER ITSELF tell us when somebody looks at it.  Every result stays in HBM as the "mirror" of the host buffer it belongs
// to (registry keyed by Ciphertext::data()); the host object gets its size / parms_id / scale updated, its buffer is resized
// WITHOUT copy or zero-fill, and the whole pages of that buffer are made inaccessible (mprotect PROT_NONE).  The next
// Evaluator call on the object finds the mirror and runs without any PCIe traffic.  The first host access - Decryptor,
// save(), operator=, user code, or the pool handing the buffer to another object - faults, the SIGSEGV handler copies the
// words down, opens the pages and drops the mirror, and the access proceeds.  Operands uploaded from valid host memory are
// kept as read-only mirrors (PROT_READ) so that a second use skips the upload; a host write drops them.  With the
// reference's allocator hook (SEAL_MALLOC / SEAL_FREE, util/defines.h:170-179, pointed at seal_alloc_hook.cpp by
// integration/config_hip) pool chunks are page-aligned, so a ciphertext buffer is whole pages; without the hook the
// unaligned head and tail of a buffer are copied down eagerly after every operation.
// This mode is OPT-IN since round 3 (SEALHIP_DROPIN_RESIDENT=1 or sealhip_dropin_set_resident(1)): it asks things of the host
// that the reference's contract does not (one thread at a time on a shadowed buffer, no system call on it without
// sealhip_dropin_settle() first, SIGSEGV left to this library) - INTEGRATION.md section 2c.  The DEFAULT uploads the operands
// and downloads the results of every call: valid for any thread and any kind of access, 40 ms instead of 12 ms for the
// chained program of section 2b.
//
// Built by integration/Makefile into integration/_build/libsealdropin*.so together with the reference's other objects;
// tests/test_dropin.py drives it through the same flat C shim as the real reference and compares word for word.
#include "seal/evaluator.h"
#include "seal/valcheck.h"
#include "sealhip.h"
#include <execinfo.h>
#include <signal.h>
#include <sys/mman.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

namespace
{
    using namespace seal;

    [[noreturn]] void raise(long hr)
    {
        uint64_t n = 0;
        SealHip_LastError(nullptr, &n);
        std::string msg(n ? n : 1, '\0');
        SealHip_LastError(&msg[0], &n);
        msg.resize(std::strlen(msg.c_str()));
        switch ((unsigned long)hr & 0xFFFFFFFFul)
        {
        case 0x80070057ul: // E_INVALIDARG
            throw std::invalid_argument(msg);
        case 0x80131509ul: // COR_E_INVALIDOPERATION
            throw std::logic_error(msg);
        case 0x80070585ul: // ERROR_INVALID_INDEX
            throw std::out_of_range(msg);
        case 0x8007000Eul:
            throw std::bad_alloc();
        default:
            throw std::runtime_error(msg.empty() ? "sealhip failure" : msg);
        }
    }
    inline void ck(long hr)
    {
        if (hr != 0)
            raise(hr);
    }

    // Device mirror of one SEALContext (all levels), its evaluator and the keys uploaded so far.
    struct Dev
    {
        void *ctx = nullptr, *ev = nullptr;
        std::mutex mu;
        std::map<std::pair<const void *, std::size_t>, void *> keys; // (first word of the key object, #indices) -> handle
    };

    Dev &device_for(const SEALContext &c)
    {
        static std::mutex mu;
        static std::map<parms_id_type, std::unique_ptr<Dev>> registry;
        std::lock_guard<std::mutex> g(mu);
        auto &slot = registry[c.key_parms_id()];
        if (slot)
            return *slot;
        auto d = std::make_unique<Dev>();
        const auto &parms = c.key_context_data()->parms();
        void *ep = nullptr;
        ck(EncParams_Create1(static_cast<uint8_t>(parms.scheme()), &ep));
        ck(EncParams_SetPolyModulusDegree(ep, parms.poly_modulus_degree()));
        std::vector<uint64_t> q;
        for (auto &m : parms.coeff_modulus())
            q.push_back(m.value());
        ck(EncParams_SetCoeffModulus(ep, q.size(), q.data()));
        if (parms.scheme() != scheme_type::ckks)
            ck(EncParams_SetPlainModulus2(ep, parms.plain_modulus().value()));
        // the chain is expanded on the device iff the reference context has levels below the first one
        const bool expanded = c.first_context_data()->next_context_data() != nullptr || c.first_context_data() == c.last_context_data();
        ck(SEALContext_Create(ep, expanded, 0, &d->ctx));
        EncParams_Destroy(ep);
        // name the levels by the reference's own parms_ids (BLAKE2b of the parameters)
        for (auto cd = c.key_context_data(); cd; cd = cd->next_context_data())
        {
            parms_id_type pid = cd->parms_id();
            ck(SEALContext_SetParmsId(d->ctx, cd->chain_index(), pid.data()));
        }
        ck(Evaluator_Create(d->ctx, &d->ev));
#ifdef SEAL_THROW_ON_TRANSPARENT_CIPHERTEXT
        ck(Evaluator_SetTransparentCheck(d->ev, true));
#endif
        slot = std::move(d);
        return *slot;
    }

    // RAII device handles
    struct DCt
    {
        void *h = nullptr;
        ~DCt()
        {
            if (h)
                Ciphertext_Destroy(h);
        }
    };
    struct DPt
    {
        void *h = nullptr;
        ~DPt()
        {
            if (h)
                Plaintext_Destroy(h);
        }
    };

    // ------------------------------------------------------------------------------------------------ host mirrors
    struct Mirror
    {
        std::uintptr_t begin = 0;      // host buffer [begin, begin + bytes)
        std::size_t bytes = 0;
        std::uintptr_t pbegin = 0, pend = 0; // the whole pages inside it (what is protected)
        void *dev = nullptr;           // device Ciphertext handle (owned)
        Dev *owner = nullptr;
        parms_id_type parms_id{};
        bool host_valid = false;       // true: host == device, pages PROT_READ;  false: host stale, pages PROT_NONE
    };
    struct Mirrors
    {
        std::recursive_mutex mu;
        std::map<std::uintptr_t, Mirror> by_begin;
        std::size_t page = 4096;
        bool eager = false, trace = false;
        bool handler_installed = false, atexit_registered = false; // (guarded by mu like everything else here)
        struct sigaction previous{};
        // statistics (tests, tools): transfers avoided / done
        std::size_t uploads = 0, downloads = 0, reused = 0, faults = 0;
    };
    Mirrors &mirrors();
    void segv_handler(int sig, siginfo_t *info, void *uctx);
    void settle_all_at_exit();
    void enable_resident(Mirrors &mm);

    // Device-resident shadows are OPT-IN (round 3): SEALHIP_DROPIN_RESIDENT=1 or sealhip_dropin_set_resident(1).
    // The default copies every result down before the call returns, which is what the reference's contract needs in general -
    // any thread, any kind of access to Ciphertext::data() (also from system calls, which do not fault on a protected page but
    // fail with EFAULT), no signal handler in the process.  See INTEGRATION.md section 2c for what resident mode asks of a host.
    // (the caller holds mm.mu)
    void enable_resident(Mirrors &mm)
    {
        if (!mm.handler_installed)
        {
            mm.handler_installed = true;
            // the library must not register our (pageable) buffers with the driver: their protection changes
            SealHip_SetStagedHostCopies(true);
            // At process exit every shadow is settled while the HIP runtime is still alive (this handler is registered late, so
            // it runs before the destructors of the pools and of the runtime): afterwards nothing is protected any more and
            // later calls, if any, copy eagerly.
            if (!mm.atexit_registered)
            {
                mm.atexit_registered = true;
                std::atexit(settle_all_at_exit);
            }
            struct sigaction sa{};
            sa.sa_sigaction = segv_handler;
            sa.sa_flags = SA_SIGINFO | SA_NODEFER;
            sigemptyset(&sa.sa_mask);
            sigaction(SIGSEGV, &sa, &mm.previous);
        }
        mm.eager = false;
    }
    // the way back: every shadow settled by the caller, the process gets its SIGSEGV disposition and the library its
    // direct host copies back - unless somebody installed another handler on top of ours in the meantime, which is then left alone
    void disable_resident(Mirrors &mm)
    {
        mm.eager = true;
        if (!mm.handler_installed)
            return;
        struct sigaction current{};
        sigaction(SIGSEGV, nullptr, &current);
        if ((current.sa_flags & SA_SIGINFO) && current.sa_sigaction == segv_handler)
        {
            sigaction(SIGSEGV, &mm.previous, nullptr);
            mm.handler_installed = false;
            SealHip_SetStagedHostCopies(false);
        }
        else if (mm.trace)
            std::fprintf(stderr, "[sealhip dropin] another SIGSEGV handler sits on top of ours: left in place, ours stays installed underneath\n");
    }
    Mirrors &mirrors()
    {
        static Mirrors *m = [] {
            auto *mm = new Mirrors; // never destroyed: the handler may run during process teardown
            mm->page = (std::size_t)sysconf(_SC_PAGESIZE);
            mm->eager = true;
            mm->trace = std::getenv("SEALHIP_DROPIN_TRACE") != nullptr;
            if (std::getenv("SEALHIP_DROPIN_RESIDENT") && std::getenv("SEALHIP_DROPIN_EAGER"))
                std::fprintf(stderr, "[sealhip dropin] SEALHIP_DROPIN_RESIDENT and SEALHIP_DROPIN_EAGER are both set: EAGER wins, results are copied down in every call\n");
            else if (std::getenv("SEALHIP_DROPIN_RESIDENT"))
                enable_resident(*mm);
            return mm;
        }();
        return *m;
    }
    void protect(const Mirror &m, int prot)
    {
        if (m.pend > m.pbegin)
            mprotect(reinterpret_cast<void *>(m.pbegin), m.pend - m.pbegin, prot);
    }
    // host copy of a stale mirror becomes current (pages opened first)
    void bring_down(Mirrors &mm, Mirror &m)
    {
        protect(m, PROT_READ | PROT_WRITE);
        if (!m.host_valid)
        {
            ck(Ciphertext_CopyToHost(m.dev, reinterpret_cast<uint64_t *>(m.begin), m.bytes / 8));
            mm.downloads++;
            m.host_valid = true;
        }
    }
    void destroy(Mirror &m)
    {
        if (m.dev)
            Ciphertext_Destroy(m.dev);
        m.dev = nullptr;
    }
    void settle_all_at_exit()
    {
        Mirrors &m = mirrors();
        std::lock_guard<std::recursive_mutex> g(m.mu);
        for (auto &kv : m.by_begin)
        {
            try
            {
                bring_down(m, kv.second);
            }
            catch (...)
            {
                protect(kv.second, PROT_READ | PROT_WRITE);
            }
            destroy(kv.second);
        }
        m.by_begin.clear();
        m.eager = true;
    }
    // every mirror overlapping [lo, hi): host made current, pages opened, mirror dropped
    void resolve_range(std::uintptr_t lo, std::uintptr_t hi, const Mirror *keep = nullptr)
    {
        Mirrors &mm = mirrors();
        std::lock_guard<std::recursive_mutex> g(mm.mu);
        auto it = mm.by_begin.lower_bound(lo);
        if (it != mm.by_begin.begin())
            --it;
        while (it != mm.by_begin.end() && it->first < hi)
        {
            Mirror &m = it->second;
            if (&m != keep && m.begin < hi && m.begin + m.bytes > lo)
            {
                bring_down(mm, m);
                destroy(m);
                it = mm.by_begin.erase(it);
            }
            else
                ++it;
        }
    }
    // called by the allocator hook when a pool chunk goes back to the system: nothing may stay protected or registered
    void forget_range(std::uintptr_t lo, std::uintptr_t hi)
    {
        Mirrors &mm = mirrors();
        std::lock_guard<std::recursive_mutex> g(mm.mu);
        auto it = mm.by_begin.lower_bound(lo);
        if (it != mm.by_begin.begin())
            --it;
        while (it != mm.by_begin.end() && it->first < hi)
        {
            Mirror &m = it->second;
            if (m.begin < hi && m.begin + m.bytes > lo)
            {
                protect(m, PROT_READ | PROT_WRITE); // the contents die with the chunk: no copy
                destroy(m);
                it = mm.by_begin.erase(it);
            }
            else
                ++it;
        }
    }
    void segv_handler(int sig, siginfo_t *info, void *uctx)
    {
        Mirrors &mm = mirrors();
        const std::uintptr_t addr = reinterpret_cast<std::uintptr_t>(info->si_addr);
        bool ours = false;
        {
            std::lock_guard<std::recursive_mutex> g(mm.mu);
            auto it = mm.by_begin.upper_bound(addr);
            if (it != mm.by_begin.begin())
            {
                --it;
                Mirror &m = it->second;
                if (addr >= m.pbegin && addr < m.pend)
                {
                    ours = true;
                    mm.faults++;
                    if (mm.trace)
                        std::fprintf(stderr, "[dropin] fault at %p in mirror [%p, +%zu) host_valid=%d\n", info->si_addr,
                                     reinterpret_cast<void *>(m.begin), m.bytes, (int)m.host_valid);
                    if (mm.trace)
                    {
                        void *frames[24];
                        backtrace_symbols_fd(frames, backtrace(frames, 24), 2);
                    }
                    bool write = true; // without the error code every fault is treated as a write (the mirror is dropped)
#if defined(__x86_64__) && defined(REG_ERR)
                    write = (static_cast<ucontext_t *>(uctx)->uc_mcontext.gregs[REG_ERR] & 0x2) != 0;
#endif
                    try
                    {
                        bring_down(mm, m);
                    }
                    catch (...)
                    {
                        // the device copy is unreachable: nothing sensible can be returned to the faulting access
                        std::abort();
                    }
                    if (write)
                    {
                        if (mm.trace)
                            std::fprintf(stderr, "[dropin]   write: mirror dropped\n");
                        destroy(m);
                        mm.by_begin.erase(it);
                    }
                    else
                        protect(m, PROT_READ); // still mirrored: a later write drops it
                }
            }
        }
        if (ours)
            return; // the faulting instruction is retried
        // not ours: hand over to whoever was installed before (or the default action)
        if (mm.previous.sa_flags & SA_SIGINFO)
        {
            if (mm.previous.sa_sigaction)
            {
                mm.previous.sa_sigaction(sig, info, uctx);
                return;
            }
        }
        else if (mm.previous.sa_handler != SIG_DFL && mm.previous.sa_handler != SIG_IGN)
        {
            mm.previous.sa_handler(sig);
            return;
        }
        signal(SIGSEGV, SIG_DFL); // re-executing the instruction now terminates the process as it would have
    }

    std::size_t word_count(const Ciphertext &x)
    {
        return x.size() * x.coeff_modulus_size() * x.poly_modulus_degree();
    }
    void push_metadata(void *h, const Ciphertext &x)
    {
        ck(Ciphertext_SetIsNTTForm(h, x.is_ntt_form()));
        ck(Ciphertext_SetScale(h, x.scale()));
        ck(Ciphertext_SetCorrectionFactor(h, x.correction_factor()));
    }
    // fresh device handle holding x's words, copied from the host buffer
    void *upload_new(Dev &d, const Ciphertext &x)
    {
        void *h = nullptr;
        ck(Ciphertext_Create3(d.ctx, nullptr, &h));
        try
        {
            if (x.size())
            {
                parms_id_type pid = x.parms_id();
                ck(Ciphertext_Resize1(h, d.ctx, pid.data(), x.size()));
                ck(Ciphertext_CopyFromHost(h, x.data(), word_count(x)));
            }
            push_metadata(h, x);
        }
        catch (...)
        {
            Ciphertext_Destroy(h);
            throw;
        }
        return h;
    }
    // A device handle for an operand.  owned == true: the caller destroys it (or hands it to publish()).
    struct Operand
    {
        void *h = nullptr;
        bool owned = false;
    };
    // the mirror of x if it is still x's: same buffer, same extent, same level, same context
    Mirror *find_mirror(Mirrors &mm, Dev &d, const Ciphertext &x)
    {
        auto it = mm.by_begin.find(reinterpret_cast<std::uintptr_t>(x.data()));
        if (it == mm.by_begin.end())
            return nullptr;
        Mirror &m = it->second;
        if (m.owner != &d || m.bytes != word_count(x) * 8 || m.parms_id != x.parms_id())
            return nullptr;
        uint64_t dev_size = 0;
        if (Ciphertext_Size(m.dev, &dev_size) != 0 || dev_size != x.size())
            return nullptr;
        return &m;
    }
    Operand acquire(Dev &d, const Ciphertext &x)
    {
        Mirrors &mm = mirrors();
        const std::uintptr_t lo = reinterpret_cast<std::uintptr_t>(x.data()), hi = lo + word_count(x) * 8;
        if (x.size() && !mm.eager)
        {
            std::lock_guard<std::recursive_mutex> g(mm.mu);
            if (Mirror *m = find_mirror(mm, d, x))
            {
                push_metadata(m->dev, x); // scale / form flags are host-side fields the caller may have changed
                mm.reused++;
                return Operand{ m->dev, false };
            }
        }
        // the host buffer is the source: whatever else claims these addresses is settled first, from ordinary context (a copy
        // routine of the runtime must never fault on a protected page)
        if (x.size())
            resolve_range(lo, hi);
        Operand op{ upload_new(d, x), true };
        mm.uploads++;
        return op;
    }
    // x := the result held by device handle h (ownership of h passes to the registry unless the buffer is too small to protect)
    void publish(const SEALContext &c, Dev &d, void *h, Ciphertext &x)
    {
        Mirrors &mm = mirrors();
        uint64_t size = 0, pid_words[4];
        ck(Ciphertext_Size(h, &size));
        ck(Ciphertext_ParmsId(h, pid_words));
        parms_id_type pid;
        std::memcpy(pid.data(), pid_words, sizeof(pid_words));
        bool ntt = false;
        double scale = 1.0;
        uint64_t cf = 1;
        ck(Ciphertext_IsNTTForm(h, &ntt));
        ck(Ciphertext_Scale(h, &scale));
        ck(Ciphertext_CorrectionFactor(h, &cf));
        auto cd = c.get_context_data(pid);
        const std::size_t need = cd ? size * cd->parms().coeff_modulus().size() * cd->parms().poly_modulus_degree() : 0;
        {
            // x's present buffer: any mirror on it goes away WITHOUT a copy - its contents are being replaced (the handle
            // being published may be that mirror's own)
            std::lock_guard<std::recursive_mutex> g(mm.mu);
            auto it = mm.by_begin.find(reinterpret_cast<std::uintptr_t>(x.data()));
            if (it != mm.by_begin.end())
            {
                protect(it->second, PROT_READ | PROT_WRITE);
                if (it->second.dev != h)
                    destroy(it->second);
                mm.by_begin.erase(it);
            }
        }
        if (x.data())
            resolve_range(reinterpret_cast<std::uintptr_t>(x.data()), reinterpret_cast<std::uintptr_t>(x.data()) + word_count(x) * 8);
        // host bookkeeping without touching the words: the DynArray is resized with fill_zero = false (and emptied first when it
        // has to grow, so that no old contents are copied), then Ciphertext::resize only records size / degree / parms_id
        auto &words = const_cast<DynArray<Ciphertext::ct_coeff_type> &>(x.dyn_array());
        if (words.capacity() < need)
            words.resize(0, false);
        words.resize(need, false);
        x.resize(c, pid, size);
        x.is_ntt_form() = ntt;
        x.scale() = scale;
        x.correction_factor() = cf;
        if (!need)
        {
            Ciphertext_Destroy(h);
            return;
        }
        Mirror m;
        m.begin = reinterpret_cast<std::uintptr_t>(x.data());
        m.bytes = need * 8;
        m.pbegin = (m.begin + mm.page - 1) / mm.page * mm.page;
        m.pend = (m.begin + m.bytes) / mm.page * mm.page;
        m.dev = h;
        m.owner = &d;
        m.parms_id = pid;
        m.host_valid = false;
        if (mm.eager || m.pend <= m.pbegin)
        {
            // nothing to protect (a buffer below two pages) or A/B mode: plain download
            ck(Ciphertext_CopyToHost(h, x.data(), need));
            mm.downloads++;
            Ciphertext_Destroy(h);
            return;
        }
        // the new buffer may overlap other mirrors' address ranges (the pool recycled memory): settle them
        resolve_range(m.begin, m.begin + m.bytes);
        // unaligned head / tail (absent with the page-aligned allocator hook) are kept current on the host
        // (each copy drains the device first: skipped altogether when the buffer is whole pages, so that the call returns
        // while the operation is still running)
        if (m.pbegin > m.begin)
            ck(Ciphertext_CopyWordsToHost(h, 0, (m.pbegin - m.begin) / 8, x.data()));
        if (m.begin + m.bytes > m.pend)
            ck(Ciphertext_CopyWordsToHost(h, (m.pend - m.begin) / 8, (m.begin + m.bytes - m.pend) / 8, x.data() + (m.pend - m.begin) / 8));
        std::lock_guard<std::recursive_mutex> g(mm.mu);
        protect(m, PROT_NONE);
        mm.by_begin[m.begin] = m;
        if (mm.trace)
            std::fprintf(stderr, "[dropin] publish [%p, +%zu) protected [%p, %p)\n", reinterpret_cast<void *>(m.begin), m.bytes,
                         reinterpret_cast<void *>(m.pbegin), reinterpret_cast<void *>(m.pend));
    }
    // an operand that was uploaded from valid host memory stays on the device as a read-only mirror of that memory
    void retain_clean(Dev &d, const Ciphertext &x, Operand &op)
    {
        Mirrors &mm = mirrors();
        if (!op.owned)
            return;
        if (mm.eager || !x.size())
        {
            Ciphertext_Destroy(op.h);
            op.owned = false;
            return;
        }
        Mirror m;
        m.begin = reinterpret_cast<std::uintptr_t>(x.data());
        m.bytes = word_count(x) * 8;
        m.pbegin = (m.begin + mm.page - 1) / mm.page * mm.page;
        m.pend = (m.begin + m.bytes) / mm.page * mm.page;
        // only buffers made of whole pages can be watched for writes (a write to an unprotected head / tail would go unseen)
        if (m.pbegin != m.begin || m.pend != m.begin + m.bytes)
        {
            Ciphertext_Destroy(op.h);
            op.owned = false;
            return;
        }
        m.dev = op.h;
        m.owner = &d;
        m.parms_id = x.parms_id();
        m.host_valid = true;
        std::lock_guard<std::recursive_mutex> g(mm.mu);
        protect(m, PROT_READ);
        mm.by_begin[m.begin] = m;
        op.owned = false;
        if (mm.trace)
            std::fprintf(stderr, "[dropin] operand kept on the device, host buffer [%p, +%zu) read-only\n", reinterpret_cast<void *>(m.begin), m.bytes);
    }
    void release(Operand &op)
    {
        if (op.owned && op.h)
            Ciphertext_Destroy(op.h);
        op.owned = false;
    }

    void upload(Dev &d, const Plaintext &p, DPt &out)
    {
        ck(Plaintext_Create1(d.ctx, &out.h));
        ck(Plaintext_Set4(out.h, p.coeff_count(), const_cast<uint64_t *>(p.data())));
        if (p.is_ntt_form())
        {
            parms_id_type pid = p.parms_id();
            ck(Plaintext_SetParmsId(out.h, pid.data()));
        }
        ck(Plaintext_SetScale(out.h, p.scale()));
    }
    void download(const DPt &in, Plaintext &p)
    {
        uint64_t count = 0, pid_words[4];
        ck(Plaintext_CoeffCount(in.h, &count));
        ck(Plaintext_GetParmsId(in.h, pid_words));
        p.parms_id() = parms_id_zero;
        p.resize(count);
        if (count)
            ck(Plaintext_CopyToHost(in.h, p.data(), count));
        parms_id_type pid;
        std::memcpy(pid.data(), pid_words, sizeof(pid_words));
        p.parms_id() = pid;
        double scale = 1.0;
        ck(Plaintext_Scale(in.h, &scale));
        p.scale() = scale;
    }

    // KSwitchKeys::data()[index] is a vector of `digits` public keys, each a size-2 key-level ciphertext
    // (kswitchkeys.h:340): the device slab per index is their concatenation.
    void *device_keys(Dev &d, const SEALContext &c, const KSwitchKeys &keys)
    {
        if (!is_metadata_valid_for(keys, c) || !is_buffer_valid(keys))
            throw std::invalid_argument("kswitch_keys is not valid for encryption parameters");
        const void *tag = nullptr;
        for (auto &k : keys.data())
            if (!k.empty())
            {
                tag = k[0].data().data();
                break;
            }
        std::lock_guard<std::mutex> g(d.mu);
        auto key = std::make_pair(tag, keys.data().size());
        auto it = d.keys.find(key);
        if (it != d.keys.end())
            return it->second;
        void *h = nullptr;
        ck(KSwitchKeys_Create1(&h));
        for (std::size_t index = 0; index < keys.data().size(); index++)
        {
            auto &digits = keys.data()[index];
            if (digits.empty())
                continue;
            std::vector<uint64_t> words;
            for (auto &pk : digits)
            {
                const Ciphertext &kc = pk.data();
                words.insert(words.end(), kc.data(), kc.data() + kc.size() * kc.coeff_modulus_size() * kc.poly_modulus_degree());
            }
            ck(KSwitchKeys_SetKey(h, d.ctx, index, digits.size(), words.data()));
        }
        d.keys[key] = h;
        return h;
    }

    inline double now_s()
    {
        return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    }
    // one in-place ciphertext operation on the device copy of x; x's host buffer becomes the (protected) shadow of the result
    template <class Fn>
    void unary(const SEALContext &c, Ciphertext &x, Fn fn)
    {
        Dev &d = device_for(c);
        const double t0 = mirrors().trace ? now_s() : 0;
        Operand a = acquire(d, x);
        const double t1 = mirrors().trace ? now_s() : 0;
        try
        {
            ck(fn(d, a.h));
        }
        catch (...)
        {
            release(a); // a borrowed mirror stays what it was: the device operations validate before they write
            throw;
        }
        const double t2 = mirrors().trace ? now_s() : 0;
        publish(c, d, a.h, x);
        if (mirrors().trace)
            std::fprintf(stderr, "[dropin] unary: acquire %.3f ms, operation %.3f ms, publish %.3f ms\n", (t1 - t0) * 1e3, (t2 - t1) * 1e3,
                         (now_s() - t2) * 1e3);
    }
    // destination := op(source) without a host copy of the source
    template <class Fn>
    void unary_to(const SEALContext &c, const Ciphertext &src, Ciphertext &dst, Fn fn)
    {
        if (&src == &dst)
        {
            unary(c, dst, fn);
            return;
        }
        Dev &d = device_for(c);
        Operand a = acquire(d, src);
        void *h = nullptr;
        try
        {
            ck(Ciphertext_Create2(a.h, &h)); // device-to-device copy
            ck(fn(d, h));
        }
        catch (...)
        {
            if (h)
                Ciphertext_Destroy(h);
            release(a);
            throw;
        }
        retain_clean(d, src, a);
        publish(c, d, h, dst);
    }
    template <class Fn>
    void binary(const SEALContext &c, Ciphertext &x, const Ciphertext &y, Fn fn)
    {
        Dev &d = device_for(c);
        Operand a = acquire(d, x), b;
        try
        {
            if (&x != &y)
                b = acquire(d, y);
            ck(fn(d, a.h, &x == &y ? a.h : b.h));
        }
        catch (...)
        {
            release(a);
            release(b);
            throw;
        }
        if (&x != &y)
            retain_clean(d, y, b);
        publish(c, d, a.h, x);
    }
    void need_valid(const Ciphertext &x, const SEALContext &c, const char *what)
    {
        // is_metadata_valid_for + is_buffer_valid, as every public method of the reference does first.  is_buffer_valid
        // (valcheck.cpp:221-237) is the size check plus contains_seed(), which READS the first word of the second polynomial:
        // for an object whose words live on the device (a result of this evaluator: never a seeded stream) only the size
        // check is made, so that validating an operand does not pull it down
        bool ok = is_metadata_valid_for(x, c);
        if (ok)
        {
            bool on_device = false;
            {
                Mirrors &mm = mirrors();
                std::lock_guard<std::recursive_mutex> g(mm.mu);
                auto it = mm.by_begin.find(reinterpret_cast<std::uintptr_t>(x.data()));
                on_device = it != mm.by_begin.end() && !it->second.host_valid && it->second.bytes == word_count(x) * 8;
            }
            ok = on_device ? x.dyn_array().size() == word_count(x) : is_buffer_valid(x);
        }
        if (!ok)
            throw std::invalid_argument(std::string(what) + " is not valid for encryption parameters");
    }
} // namespace

namespace seal
{
    Evaluator::Evaluator(const SEALContext &context) : context_(context)
    {
        if (!context_.parameters_set())
            throw std::invalid_argument("encryption parameters are not set correctly");
    }

    void Evaluator::negate_inplace(Ciphertext &encrypted) const
    {
        need_valid(encrypted, context_, "encrypted");
        unary(context_, encrypted, [](Dev &d, void *a) { return Evaluator_Negate(d.ev, a, a); });
    }
    void Evaluator::add_inplace(Ciphertext &encrypted1, const Ciphertext &encrypted2) const
    {
        need_valid(encrypted1, context_, "encrypted1");
        need_valid(encrypted2, context_, "encrypted2");
        binary(context_, encrypted1, encrypted2, [](Dev &d, void *a, void *b) { return Evaluator_Add(d.ev, a, b, a); });
    }
    void Evaluator::sub_inplace(Ciphertext &encrypted1, const Ciphertext &encrypted2) const
    {
        need_valid(encrypted1, context_, "encrypted1");
        need_valid(encrypted2, context_, "encrypted2");
        binary(context_, encrypted1, encrypted2, [](Dev &d, void *a, void *b) { return Evaluator_Sub(d.ev, a, b, a); });
    }
    void Evaluator::add_many(const std::vector<Ciphertext> &encrypteds, Ciphertext &destination) const
    {
        if (encrypteds.empty())
            throw std::invalid_argument("encrypteds cannot be empty");
        for (auto &e : encrypteds)
            if (&e == &destination)
                throw std::invalid_argument("encrypteds must be different from destination");
        destination = encrypteds[0];
        for (std::size_t i = 1; i < encrypteds.size(); i++)
            add_inplace(destination, encrypteds[i]);
    }
    void Evaluator::multiply_inplace(Ciphertext &encrypted1, const Ciphertext &encrypted2, MemoryPoolHandle pool) const
    {
        need_valid(encrypted1, context_, "encrypted1");
        need_valid(encrypted2, context_, "encrypted2");
        if (!pool)
            throw std::invalid_argument("pool is uninitialized");
        binary(context_, encrypted1, encrypted2, [](Dev &d, void *a, void *b) { return Evaluator_Multiply(d.ev, a, b, a, nullptr); });
    }
    void Evaluator::square_inplace(Ciphertext &encrypted, MemoryPoolHandle pool) const
    {
        need_valid(encrypted, context_, "encrypted");
        if (!pool)
            throw std::invalid_argument("pool is uninitialized");
        unary(context_, encrypted, [](Dev &d, void *a) { return Evaluator_Square(d.ev, a, a, nullptr); });
    }
    void Evaluator::relinearize_internal(
        Ciphertext &encrypted, const RelinKeys &relin_keys, std::size_t destination_size, MemoryPoolHandle pool) const
    {
        if (!context_.get_context_data(encrypted.parms_id()))
            throw std::invalid_argument("encrypted is not valid for encryption parameters");
        if (relin_keys.parms_id() != context_.key_parms_id())
            throw std::invalid_argument("relin_keys is not valid for encryption parameters");
        if (!pool)
            throw std::invalid_argument("pool is uninitialized");
        if (destination_size != 2)
        {
            if (destination_size < 2 || destination_size > encrypted.size())
                throw std::invalid_argument("destination_size must be at least 2 and less than or equal to current count");
            throw std::logic_error("the device path relinearizes down to size 2 only");
        }
        Dev &d = device_for(context_);
        void *keys = device_keys(d, context_, relin_keys);
        unary(context_, encrypted, [keys](Dev &dd, void *a) { return Evaluator_Relinearize(dd.ev, a, keys, a, nullptr); });
    }

    void Evaluator::mod_switch_to_next(const Ciphertext &encrypted, Ciphertext &destination, MemoryPoolHandle pool) const
    {
        need_valid(encrypted, context_, "encrypted");
        if (!pool)
            throw std::invalid_argument("pool is uninitialized");
        unary_to(context_, encrypted, destination, [](Dev &d, void *a) { return Evaluator_ModSwitchToNext1(d.ev, a, a, nullptr); });
    }
    void Evaluator::mod_switch_to_inplace(Ciphertext &encrypted, parms_id_type parms_id, MemoryPoolHandle pool) const
    {
        if (!context_.get_context_data(encrypted.parms_id()))
            throw std::invalid_argument("encrypted is not valid for encryption parameters");
        if (!pool)
            throw std::invalid_argument("pool is uninitialized");
        unary(context_, encrypted, [&parms_id](Dev &d, void *a) { return Evaluator_ModSwitchTo1(d.ev, a, parms_id.data(), a, nullptr); });
    }
    void Evaluator::mod_switch_drop_to_next(Plaintext &plain) const
    {
        Dev &d = device_for(context_);
        DPt p;
        upload(d, plain, p);
        ck(Evaluator_ModSwitchToNext2(d.ev, p.h, p.h));
        download(p, plain);
    }
    void Evaluator::mod_switch_to_inplace(Plaintext &plain, parms_id_type parms_id) const
    {
        if (!is_valid_for(plain, context_))
            throw std::invalid_argument("plain is not valid for encryption parameters");
        Dev &d = device_for(context_);
        DPt p;
        upload(d, plain, p);
        ck(Evaluator_ModSwitchTo2(d.ev, p.h, parms_id.data(), p.h));
        download(p, plain);
    }
    void Evaluator::rescale_to_next(const Ciphertext &encrypted, Ciphertext &destination, MemoryPoolHandle pool) const
    {
        need_valid(encrypted, context_, "encrypted");
        if (!pool)
            throw std::invalid_argument("pool is uninitialized");
        unary_to(context_, encrypted, destination, [](Dev &d, void *a) { return Evaluator_RescaleToNext(d.ev, a, a, nullptr); });
    }
    void Evaluator::rescale_to_inplace(Ciphertext &encrypted, parms_id_type parms_id, MemoryPoolHandle pool) const
    {
        need_valid(encrypted, context_, "encrypted");
        if (!pool)
            throw std::invalid_argument("pool is uninitialized");
        unary(context_, encrypted, [&parms_id](Dev &d, void *a) { return Evaluator_RescaleTo(d.ev, a, parms_id.data(), a, nullptr); });
    }
    void Evaluator::mod_reduce_to_next_inplace(Ciphertext &encrypted, MemoryPoolHandle pool) const
    {
        need_valid(encrypted, context_, "encrypted");
        if (!pool)
            throw std::invalid_argument("pool is uninitialized");
        unary(context_, encrypted, [](Dev &d, void *a) { return Evaluator_ModReduceToNext(d.ev, a, a, nullptr); });
    }
    void Evaluator::mod_reduce_to_inplace(Ciphertext &encrypted, parms_id_type parms_id, MemoryPoolHandle pool) const
    {
        // evaluator.cpp:1620-1647: drop moduli until the target level is reached
        auto context_data_ptr = context_.get_context_data(encrypted.parms_id());
        auto target_context_data_ptr = context_.get_context_data(parms_id);
        if (!context_data_ptr)
            throw std::invalid_argument("encrypted is not valid for encryption parameters");
        if (!target_context_data_ptr)
            throw std::invalid_argument("parms_id is not valid for encryption parameters");
        if (context_data_ptr->chain_index() < target_context_data_ptr->chain_index())
            throw std::invalid_argument("cannot switch to higher level modulus");
        while (encrypted.parms_id() != parms_id)
            mod_reduce_to_next_inplace(encrypted, pool);
    }

    void Evaluator::multiply_many(
        const std::vector<Ciphertext> &encrypteds, const RelinKeys &relin_keys, Ciphertext &destination, MemoryPoolHandle pool) const
    {
        if (encrypteds.empty())
            throw std::invalid_argument("encrypteds vector must not be empty");
        if (!pool)
            throw std::invalid_argument("pool is uninitialized");
        for (auto &e : encrypteds)
            if (&e == &destination)
                throw std::invalid_argument("encrypteds must be different from destination");
        Dev &d = device_for(context_);
        void *keys = device_keys(d, context_, relin_keys);
        std::vector<Operand> handles(encrypteds.size());
        std::vector<void *> raw;
        void *out = nullptr;
        try
        {
            for (std::size_t i = 0; i < encrypteds.size(); i++)
            {
                // identical operands share one device handle, as the reference detects them by data pointer (evaluator.cpp:1700)
                std::size_t same = i;
                for (std::size_t j = 0; j < i; j++)
                    if (encrypteds[j].data() == encrypteds[i].data())
                        same = j;
                if (same == i)
                    handles[i] = acquire(d, encrypteds[i]);
                raw.push_back(handles[same].h);
            }
            ck(Ciphertext_Create3(d.ctx, nullptr, &out));
            ck(Evaluator_MultiplyMany(d.ev, raw.size(), raw.data(), keys, out, nullptr));
        }
        catch (...)
        {
            if (out)
                Ciphertext_Destroy(out);
            for (auto &h : handles)
                release(h);
            throw;
        }
        for (std::size_t i = 0; i < encrypteds.size(); i++)
            retain_clean(d, encrypteds[i], handles[i]);
        publish(context_, d, out, destination);
    }
    void Evaluator::exponentiate_inplace(Ciphertext &encrypted, uint64_t exponent, const RelinKeys &relin_keys, MemoryPoolHandle pool) const
    {
        if (!context_.get_context_data(encrypted.parms_id()))
            throw std::invalid_argument("encrypted is not valid for encryption parameters");
        if (!context_.get_context_data(relin_keys.parms_id()))
            throw std::invalid_argument("relin_keys is not valid for encryption parameters");
        if (!pool)
            throw std::invalid_argument("pool is uninitialized");
        if (exponent == 0)
            throw std::invalid_argument("exponent cannot be 0");
        if (exponent == 1)
            return;
        Dev &d = device_for(context_);
        void *keys = device_keys(d, context_, relin_keys);
        unary(context_, encrypted, [keys, exponent](Dev &dd, void *a) { return Evaluator_Exponentiate(dd.ev, a, exponent, keys, a, nullptr); });
    }

    void Evaluator::add_plain_inplace(Ciphertext &encrypted, const Plaintext &plain, MemoryPoolHandle) const
    {
        need_valid(encrypted, context_, "encrypted");
        if (!is_metadata_valid_for(plain, context_) || !is_buffer_valid(plain))
            throw std::invalid_argument("plain is not valid for encryption parameters");
        Dev &d = device_for(context_);
        DPt p;
        upload(d, plain, p);
        unary(context_, encrypted, [&p](Dev &dd, void *a) { return Evaluator_AddPlain(dd.ev, a, p.h, a); });
    }
    void Evaluator::sub_plain_inplace(Ciphertext &encrypted, const Plaintext &plain, MemoryPoolHandle) const
    {
        need_valid(encrypted, context_, "encrypted");
        if (!is_metadata_valid_for(plain, context_) || !is_buffer_valid(plain))
            throw std::invalid_argument("plain is not valid for encryption parameters");
        Dev &d = device_for(context_);
        DPt p;
        upload(d, plain, p);
        unary(context_, encrypted, [&p](Dev &dd, void *a) { return Evaluator_SubPlain(dd.ev, a, p.h, a); });
    }
    void Evaluator::multiply_plain_inplace(Ciphertext &encrypted, const Plaintext &plain, MemoryPoolHandle pool) const
    {
        need_valid(encrypted, context_, "encrypted");
        if (!is_metadata_valid_for(plain, context_) || !is_buffer_valid(plain))
            throw std::invalid_argument("plain is not valid for encryption parameters");
        if (!pool)
            throw std::invalid_argument("pool is uninitialized");
        Dev &d = device_for(context_);
        DPt p;
        upload(d, plain, p);
        unary(context_, encrypted, [&p](Dev &dd, void *a) { return Evaluator_MultiplyPlain(dd.ev, a, p.h, a, nullptr); });
    }
    void Evaluator::transform_to_ntt_inplace(Plaintext &plain, parms_id_type parms_id, MemoryPoolHandle pool) const
    {
        if (!is_valid_for(plain, context_))
            throw std::invalid_argument("plain is not valid for encryption parameters");
        if (!pool)
            throw std::invalid_argument("pool is uninitialized");
        Dev &d = device_for(context_);
        DPt p;
        upload(d, plain, p);
        ck(Evaluator_TransformToNTT1(d.ev, p.h, parms_id.data(), p.h, nullptr));
        download(p, plain);
    }
    void Evaluator::transform_to_ntt_inplace(Ciphertext &encrypted) const
    {
        need_valid(encrypted, context_, "encrypted");
        unary(context_, encrypted, [](Dev &d, void *a) { return Evaluator_TransformToNTT2(d.ev, a, a); });
    }
    void Evaluator::transform_from_ntt_inplace(Ciphertext &encrypted_ntt) const
    {
        need_valid(encrypted_ntt, context_, "encrypted");
        unary(context_, encrypted_ntt, [](Dev &d, void *a) { return Evaluator_TransformFromNTT(d.ev, a, a); });
    }

    void Evaluator::apply_galois_inplace(
        Ciphertext &encrypted, uint32_t galois_elt, const GaloisKeys &galois_keys, MemoryPoolHandle pool) const
    {
        need_valid(encrypted, context_, "encrypted");
        if (galois_keys.parms_id() != context_.key_parms_id())
            throw std::invalid_argument("galois_keys is not valid for encryption parameters");
        if (!pool)
            throw std::invalid_argument("pool is uninitialized");
        Dev &d = device_for(context_);
        void *keys = device_keys(d, context_, galois_keys);
        unary(context_, encrypted, [keys, galois_elt](Dev &dd, void *a) { return Evaluator_ApplyGalois(dd.ev, a, galois_elt, keys, a, nullptr); });
    }
    void Evaluator::rotate_internal(Ciphertext &encrypted, int steps, const GaloisKeys &galois_keys, MemoryPoolHandle pool) const
    {
        auto context_data_ptr = context_.get_context_data(encrypted.parms_id());
        if (!context_data_ptr)
            throw std::invalid_argument("encrypted is not valid for encryption parameters");
        if (!context_data_ptr->qualifiers().using_batching)
            throw std::logic_error("encryption parameters do not support batching");
        if (galois_keys.parms_id() != context_.key_parms_id())
            throw std::invalid_argument("galois_keys is not valid for encryption parameters");
        if (!pool)
            throw std::invalid_argument("pool is uninitialized");
        if (steps == 0)
            return;
        Dev &d = device_for(context_);
        void *keys = device_keys(d, context_, galois_keys);
        const bool ckks = context_data_ptr->parms().scheme() == scheme_type::ckks;
        unary(context_, encrypted, [keys, steps, ckks](Dev &dd, void *a) {
            return ckks ? Evaluator_RotateVector(dd.ev, a, steps, keys, a, nullptr) : Evaluator_RotateRows(dd.ev, a, steps, keys, a, nullptr);
        });
    }
} // namespace seal

// ---- the reference's allocator hook (SEAL_MALLOC / SEAL_FREE, native/src/seal/util/defines.h:170-179; used by
// util/mempool.cpp:43,91 only).  integration/config_hip/seal/util/config.h points the two macros here when mempool.cpp is
// compiled for the drop-in: pool chunks become page-aligned (every ciphertext buffer is then whole pages: no unaligned head
// or tail to keep current) and a chunk that goes back to the system takes its mirrors with it.
namespace
{
    struct Chunks
    {
        std::mutex mu;
        std::map<std::uintptr_t, std::size_t> sizes; // page-aligned chunks handed to the pool
    };
    Chunks &chunks()
    {
        static Chunks *c = new Chunks; // never destroyed: the reference's global pool frees its chunks during static destruction
        return *c;
    }
} // namespace
extern "C" void *sealhip_host_alloc(std::size_t size)
{
    const std::size_t page = (std::size_t)sysconf(_SC_PAGESIZE);
    if (size < 4 * page)
        return std::malloc(size ? size : 1);
    const std::size_t rounded = (size + page - 1) / page * page;
    void *p = std::aligned_alloc(page, rounded);
    if (!p)
        throw std::bad_alloc();
    Chunks &c = chunks();
    std::lock_guard<std::mutex> g(c.mu);
    c.sizes[reinterpret_cast<std::uintptr_t>(p)] = rounded;
    return p;
}
extern "C" void sealhip_host_free(void *ptr)
{
    if (!ptr)
        return;
    std::size_t bytes = 0;
    {
        Chunks &c = chunks();
        std::lock_guard<std::mutex> g(c.mu);
        auto it = c.sizes.find(reinterpret_cast<std::uintptr_t>(ptr));
        if (it != c.sizes.end())
        {
            bytes = it->second;
            c.sizes.erase(it);
        }
    }
    if (bytes)
        forget_range(reinterpret_cast<std::uintptr_t>(ptr), reinterpret_cast<std::uintptr_t>(ptr) + bytes);
    std::free(ptr);
}
// counters for tests and tools: uploads, downloads, operands found on the device, faults served
// resident mode on / off at run time (off: every shadow is settled first).  Returns the previous setting.
extern "C" int sealhip_dropin_set_resident(int on)
{
    Mirrors &mm = mirrors();
    std::lock_guard<std::recursive_mutex> g(mm.mu);
    const int was = mm.eager ? 0 : 1;
    if (on)
        enable_resident(mm);
    else if (!mm.eager)
    {
        resolve_range(0, ~(std::uintptr_t)0);
        disable_resident(mm);
    }
    return was;
}
// resident mode: make the host copies of [ptr, ptr + bytes) current and unprotected (bytes == 0: of everything).  What a host
// calls before it hands Ciphertext::data() to anything that is not an ordinary load or store of its own threads: write(2) /
// send(2), another device's copy engine, a thread that must not take a signal.
extern "C" void sealhip_dropin_settle(const void *ptr, std::size_t bytes)
{
    if (!bytes)
        resolve_range(0, ~(std::uintptr_t)0);
    else
        resolve_range(reinterpret_cast<std::uintptr_t>(ptr), reinterpret_cast<std::uintptr_t>(ptr) + bytes);
}
extern "C" void sealhip_dropin_stats(uint64_t *uploads, uint64_t *downloads, uint64_t *reused, uint64_t *faults)
{
    Mirrors &mm = mirrors();
    std::lock_guard<std::recursive_mutex> g(mm.mu);
    if (uploads)
        *uploads = mm.uploads;
    if (downloads)
        *downloads = mm.downloads;
    if (reused)
        *reused = mm.reused;
    if (faults)
        *faults = mm.faults;
}
