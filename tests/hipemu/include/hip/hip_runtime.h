// TEST INFRASTRUCTURE ONLY.  A stand-in <hip/hip_runtime.h> that lets g++ compile the
// product's HIP sources (seal_amd/csrc/*.hip, unchanged, no #ifdefs in them) into
// tests/hipemu/libsealhip_emu.so, where every kernel launch is executed on the CPU by
// running each GPU thread as a ucontext fiber, one workgroup at a time:
//   * __syncthreads()      -> the fiber yields until every live fiber of the block arrived
//   * __shared__ / HIP_DYNAMIC_SHARED -> one static buffer (blocks run serially)
//   * __shfl* (wave64)     -> lane exchange through a per-wave mailbox
// Purpose: debug kernel index arithmetic and host orchestration in the GPU-less build
// container (a gpurun round trip costs minutes and the budget is 90 GPU-minutes).
// This is NOT a CPU fallback of the product: seal_amd/ never loads the emulated library,
// `-m gpu` tests, smoke() and bench.py use only the gfx950 build, and nothing here is
// used to claim parity or performance.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <functional>

#define __host__
#define __device__
#define __global__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static
#define __restrict__ __restrict
#define HIP_DYNAMIC_SHARED(type, var) type *var = reinterpret_cast<type *>(hipemu::dyn_lds());

struct dim3
{
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_)
    {}
};
struct alignas(16) double2
{
    double x, y;
};
// blocks and lanes run one at a time in the emulator: a plain read-modify-write is atomic
static inline unsigned long long atomicMax(unsigned long long *p, unsigned long long v)
{
    unsigned long long old = *p;
    if (v > old)
        *p = v;
    return old;
}
static inline unsigned atomicMax(unsigned *p, unsigned v)
{
    unsigned old = *p;
    if (v > old)
        *p = v;
    return old;
}
static inline unsigned atomicOr(unsigned *p, unsigned v)
{
    unsigned old = *p;
    *p = old | v;
    return old;
}
static inline int __popc(unsigned x)
{
    return __builtin_popcount(x);
}
static inline long long __double_as_longlong(double d)
{
    long long r;
    std::memcpy(&r, &d, 8);
    return r;
}
struct alignas(16) ulonglong2
{
    unsigned long long x, y;
};
struct uint3_emu
{
    unsigned x, y, z;
};
extern uint3_emu threadIdx, blockIdx;
extern dim3 blockDim, gridDim;

typedef int hipError_t;
typedef void *hipStream_t;
typedef struct hipEvent_emu *hipEvent_t;
enum
{
    hipSuccess = 0,
    hipErrorInvalidValue = 1,
    hipErrorOutOfMemory = 2,
    hipErrorNoDevice = 100
};
enum hipMemcpyKind
{
    hipMemcpyHostToHost = 0,
    hipMemcpyHostToDevice = 1,
    hipMemcpyDeviceToHost = 2,
    hipMemcpyDeviceToDevice = 3,
    hipMemcpyDefault = 4
};
struct hipDeviceProp_t
{
    char name[256];
    int multiProcessorCount;
    size_t totalGlobalMem;
};

namespace hipemu
{
    void *dyn_lds();
    void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()> &body);
    void syncthreads();
    uint64_t shfl_exchange(uint64_t value, int src_lane); // wave64 mailbox
    int lane_id();
} // namespace hipemu

// HIPEMU_TRACE=1 prints the name of every launched kernel to stderr (to check WHICH kernel a test exercised)
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    (std::getenv("HIPEMU_TRACE") ? (void)std::fprintf(stderr, "hipemu launch %s\n", #kernel) : (void)0, \
     hipemu::launch(dim3(grid), dim3(block), (shmem), [&]() { kernel(__VA_ARGS__); }))

static inline void __syncthreads()
{
    hipemu::syncthreads();
}
// A wavefront executes in lockstep on the GPU, so a compiler-level barrier is all the hardware
// needs between a wave-local LDS write and the reads of other lanes; fibers run one lane at a
// time, so here it has to be a real rendezvous (every kernel calls it in uniform control flow).
#define __builtin_amdgcn_wave_barrier() hipemu::syncthreads()
enum hipFuncAttribute
{
    hipFuncAttributeMaxDynamicSharedMemorySize = 8
};
template <typename Fn>
static inline int hipFuncSetAttribute(Fn, hipFuncAttribute, int)
{
    return 0;
}

template <typename T>
static inline T __shfl(T v, int src_lane, int width = 64)
{
    static_assert(sizeof(T) <= 8, "shfl emu: <= 8 bytes");
    int lane = hipemu::lane_id();
    int src = (lane & ~(width - 1)) | (src_lane & (width - 1));
    uint64_t raw = 0;
    std::memcpy(&raw, &v, sizeof(T));
    raw = hipemu::shfl_exchange(raw, src);
    T out;
    std::memcpy(&out, &raw, sizeof(T));
    return out;
}
template <typename T>
static inline T __shfl_xor(T v, int mask, int width = 64)
{
    return __shfl(v, hipemu::lane_id() ^ mask, width);
}

static inline uint64_t __umul64hi(uint64_t a, uint64_t b)
{
    return (uint64_t)(((unsigned __int128)a * b) >> 64);
}
static inline uint32_t __umulhi(uint32_t a, uint32_t b)
{
    return (uint32_t)(((uint64_t)a * b) >> 32);
}
static inline uint32_t __brev(uint32_t x)
{
    uint32_t r = 0;
    for (int i = 0; i < 32; i++)
        r |= ((x >> i) & 1u) << (31 - i);
    return r;
}

// ---- runtime API subset (device memory == host memory) ----
static inline hipError_t hipMalloc(void **p, size_t bytes)
{
    *p = std::malloc(bytes ? bytes : 1);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
template <typename T>
static inline hipError_t hipMalloc(T **p, size_t bytes)
{
    return hipMalloc(reinterpret_cast<void **>(p), bytes);
}
enum { hipHostMallocDefault = 0 };
static inline hipError_t hipHostMalloc(void **p, size_t bytes, unsigned = 0)
{
    *p = malloc(bytes);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
static inline hipError_t hipHostFree(void *p)
{
    free(p);
    return hipSuccess;
}
static inline hipError_t hipFree(void *p)
{
    std::free(p);
    return hipSuccess;
}
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind)
{
    std::memmove(d, s, n);
    return hipSuccess;
}
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t)
{
    std::memmove(d, s, n);
    return hipSuccess;
}
static inline hipError_t hipMemset(void *d, int v, size_t n)
{
    std::memset(d, v, n);
    return hipSuccess;
}
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t)
{
    std::memset(d, v, n);
    return hipSuccess;
}
static inline hipError_t hipDeviceSynchronize()
{
    return hipSuccess;
}
static inline hipError_t hipStreamSynchronize(hipStream_t)
{
    return hipSuccess;
}
static inline hipError_t hipGetLastError()
{
    return hipSuccess;
}
static inline hipError_t hipGetDeviceCount(int *n)
{
    *n = 1;
    return hipSuccess;
}
static inline hipError_t hipGetDevice(int *d)
{
    *d = 0;
    return hipSuccess;
}
static inline hipError_t hipSetDevice(int)
{
    return hipSuccess;
}
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int)
{
    std::memset(p, 0, sizeof(*p));
    std::strcpy(p->name, "hipemu (CPU fibers, test only)");
    p->multiProcessorCount = 4;
    p->totalGlobalMem = size_t(8) << 30;
    return hipSuccess;
}
static inline const char *hipGetErrorString(hipError_t e)
{
    return e == hipSuccess ? "hipSuccess" : "hipemu error";
}
static inline hipError_t hipEventCreate(hipEvent_t *e)
{
    *e = nullptr;
    return hipSuccess;
}
// streams are synchronous in the emulator: creation hands out a dummy handle, waits are no-ops
enum { hipStreamDefault = 0, hipStreamNonBlocking = 1, hipEventDisableTiming = 2 };
static inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned)
{
    *s = nullptr;
    return hipSuccess;
}
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned)
{
    *e = nullptr;
    return hipSuccess;
}
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned)
{
    return hipSuccess;
}
static inline hipError_t hipEventDestroy(hipEvent_t)
{
    return hipSuccess;
}
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t = nullptr)
{
    return hipSuccess;
}
static inline hipError_t hipEventSynchronize(hipEvent_t)
{
    return hipSuccess;
}
// stream capture / graphs: the emulator executes eagerly, so "capturing" runs the work once and a graph cannot be replayed
typedef struct hipGraph_emu *hipGraph_t;
typedef struct hipGraphExec_emu *hipGraphExec_t;
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal = 0, hipStreamCaptureModeThreadLocal = 1, hipStreamCaptureModeRelaxed = 2 };
static inline hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode)
{
    return hipSuccess;
}
static inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t *g)
{
    *g = nullptr;
    return hipSuccess;
}
static inline hipError_t hipGraphInstantiate(hipGraphExec_t *e, hipGraph_t, void *, void *, size_t)
{
    *e = nullptr;
    return hipSuccess;
}
static inline hipError_t hipGraphDestroy(hipGraph_t)
{
    return hipSuccess;
}
static inline hipError_t hipGraphExecDestroy(hipGraphExec_t)
{
    return hipSuccess;
}
static inline hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t)
{
    return hipErrorInvalidValue; // nothing was recorded
}
static inline hipError_t hipStreamDestroy(hipStream_t)
{
    return hipSuccess;
}
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t)
{
    *ms = 0.f;
    return hipSuccess;
}
