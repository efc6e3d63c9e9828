// Throughput of the double-precision butterfly (field.h) as compiled by hipcc, registers only:
// each thread runs REPS radix-16 phases (4 stages, 32 butterflies, 16 fix()) on 16 doubles.
#include "../../seal_amd/csrc/field.h"
#include <cstdio>
using namespace sealhip;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int REPS = 512;
template <bool FP>
__global__ void __launch_bounds__(256) k(uint64_t *out, FpDesc fd, ModDesc md, double w0, ShoupOp sw)
{
    typedef Field<FP> F;
    typename F::Mod m = F::make_mod(md, fd);
    typename F::elem x[16];
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = F::from_canon((uint64_t)(threadIdx.x * 16 + i + 1) * 1234567ull % md.q, m);
    typename F::tw_t tw[16];
#pragma unroll
    for (int i = 0; i < 16; i++) { if constexpr (FP) tw[i] = w0 + i * 3.0 + threadIdx.x; else tw[i] = ShoupOp{ sw.w + i, sw.wq + i }; }
    for (int r = 0; r < REPS; r++)
    {
#pragma unroll
        for (int t = 0; t < 4; t++)
        {
            const int bit = 3 - t;
#pragma unroll
            for (int g = 0; g < (8 >> bit); g++)
#pragma unroll
                for (int kk = 0; kk < (1 << bit); kk++)
                {
                    const int e0 = (g << (bit + 1)) | kk;
                    F::bfly_fwd(x[e0], x[e0 | (1 << bit)], tw[(1 << t) + g], m);
                }
        }
#pragma unroll
        for (int i = 0; i < 16; i++) F::fix(x[i], m);
    }
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s ^= F::raw(x[i]);
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <bool FP>
int run(const char *name, uint64_t *out, int cus, int wg_per_cu)
{
    uint64_t q = FP ? 1125899906826241ull : 1152921504606748673ull;
    FpDesc fd{ (double)q, 1.0 / (double)q, 4294967296.0, q };
    ModDesc md{ q, 2 * q, 0, 0 };
    unsigned __int128 r = ((unsigned __int128)1 << 127) / q; r <<= 1; md.ratio_lo = (uint64_t)r; md.ratio_hi = (uint64_t)(r >> 64);
    ShoupOp sw{ 123456789012345ull, (uint64_t)((((unsigned __int128)123456789012345ull) << 64) / q) };
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int blocks = cus * wg_per_cu;
    hipLaunchKernelGGL(k<FP>, dim3(blocks), dim3(256), 0, 0, out, fd, md, 987654321.0, sw);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<FP>, dim3(blocks), dim3(256), 0, 0, out, fd, md, 987654321.0, sw);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double bfly_per_simd = (double)wg_per_cu * REPS * 32; // wave-butterflies per SIMD
    printf("%-6s waves/SIMD=%d: %7.3f ms  %6.2f ns per wave-butterfly per SIMD (incl. 0.5 fix per butterfly)\n", name, wg_per_cu, ms, ms * 1e6 / bfly_per_simd);
    return 0;
}
int main()
{
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    uint64_t *out; CK(hipMalloc(&out, (size_t)cus * 8 * 256 * 8));
    for (int w = 1; w <= 8; w *= 2) run<true>("fp64", out, cus, w);
    for (int w = 1; w <= 8; w *= 2) run<false>("int64", out, cus, w);
    return 0;
}
