// Dependent-issue latency of v_fma_f64 / v_mad_u64_u32 on gfx950: ITERS x 16 instructions per lane arranged
// as CH independent dependency chains, run with W waves per SIMD.  cycles/instruction/wave from wall time.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITERS = 4096;
template <int OP, int CH>
__global__ void __launch_bounds__(256) k(uint64_t *out, uint64_t seed)
{
    double d[16]; uint64_t a[16]; uint32_t c[16];
    double bd = 1.0000001 + (double)threadIdx.x * 1e-9;
    uint32_t b32 = (uint32_t)seed * (threadIdx.x + 1) | 1;
#pragma unroll
    for (int i = 0; i < 16; i++) { d[i] = 1.5 + i; a[i] = seed + i; c[i] = (uint32_t)seed + i; }
    for (int it = 0; it < ITERS; it++)
    {
#pragma unroll
        for (int r = 0; r < 16 / CH; r++)
#pragma unroll
            for (int i = 0; i < CH; i++)
            {
                if (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[i]) : "v"(bd));
                if (OP == 1) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[i]) : "v"(c[i]), "v"(b32) : "vcc");
                if (OP == 2) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(bd));
                if (OP == 3) asm volatile("v_add_u32 %0, %0, %1" : "+v"(c[i]) : "v"(b32));
                if (OP == 4) asm volatile("v_rndne_f64 %0, %0" : "+v"(d[i]));
            }
    }
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s ^= a[i] ^ c[i] ^ (uint64_t)__double_as_longlong(d[i]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP, int CH>
int run(const char *name, uint64_t *d_out, int cus, int waves_per_simd)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int blocks = cus * waves_per_simd;
    hipLaunchKernelGGL((k<OP, CH>), dim3(blocks), dim3(256), 0, 0, d_out, 12345ull);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<OP, CH>), dim3(blocks), dim3(256), 0, 0, d_out, 12345ull);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double insts_per_wave = (double)ITERS * 16;
    printf("%-14s chains=%2d waves/SIMD=%d: %7.3f ms  %6.2f ns per instruction per wave  (%5.2f ns per instruction per SIMD)\n", name, CH,
           waves_per_simd, ms, ms * 1e6 / insts_per_wave, ms * 1e6 / insts_per_wave / waves_per_simd);
    return 0;
}
int main()
{
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    uint64_t *d_out; CK(hipMalloc(&d_out, (size_t)cus * 8 * 256 * 8));
    for (int w = 1; w <= 4; w *= 2)
    {
        run<0, 1>("v_fma_f64", d_out, cus, w); run<0, 2>("v_fma_f64", d_out, cus, w); run<0, 4>("v_fma_f64", d_out, cus, w); run<0, 8>("v_fma_f64", d_out, cus, w); run<0, 16>("v_fma_f64", d_out, cus, w);
        run<2, 1>("v_add_f64", d_out, cus, w); run<2, 4>("v_add_f64", d_out, cus, w);
        run<4, 1>("v_rndne_f64", d_out, cus, w); run<4, 4>("v_rndne_f64", d_out, cus, w);
        run<1, 1>("v_mad_u64_u32", d_out, cus, w); run<1, 2>("v_mad_u64_u32", d_out, cus, w); run<1, 4>("v_mad_u64_u32", d_out, cus, w); run<1, 16>("v_mad_u64_u32", d_out, cus, w);
        run<3, 1>("v_add_u32", d_out, cus, w); run<3, 4>("v_add_u32", d_out, cus, w); run<3, 16>("v_add_u32", d_out, cus, w);
    }
    return 0;
}
