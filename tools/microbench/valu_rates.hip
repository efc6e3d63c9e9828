// Micro-benchmark: issue rate of single VALU instructions on gfx950, written as inline asm so the
// compiler cannot fold, fuse or hoist them.  Each kernel runs ITERS x 16 independent instances of
// one instruction per lane; every CU gets 8 waves per SIMD (256-thread blocks x 8 per CU).
// Reports cycles per wave-instruction per SIMD using the measured shader clock (s_memtime delta).
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 2048;

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int OP>
__global__ void __launch_bounds__(256) k(uint64_t *out, uint64_t seed, unsigned long long *clk)
{
    uint64_t a[16];
    uint32_t c[16];
    double d[16];
    uint64_t b = seed * (threadIdx.x + 1) + 0x9E3779B97F4A7C15ull;
    uint32_t b32 = (uint32_t)b | 1;
    double bd = 1.0000001 + (double)threadIdx.x * 1e-9;
#pragma unroll
    for (int i = 0; i < 16; i++)
    {
        a[i] = b + i * 0x1234567ull;
        c[i] = (uint32_t)(b >> 7) + i;
        d[i] = 1.5 + i;
    }
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; it++)
    {
#define X(i)                                                                                                     \
    if (OP == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[i]) : "v"(c[i]), "v"(b32) : "vcc");   \
    if (OP == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(c[i]) : "v"(b32));                                \
    if (OP == 2) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(c[i]) : "v"(b32));                                \
    if (OP == 3) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[i]) : "v"(bd));                                \
    if (OP == 4) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(bd));                                    \
    if (OP == 5) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(bd));                                    \
    if (OP == 6) asm volatile("v_rndne_f64 %0, %0" : "+v"(d[i]));                                                \
    if (OP == 7) asm volatile("v_add_u32 %0, %0, %1" : "+v"(c[i]) : "v"(b32));                                   \
    if (OP == 8) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(c[i]) : "v"(b32) : "vcc");                   \
    if (OP == 9) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(c[i]) : "v"(b32) : "vcc");                  \
    if (OP == 10) asm volatile("v_cmp_ge_u64 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");                        \
    if (OP == 11) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(a[i]) : "v"(b));                            \
    if (OP == 12) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(c[i]) : "v"(b32));                              \
    if (OP == 13) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(c[i]) : "v"(b32));                          \
    if (OP == 14) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(d[i]) : "v"(c[i]));                                 \
    if (OP == 15) asm volatile("v_cvt_u32_f64 %0, %1" : "=v"(c[i]) : "v"(d[i]));                                 \
    if (OP == 16) asm volatile("v_cmp_ge_f64 vcc, %0, %1" : : "v"(d[i]), "v"(bd) : "vcc");                       \
    if (OP == 17) asm volatile("v_sub_co_u32 %0, vcc, %0, %1" : "+v"(c[i]) : "v"(b32) : "vcc");                  \
    if (OP == 18) asm volatile("v_subb_co_u32 %0, vcc, %0, %1, vcc" : "+v"(c[i]) : "v"(b32) : "vcc");            \
    if (OP == 19) asm volatile("v_min_u32 %0, %0, %1" : "+v"(c[i]) : "v"(b32));                                  \
    if (OP == 20) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(a[i]) : "v"(c[i]), "s"(0x7fffffe1u) : "vcc"); \
    if (OP == 21) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(c[i]) : "v"(b32));                               \
    if (OP == 22) asm volatile("v_floor_f64 %0, %0" : "+v"(d[i]));                                               \
    if (OP == 23) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(a[i]) : "v"(c[i]), "v"(b32) : "vcc");
        REP16(X)
#undef X
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++)
        s ^= a[i] ^ c[i] ^ (uint64_t)__double_as_longlong(d[i]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0)
        *clk = t1 - t0;
}

template <int OP>
int run(const char *name, uint64_t *d_out, unsigned long long *d_clk, int blocks)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 12345ull, d_clk);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 12345ull, d_clk);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long clk = 0;
    CK(hipMemcpy(&clk, d_clk, 8, hipMemcpyDeviceToHost));
    // one block's loop: 8 waves/SIMD resident (if blocks = 8/CU) x ITERS x 16 wave-instructions each
    double per_simd_wave_insts = 8.0 * ITERS * 16; // per SIMD when exactly 8 blocks/CU of 4 waves each -> 8 waves per SIMD
    double ops = (double)blocks * 256 * ITERS * 16;
    double wave_ops = ops / 64.0;
    double cyc_wall24 = (ms * 1e-3) * 2.4e9 * 1024.0 / wave_ops;
    // s_memtime / readcyclecounter ticks at a constant 100 MHz on gfx9; report raw too
    printf("%-26s %8.3f ms  %9.1f Gop/s  %6.2f cyc/wave-inst/SIMD @2.4GHz-equivalent   (block-0 counter delta %llu, %g insts/SIMD)\n",
           name, ms, ops / (ms * 1e-3) / 1e9, cyc_wall24, clk, per_simd_wave_insts);
    return 0;
}

int main()
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s CUs=%d clock=%d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
    int blocks = prop.multiProcessorCount * 8;
    uint64_t *d_out;
    unsigned long long *d_clk;
    CK(hipMalloc(&d_out, (size_t)blocks * 256 * 8));
    CK(hipMalloc(&d_clk, 8));
    run<7>("v_add_u32", d_out, d_clk, blocks);
    run<8>("v_add_co_u32", d_out, d_clk, blocks);
    run<17>("v_sub_co_u32", d_out, d_clk, blocks);
    run<18>("v_subb_co_u32", d_out, d_clk, blocks);
    run<9>("v_cndmask_b32", d_out, d_clk, blocks);
    run<19>("v_min_u32", d_out, d_clk, blocks);
    run<10>("v_cmp_ge_u64", d_out, d_clk, blocks);
    run<11>("v_lshl_add_u64", d_out, d_clk, blocks);
    run<21>("v_pk_add_u16", d_out, d_clk, blocks);
    run<1>("v_mul_lo_u32", d_out, d_clk, blocks);
    run<2>("v_mul_hi_u32", d_out, d_clk, blocks);
    run<12>("v_mul_u32_u24", d_out, d_clk, blocks);
    run<13>("v_mad_u32_u24", d_out, d_clk, blocks);
    run<0>("v_mad_u64_u32", d_out, d_clk, blocks);
    run<20>("v_mad_u64_u32 (sgpr,0)", d_out, d_clk, blocks);
    run<23>("v_mad_i64_i32", d_out, d_clk, blocks);
    run<3>("v_fma_f64", d_out, d_clk, blocks);
    run<4>("v_mul_f64", d_out, d_clk, blocks);
    run<5>("v_add_f64", d_out, d_clk, blocks);
    run<6>("v_rndne_f64", d_out, d_clk, blocks);
    run<22>("v_floor_f64", d_out, d_clk, blocks);
    run<14>("v_cvt_f64_u32", d_out, d_clk, blocks);
    run<15>("v_cvt_u32_f64", d_out, d_clk, blocks);
    run<16>("v_cmp_ge_f64", d_out, d_clk, blocks);
    return 0;
}
