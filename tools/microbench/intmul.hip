// Micro-benchmark: 32/64-bit integer multiply issue rates on gfx950, candidate
// 64-bit Shoup modular multiplies, and a streaming copy (practical HBM roofline).
// Build: hipcc --offload-arch=gfx950 -O3 -o intmul intmul.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 4096;
constexpr int UNROLL = 8;

template <int OP>
__global__ void __launch_bounds__(256) k_op(uint64_t *out, uint64_t seed)
{
    uint64_t a[UNROLL];
    uint64_t b = seed * (threadIdx.x + 1) + 0x9E3779B97F4A7C15ull;
    uint64_t q = (seed | 1) + 0x0FFFFFFFFFFFC001ull * 0 + 1152921504606830593ull; // 60-bit prime-ish
#pragma unroll
    for (int i = 0; i < UNROLL; i++) a[i] = b + i * 0x1234567ull;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < UNROLL; i++) {
            if (OP == 0) { // v_mul_lo_u32
                uint32_t x = (uint32_t)a[i]; x = x * (uint32_t)b + 1u; a[i] = x;
            } else if (OP == 1) { // v_mul_hi_u32
                uint32_t x = (uint32_t)a[i]; x = __umulhi(x, (uint32_t)b) + 0x80000001u; a[i] = x;
            } else if (OP == 2) { // v_mad_u64_u32
                a[i] = (uint64_t)(uint32_t)a[i] * (uint32_t)b + a[i];
            } else if (OP == 3) { // 64-bit add
                a[i] = a[i] + b;
            } else if (OP == 4) { // __umul64hi
                a[i] = __umul64hi(a[i], b) + b;
            } else if (OP == 5) { // 64x64 low
                a[i] = a[i] * b + 1;
            } else if (OP == 6) { // fp64 fma
                double d = __longlong_as_double(a[i]);
                d = __fma_rn(d, 1.0000001, 0.5);
                a[i] = __double_as_longlong(d);
            } else if (OP == 7) { // Shoup lazy mulmod: x*w - hi(x*w')*q
                uint64_t x = a[i];
                uint64_t hi = __umul64hi(x, b);
                a[i] = x * (b >> 4) - hi * q;
            } else if (OP == 8) { // Harvey CT butterfly on pairs (a[i], a[i^1])
                uint64_t x = a[i], y = a[i ^ 1];
                uint64_t two_q = q << 1;
                x = x >= two_q ? x - two_q : x;
                uint64_t hi = __umul64hi(y, b);
                uint64_t t = y * (b >> 4) - hi * q;
                a[i] = x + t;
            } else if (OP == 9) { // 32-bit add (baseline VALU rate)
                uint32_t x = (uint32_t)a[i]; x = x + (uint32_t)b; a[i] = x;
            } else if (OP == 10) { // v_mul_u32_u24
                uint32_t x = (uint32_t)a[i] & 0xffffff; x = __umul24(x, (uint32_t)b & 0xffffff) + 1; a[i] = x;
            }
        }
    }
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < UNROLL; i++) s ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) k_copy(const ulonglong2 *__restrict__ in, ulonglong2 *__restrict__ out, size_t n)
{
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = in[i];
}

template <int OP>
int run(const char *name, uint64_t *d_out, int blocks)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_op<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 12345ull);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_op<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 12345ull);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double ops = (double)blocks * 256 * ITERS * UNROLL;
    double gops = ops / (ms * 1e-3) / 1e9;
    // cycles per wave-instruction per SIMD at 2.4 GHz, 1024 SIMDs
    double wave_ops = ops / 64.0;
    double cyc = (ms * 1e-3) * 2.4e9 * 1024.0 / wave_ops;
    printf("%-28s %8.3f ms  %10.1f Gop/s  ~%6.2f cyc/wave-op/SIMD (at 2.4GHz)\n", name, ms, gops, cyc);
    return 0;
}

int main()
{
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s CUs=%d clock=%d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
    int blocks = prop.multiProcessorCount * 8;
    uint64_t *d_out; CK(hipMalloc(&d_out, (size_t)blocks * 256 * 8));
    run<9>("v_add_u32", d_out, blocks);
    run<0>("v_mul_lo_u32", d_out, blocks);
    run<1>("v_mul_hi_u32", d_out, blocks);
    run<10>("v_mul_u32_u24", d_out, blocks);
    run<2>("v_mad_u64_u32", d_out, blocks);
    run<3>("add_u64", d_out, blocks);
    run<4>("umul64hi", d_out, blocks);
    run<5>("mul64 lo (+1)", d_out, blocks);
    run<6>("v_fma_f64", d_out, blocks);
    run<7>("shoup mulmod lazy", d_out, blocks);
    run<8>("harvey half-butterfly", d_out, blocks);

    // streaming copy
    size_t bytes = (size_t)2 << 30; // 2 GiB in, 2 GiB out
    ulonglong2 *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 1, bytes));
    size_t n = bytes / 16;
    for (int rep = 0; rep < 2; rep++) {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_copy, dim3(256 * 8), dim3(256), 0, 0, a, b, n);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("copy 2GiB->2GiB: %.3f ms  %.1f GB/s (read+write)\n", ms, 2.0 * bytes / (ms * 1e-3) / 1e9);
    }
    return 0;
}
