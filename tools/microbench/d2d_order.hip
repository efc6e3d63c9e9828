// Does the runtime order  hipMemcpy(H2D, pageable) -> hipMemcpyAsync(D2D) -> hipMemsetAsync -> kernel (in place) -> hipMemcpy(D2H, pageable)
// on ONE stream the way the library relied on until round 3 (Ciphertext::resize 2 -> 3 followed by the in-place 2 x 2 product)?
// This is the exact shape of the one wrong device result of round 3 (profiles/r03_fuzz_stress.txt: 256 consecutive words of the
// middle polynomial), stripped of the library: no pool, no modular arithmetic - the kernel writes three functions of its two inputs that
// the host can recompute, so a word that is wrong tells where it came from (the stale source word, zero, the previous iteration ...).
// usage: d2d_order [iterations] [mode]   mode 0 = copy+memset then in-place kernel (the old path), 1 = out-of-place kernel into a fresh
// slab (the round-3 path), 2 = copy by a kernel on the same stream instead of hipMemcpyAsync
// build: hipcc -O2 --offload-arch=gfx950 d2d_order.hip -o d2d_order
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                                              \
    do                                                                                     \
    {                                                                                      \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess)                                                              \
        {                                                                                  \
            std::fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
            std::exit(2);                                                                  \
        }                                                                                  \
    } while (0)

__host__ __device__ inline uint64_t f0(uint64_t a) { return a * a + 1; }
__host__ __device__ inline uint64_t f1(uint64_t a, uint64_t b) { return 2 * a * b + 3; }
__host__ __device__ inline uint64_t f2(uint64_t b) { return b * b + 5; }

__global__ void __launch_bounds__(256) tensor_kernel(const uint64_t *x, uint64_t *out, size_t pw)
{
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < pw; i += (size_t)gridDim.x * 256)
    {
        uint64_t a = x[i], b = x[pw + i];
        out[i] = f0(a);
        out[pw + i] = f1(a, b);
        out[2 * pw + i] = f2(b);
    }
}
__global__ void __launch_bounds__(256) copy_kernel(const uint64_t *src, uint64_t *dst, size_t words)
{
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < words; i += (size_t)gridDim.x * 256)
        dst[i] = src[i];
}
__global__ void __launch_bounds__(256) fill_kernel(uint64_t *dst, size_t words, uint64_t v)
{
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < words; i += (size_t)gridDim.x * 256)
        dst[i] = v;
}

int main(int argc, char **argv)
{
    const long iters = argc > 1 ? std::atol(argv[1]) : 20000;
    const int mode = argc > 2 ? std::atoi(argv[2]) : 0;
    const int use_stream = argc > 3 ? std::atoi(argv[3]) : 0;
    hipStream_t s = nullptr;
    if (use_stream)
        CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    // plane sizes of the fuzz's small contexts: batch (1..3) x K (2..3) x N (2048 / 4096 / 8192)
    const size_t planes[] = { 3 * 2 * 4096, 2 * 2 * 2048, 1 * 3 * 8192, 3 * 3 * 4096, 1 * 2 * 4096, 3 * 2 * 8192 };
    // a small rotating set of device blocks standing in for the pool: the same addresses come back with other contents
    const size_t maxw = 3 * 3 * 3 * 8192;
    std::vector<uint64_t *> blocks(6);
    for (auto &b : blocks)
        CK(hipMalloc(reinterpret_cast<void **>(&b), maxw * 8));
    long bad_iters = 0;
    uint64_t lcg = 88172645463325252ull;
    for (long it = 0; it < iters; it++)
    {
        const size_t pw = planes[it % 6];
        uint64_t *src = blocks[(it * 2) % 6], *dst = blocks[(it * 2 + 1 + (it / 6) % 2 * 2) % 6];
        if (src == dst)
            dst = blocks[(it * 2 + 1) % 6];
        // fresh pageable buffers every iteration, as numpy gives the harness
        std::vector<uint64_t> in(2 * pw), got(3 * pw);
        for (auto &w : in)
        {
            lcg ^= lcg << 13, lcg ^= lcg >> 7, lcg ^= lcg << 17;
            w = lcg;
        }
        // what the slab held before: a recognisable pattern
        fill_kernel<<<96, 256, 0, s>>>(dst, 3 * pw, 0xDEAD000000000000ull | (uint64_t)it);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(src, in.data(), 2 * pw * 8, hipMemcpyHostToDevice));
        const unsigned grid = (unsigned)((pw + 255) / 256 > 2048 ? 2048 : (pw + 255) / 256);
        if (mode == 0)
        {
            CK(hipMemcpyAsync(dst, src, 2 * pw * 8, hipMemcpyDeviceToDevice, s));
            CK(hipMemsetAsync(dst + 2 * pw, 0, pw * 8, s));
            tensor_kernel<<<grid, 256, 0, s>>>(dst, dst, pw);
        }
        else if (mode == 1)
            tensor_kernel<<<grid, 256, 0, s>>>(src, dst, pw);
        else
        {
            copy_kernel<<<grid, 256, 0, s>>>(src, dst, 2 * pw);
            fill_kernel<<<grid, 256, 0, s>>>(dst + 2 * pw, pw, 0);
            tensor_kernel<<<grid, 256, 0, s>>>(dst, dst, pw);
        }
        CK(hipGetLastError());
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(got.data(), dst, 3 * pw * 8, hipMemcpyDeviceToHost));
        size_t nbad = 0, first = 0, last = 0, eq_src = 0, eq_zero = 0, eq_fill = 0;
        for (size_t i = 0; i < pw; i++)
        {
            const uint64_t exp[3] = { f0(in[i]), f1(in[i], in[pw + i]), f2(in[pw + i]) };
            for (int p = 0; p < 3; p++)
            {
                const uint64_t g = got[p * pw + i];
                if (g == exp[p])
                    continue;
                if (!nbad)
                    first = p * pw + i;
                last = p * pw + i;
                nbad++;
                eq_src += p < 2 && g == in[p * pw + i];
                eq_zero += g == 0;
                eq_fill += (g >> 48) == 0xDEAD;
            }
        }
        if (nbad)
        {
            bad_iters++;
            std::vector<uint64_t> again(3 * pw);
            CK(hipMemcpy(again.data(), dst, 3 * pw * 8, hipMemcpyDeviceToHost));
            const bool same = std::memcmp(again.data(), got.data(), 3 * pw * 8) == 0;
            std::printf(
                "iteration %ld mode %d plane %zu words: %zu wrong words, first %zu (poly %zu word %zu) last %zu; %zu equal the copied source word, "
                "%zu are zero, %zu are the slab's previous contents; a second download %s\n",
                it, mode, pw, nbad, first, first / pw, first % pw, last, eq_src, eq_zero, eq_fill,
                same ? "shows the same words (the device memory is wrong)" : "differs (the transfer was wrong)");
            if (bad_iters > 20)
                break;
        }
    }
    std::printf("mode %d stream %s: %ld iterations, %ld with wrong words\n", mode, use_stream ? "non-blocking" : "NULL", iters, bad_iters);
    return bad_iters ? 1 : 0;
}
