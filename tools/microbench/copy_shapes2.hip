// What makes the plain float4 copy (6.2 TB/s) faster than the tile shapes of the transform passes (5.4 TB/s)?
// (profiles/r03_microbench_copy_footprint.txt)  One 4 GiB copy per variant; each isolates one property.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));

// U accesses of T per thread, tile of 256*U elements per workgroup iteration; STRIDE_IN_TILE: element e of thread t at
// e*256 + t (wave instruction = contiguous 64 elements) ; grid-stride when the grid is smaller than the tile count
template <typename T, int U, int WAVES>
__global__ void __launch_bounds__(256, WAVES) k_tiles(const T *in, T *out, size_t n)
{
    const size_t tiles = n / (256 * U);
    for (size_t t = blockIdx.x; t < tiles; t += gridDim.x)
    {
        const T *p = in + t * 256 * U + threadIdx.x;
        T *q = out + t * 256 * U + threadIdx.x;
        T v[U];
#pragma unroll
        for (int e = 0; e < U; e++)
            v[e] = p[e * 256];
#pragma unroll
        for (int e = 0; e < U; e++)
            q[e * 256] = v[e];
    }
}
// the same bytes per thread, but every load is followed by its store (no batch of U loads then U stores)
template <typename T, int U>
__global__ void __launch_bounds__(256) k_interleaved(const T *in, T *out, size_t n)
{
    const size_t tiles = n / (256 * U);
    for (size_t t = blockIdx.x; t < tiles; t += gridDim.x)
    {
        const T *p = in + t * 256 * U + threadIdx.x;
        T *q = out + t * 256 * U + threadIdx.x;
#pragma unroll
        for (int e = 0; e < U; e++)
        {
            T v = p[e * 256];
            asm volatile("" ::: "memory");
            q[e * 256] = v;
        }
    }
}
// software-pipelined persistent loop: tile t+1's loads are issued before tile t's stores (the transform passes' prefetch)
template <typename T, int U>
__global__ void __launch_bounds__(256) k_prefetch(const T *in, T *out, size_t n)
{
    const size_t tiles = n / (256 * U);
    size_t t = blockIdx.x;
    if (t >= tiles)
        return;
    T nxt[U];
#pragma unroll
    for (int e = 0; e < U; e++)
        nxt[e] = in[t * 256 * U + threadIdx.x + e * 256];
    for (; t < tiles; t += gridDim.x)
    {
        T v[U];
#pragma unroll
        for (int e = 0; e < U; e++)
            v[e] = nxt[e];
        if (t + gridDim.x < tiles)
        {
#pragma unroll
            for (int e = 0; e < U; e++)
                nxt[e] = in[(t + gridDim.x) * 256 * U + threadIdx.x + e * 256];
        }
        T *q = out + t * 256 * U + threadIdx.x;
#pragma unroll
        for (int e = 0; e < U; e++)
            q[e * 256] = v[e];
    }
}
// pass 1's read pattern: 16 rows at a stride of 2 KiB * RS, 16 lanes (128 B) contiguous per row
template <int U>
__global__ void __launch_bounds__(256) k_columns(const uint64_t *in, uint64_t *out, size_t n)
{
    // a "transform" of 65536 words = 256 rows x 256 columns; workgroup = 16 columns x 256 rows; thread (c = tid & 15, hi = tid >> 4)
    // reads rows hi + 16 e, e < 16, writes tile order (contiguous)
    const size_t tiles = n / 4096;
    for (size_t t = blockIdx.x; t < tiles; t += gridDim.x)
    {
        const size_t tr = t >> 4, cg = t & 15;
        const uint64_t *p = in + tr * 65536 + cg * 16 + (threadIdx.x & 15) + (size_t)(threadIdx.x >> 4) * 256;
        uint64_t v[16];
#pragma unroll
        for (int e = 0; e < 16; e++)
            v[e] = p[(size_t)e * 16 * 256];
        uint64_t *q = out + t * 4096 + threadIdx.x;
#pragma unroll
        for (int e = 0; e < 16; e++)
            q[e * 256] = v[e];
    }
}

template <class Launch>
int timed(const char *name, size_t bytes, Launch launch)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    launch();
    CK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int r = 0; r < 5; r++)
    {
        CK(hipEventRecord(e0));
        launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    printf("%-72s %7.3f ms %7.1f GB/s\n", name, best, 2.0 * bytes / (best * 1e-3) / 1e9);
    return 0;
}
#define L(kern, grid, T, n) [&] { hipLaunchKernelGGL(kern, dim3((unsigned)(grid)), dim3(256), 0, 0, (const T *)a, (T *)b, (size_t)(n)); }

int main()
{
    const size_t bytes = size_t(4) << 30;
    void *a, *b;
    CK(hipMalloc(&a, bytes));
    CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 1, bytes));
    CK(hipMemset(b, 2, bytes));
    const size_t n16 = bytes / 16, n8 = bytes / 8;
    timed("16 B x 1 per thread, one tile per workgroup (the plain copy)", bytes, L((k_tiles<u64x2, 1, 1>), n16 / 256, u64x2, n16));
    timed("16 B x 1 per thread, 4096 persistent workgroups", bytes, L((k_tiles<u64x2, 1, 1>), 4096, u64x2, n16));
    timed("16 B x 1 per thread, 16384 persistent workgroups", bytes, L((k_tiles<u64x2, 1, 1>), 16384, u64x2, n16));
    timed("16 B x 2 per thread, one tile per workgroup", bytes, L((k_tiles<u64x2, 2, 1>), n16 / 512, u64x2, n16));
    timed("16 B x 4 per thread, one tile per workgroup", bytes, L((k_tiles<u64x2, 4, 1>), n16 / 1024, u64x2, n16));
    timed("8 B x 1 per thread, one tile per workgroup", bytes, L((k_tiles<uint64_t, 1, 1>), n8 / 256, uint64_t, n8));
    timed("8 B x 2 per thread, one tile per workgroup", bytes, L((k_tiles<uint64_t, 2, 1>), n8 / 512, uint64_t, n8));
    timed("8 B x 4 per thread, one tile per workgroup", bytes, L((k_tiles<uint64_t, 4, 1>), n8 / 1024, uint64_t, n8));
    timed("8 B x 16 per thread, one tile per workgroup (pass shape)", bytes, L((k_tiles<uint64_t, 16, 1>), n8 / 4096, uint64_t, n8));
    timed("8 B x 16 per thread, every load followed by its store", bytes, L((k_interleaved<uint64_t, 16>), n8 / 4096, uint64_t, n8));
    timed("16 B x 8 per thread, every load followed by its store", bytes, L((k_interleaved<u64x2, 8>), n16 / 2048, u64x2, n16));
    timed("8 B x 16 per thread, 2048 persistent workgroups, next tile prefetched", bytes, L((k_prefetch<uint64_t, 16>), 2048, uint64_t, n8));
    timed("8 B x 16 per thread, 8192 persistent workgroups, next tile prefetched", bytes, L((k_prefetch<uint64_t, 16>), 8192, uint64_t, n8));
    timed("8 B x 16 per thread, column reads of pass 1 (128-B runs, 2 KiB apart)", bytes, L((k_columns<16>), n8 / 4096, uint64_t, n8));
    timed("8 B x 16 per thread, one tile per WG, at most 4 waves per SIMD", bytes, L((k_tiles<uint64_t, 16, 4>), n8 / 4096, uint64_t, n8));
    return 0;
}
