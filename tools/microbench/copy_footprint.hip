// Where does the guide's 6.29 TB/s float4 copy come from, and can the NTT passes have it?  
// The same copy at footprints from 256 MiB to 8 GiB per direction, in three shapes: one float4 per thread (a plain streaming
// kernel: the launch IS the loop), 16 B per lane with 8 loads in flight per thread (grid-stride), 8 B per lane with 16 in
// flight (the shape of the transform passes).  The Infinity Cache is 256 MiB: a footprint of a few hundred MiB is partly
// served from it, one of several GiB is not.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));

__global__ void __launch_bounds__(256) k_one(const float4 *in, float4 *out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n)
        out[i] = in[i];
}
template <typename T, int U>
__global__ void __launch_bounds__(256) k_tiles(const T *in, T *out, size_t n)
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

template <class Launch>
int timed(const char *name, size_t bytes, Launch launch)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    launch();
    CK(hipDeviceSynchronize());
    float best = 1e9f, sum = 0;
    const int reps = 7;
    for (int r = 0; r < reps; r++)
    {
        CK(hipEventRecord(e0));
        launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
        sum += ms;
    }
    printf("  %-44s best %7.3f ms %7.1f GB/s   mean %7.1f GB/s\n", name, best, 2.0 * bytes / (best * 1e-3) / 1e9, 2.0 * bytes / (sum / reps * 1e-3) / 1e9);
    return 0;
}

int main()
{
    const size_t max_bytes = size_t(8) << 30;
    void *a, *b;
    CK(hipMalloc(&a, max_bytes));
    CK(hipMalloc(&b, max_bytes));
    CK(hipMemset(a, 1, max_bytes));
    CK(hipMemset(b, 2, max_bytes));
    for (size_t mib : { 256, 512, 1024, 2048, 4096, 8192 })
    {
        const size_t bytes = mib << 20;
        printf("footprint %5zu MiB per direction (read + write = 2x)\n", mib);
        const size_t n4 = bytes / 16;
        timed("one float4 per thread", bytes, [&] { hipLaunchKernelGGL(k_one, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, 0, (const float4 *)a, (float4 *)b, n4); });
        timed("16 B/lane x 8 in flight, one tile per workgroup", bytes, [&] { hipLaunchKernelGGL((k_tiles<u64x2, 8>), dim3((unsigned)(n4 / (256 * 8))), dim3(256), 0, 0, (const u64x2 *)a, (u64x2 *)b, n4); });
        timed("8 B/lane x 16 in flight, one tile per workgroup", bytes, [&] { hipLaunchKernelGGL((k_tiles<uint64_t, 16>), dim3((unsigned)(bytes / 8 / (256 * 16))), dim3(256), 0, 0, (const uint64_t *)a, (uint64_t *)b, bytes / 8); });
        timed("8 B/lane x 16 in flight, 4096 workgroups", bytes, [&] { hipLaunchKernelGGL((k_tiles<uint64_t, 16>), dim3(4096), dim3(256), 0, 0, (const uint64_t *)a, (uint64_t *)b, bytes / 8); });
    }
    return 0;
}
