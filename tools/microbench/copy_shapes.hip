// How fast can gfx950 copy?  (the two passes of the 2^16 NTT are copies with arithmetic in between)
// Variants: bytes per lane per access (8 / 16), loads in flight per thread (4 / 8 / 16), grid (one tile per workgroup vs a
// persistent grid-stride loop), nontemporal hints, in place vs out of place.  4 GiB moved each way, far beyond the caches.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));

template <typename T, int U, bool NT>
__global__ void __launch_bounds__(256) k_copy(const T *in, T *out, size_t n /* elements */)
{
    // grid-stride over tiles of 256*U elements; U loads in flight per thread, then U stores
    const size_t tiles = n / (256 * U);
    for (size_t t = blockIdx.x; t < tiles; t += gridDim.x)
    {
        const T *p = in + t * 256 * U + threadIdx.x;
        T *q = out + t * 256 * U + threadIdx.x;
        T v[U];
#pragma unroll
        for (int e = 0; e < U; e++)
            v[e] = NT ? __builtin_nontemporal_load(p + e * 256) : p[e * 256];
#pragma unroll
        for (int e = 0; e < U; e++)
        {
            if (NT)
                __builtin_nontemporal_store(v[e], q + e * 256);
            else
                q[e * 256] = v[e];
        }
    }
}

template <typename T, int U, bool NT>
int run(const char *name, const void *in, void *out, size_t bytes, unsigned grid)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const size_t n = bytes / sizeof(T);
    const unsigned g = grid ? grid : (unsigned)(n / (256 * U));
    hipLaunchKernelGGL((k_copy<T, U, NT>), dim3(g), dim3(256), 0, 0, (const T *)in, (T *)out, n);
    CK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int r = 0; r < 5; r++)
    {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_copy<T, U, NT>), dim3(g), dim3(256), 0, 0, (const T *)in, (T *)out, n);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best)
            best = ms;
    }
    printf("%-64s grid %8u  %7.3f ms  %7.1f GB/s (read + write)\n", name, g, best, 2.0 * bytes / (best * 1e-3) / 1e9);
    return 0;
}

int main()
{
    const size_t bytes = size_t(4) << 30;
    void *a, *b;
    CK(hipMalloc(&a, bytes));
    CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 1, bytes));
    CK(hipMemset(b, 2, bytes));
    for (unsigned grid : { 0u, 2048u, 4096u, 8192u })
    {
        run<uint64_t, 16, false>("8 B/lane, 16 in flight (the NTT passes' shape)", a, b, bytes, grid);
        run<uint64_t, 8, false>("8 B/lane, 8 in flight", a, b, bytes, grid);
        run<u64x2, 8, false>("16 B/lane, 8 in flight", a, b, bytes, grid);
        run<u64x2, 4, false>("16 B/lane, 4 in flight", a, b, bytes, grid);
        run<u64x2, 16, false>("16 B/lane, 16 in flight", a, b, bytes, grid);
        run<u64x2, 8, true>("16 B/lane, 8 in flight, nontemporal", a, b, bytes, grid);
        run<uint64_t, 16, true>("8 B/lane, 16 in flight, nontemporal", a, b, bytes, grid);
    }
    run<u64x2, 8, false>("16 B/lane, 8 in flight, IN PLACE", a, a, bytes, 0);
    run<uint64_t, 16, false>("8 B/lane, 16 in flight, IN PLACE", a, a, bytes, 0);
    run<u64x2, 8, false>("16 B/lane, 8 in flight, IN PLACE", a, a, bytes, 4096);
    return 0;
}
