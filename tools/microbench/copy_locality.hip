// Is the plain copy's 6.3 TB/s a matter of WHICH workgroup (hence which XCD: workgroup i runs on XCD i mod 8) touches which
// 4 KiB chunk?  The same kernel - one 16-byte element per thread, one 4 KiB chunk per workgroup - with the chunk index permuted.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));

// MODE 0: chunk = block; 1: block + shift (mod tiles); 2: XCD-partitioned (XCD x streams the x-th eighth of the buffer);
// 3: chunk = block ^ shift; 4: groups of `shift` consecutive chunks per XCD: chunk = (block / (8*shift)) * 8*shift + (block % 8) * shift + (block / 8) % shift
template <int MODE>
__global__ void __launch_bounds__(256) k_copy(const u64x2 *in, u64x2 *out, size_t tiles, unsigned shift)
{
    size_t b = blockIdx.x, c;
    if (MODE == 0)
        c = b;
    else if (MODE == 1)
        c = (b + shift) % tiles;
    else if (MODE == 2)
        c = (b & 7) * (tiles >> 3) + (b >> 3);
    else if (MODE == 3)
        c = b ^ shift;
    else
    {
        const size_t g = (size_t)8 * shift;
        c = (b / g) * g + (b & 7) * shift + ((b >> 3) % shift);
    }
    const size_t i = c * 256 + threadIdx.x;
    out[i] = in[i];
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
#define RUN(MODE, shift, label) timed(label, bytes, [&] { hipLaunchKernelGGL((k_copy<MODE>), dim3((unsigned)tiles), dim3(256), 0, 0, (const u64x2 *)a, (u64x2 *)b, tiles, (unsigned)(shift)); })

int main()
{
    const size_t bytes = size_t(4) << 30;
    void *a, *b;
    CK(hipMalloc(&a, bytes));
    CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 1, bytes));
    CK(hipMemset(b, 2, bytes));
    const size_t tiles = bytes / 4096;
    RUN(0, 0, "chunk = workgroup");
    RUN(1, 1, "chunk = workgroup + 1");
    RUN(1, 2, "chunk = workgroup + 2");
    RUN(1, 4, "chunk = workgroup + 4");
    RUN(1, 7, "chunk = workgroup + 7");
    RUN(3, 1, "chunk = workgroup ^ 1");
    RUN(3, 7, "chunk = workgroup ^ 7");
    RUN(2, 0, "XCD x streams the x-th eighth of the buffer");
    RUN(4, 2, "each XCD takes 2 consecutive chunks (8 KiB runs)");
    RUN(4, 4, "each XCD takes 4 consecutive chunks (16 KiB runs)");
    RUN(4, 8, "each XCD takes 8 consecutive chunks (32 KiB runs)");
    RUN(4, 64, "each XCD takes 64 consecutive chunks (256 KiB runs)");
    RUN(4, 512, "each XCD takes 512 consecutive chunks (2 MiB runs)");
    return 0;
}
