// Achievable HBM write (and read) bandwidth on gfx950 for the access shapes of the two-pass NTT:
// each wave64 instruction moves 512 B as S segments of 512/S bytes that are `stride` bytes apart.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// tile of 4096 words (32 KiB) per 256-thread workgroup, 16 words per thread.
// mode 0: word index = k*256 + tid                    (2 KiB contiguous per instruction: pass-2 shape)
// mode 1: word = ((tid>>4)*16 + k)*256... ks1 shape: for register k: lanes c=tid&15 contiguous, ra=tid>>4 -> +4096 words
//         inside a 16-tile group (tile g owns columns), i.e. word = (ra*16 + g)*256 + k*16 + c  within a 65536-word transform
// mode 2: 32-lane segments: word = ((tid>>5)*8 + g8)*512 ... (256-byte segments)
template <int MODE, bool WRITE>
__global__ void __launch_bounds__(256) k(uint64_t *buf, size_t transforms)
{
    const unsigned tid = threadIdx.x;
    const size_t tile = blockIdx.x; // 16 tiles per 65536-word transform
    uint64_t *t = buf + (tile >> 4) * 65536;
    const unsigned g = tile & 15;
    uint64_t acc = 0;
#pragma unroll
    for (int kk = 0; kk < 16; kk++)
    {
        size_t w;
        if (MODE == 0)
            w = (size_t)g * 4096 + kk * 256 + tid;
        else if (MODE == 1)
            w = ((size_t)(tid >> 4) * 16 + g) * 256 + kk * 16 + (tid & 15);
        else
            w = ((size_t)(tid >> 5) * 16 + g) * 512 + kk * 32 + (tid & 31); // 8 row groups x 16 tiles x 512 words
        if (WRITE)
            t[w] = tile * 131 + kk + tid;
        else
            acc += t[w];
    }
    if (!WRITE && acc == 0x123456789ull)
        buf[0] = acc;
}
template <int MODE, bool WRITE>
int run(const char *name, uint64_t *buf, size_t transforms)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<MODE, WRITE>), dim3(transforms * 16), dim3(256), 0, 0, buf, transforms);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 5; r++)
        hipLaunchKernelGGL((k<MODE, WRITE>), dim3(transforms * 16), dim3(256), 0, 0, buf, transforms);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    printf("%-46s %8.3f ms  %8.1f GB/s\n", name, ms, transforms * 65536.0 * 8 / (ms * 1e-3) / 1e9);
    return 0;
}
int main()
{
    size_t transforms = 3360; // 1.64 GiB, the intermediate of one key-switch step at batch 16
    uint64_t *buf; CK(hipMalloc(&buf, transforms * 65536 * 8));
    CK(hipMemset(buf, 1, transforms * 65536 * 8));
    run<0, true>("write 2 KiB contiguous per WG instruction", buf, transforms);
    run<1, true>("write 4 x 128 B per wave instruction (ks1)", buf, transforms);
    run<2, true>("write 2 x 256 B per wave instruction", buf, transforms);
    run<0, false>("read  2 KiB contiguous per WG instruction", buf, transforms);
    run<1, false>("read  4 x 128 B per wave instruction", buf, transforms);
    run<2, false>("read  2 x 256 B per wave instruction", buf, transforms);
    return 0;
}
