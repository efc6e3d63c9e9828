// What does mprotect() on a host buffer cost next to a live HIP context?  (device-resident drop-in, integration/)
// The ROCm driver mirrors parts of the process address space (userptr registrations made by pageable hipMemcpy, SVM); a
// protection change there goes through MMU notifiers.  Measured: mprotect PROT_NONE / PROT_READ|WRITE on a 16 MiB buffer
//   a. before any HIP call,  b. after HIP initialisation,  c. after the buffer was the source / destination of a pageable hipMemcpy,
//   d. for a buffer that was copied through a pinned staging buffer instead (never registered).
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
static void measure(const char *what, void *p, size_t bytes)
{
    double t = 1e9, u = 1e9;
    for (int r = 0; r < 5; r++)
    {
        double a = now();
        mprotect(p, bytes, PROT_NONE);
        double b = now();
        mprotect(p, bytes, PROT_READ | PROT_WRITE);
        double c = now();
        if (b - a < t)
            t = b - a;
        if (c - b < u)
            u = c - b;
    }
    double worst = 0;
    for (int r = 0; r < 5; r++)
    {
        double a = now();
        mprotect(p, bytes, PROT_NONE);
        mprotect(p, bytes, PROT_READ | PROT_WRITE);
        double d = now() - a;
        if (d > worst)
            worst = d;
    }
    printf("%-70s protect %8.1f us  unprotect %8.1f us  (worst pair %8.1f us)\n", what, t * 1e6, u * 1e6, worst * 1e6);
}
int main()
{
    const size_t bytes = 16u << 20;
    void *a = aligned_alloc(4096, bytes), *b = aligned_alloc(4096, bytes), *c = aligned_alloc(4096, bytes);
    memset(a, 1, bytes);
    memset(b, 2, bytes);
    memset(c, 3, bytes);
    measure("a. no HIP context yet", a, bytes);
    void *d = nullptr;
    if (hipMalloc(&d, bytes) != hipSuccess)
        return 1;
    hipDeviceSynchronize();
    measure("b. HIP initialised, buffer never given to HIP", a, bytes);
    hipMemcpy(d, b, bytes, hipMemcpyHostToDevice);
    measure("c1. after pageable hipMemcpy H2D from the buffer", b, bytes);
    hipMemcpy(b, d, bytes, hipMemcpyDeviceToHost);
    measure("c2. after pageable hipMemcpy D2H into the buffer", b, bytes);
    measure("    (an untouched neighbour buffer at the same time)", a, bytes);
    void *pin = nullptr;
    hipHostMalloc(&pin, bytes, hipHostMallocDefault);
    memcpy(pin, c, bytes);
    hipMemcpy(d, pin, bytes, hipMemcpyHostToDevice);
    hipMemcpy(pin, d, bytes, hipMemcpyDeviceToHost);
    memcpy(c, pin, bytes);
    measure("d. buffer copied through a pinned staging buffer", c, bytes);
    // a kernel in flight while protecting?
    hipMemsetAsync(d, 0, bytes, nullptr);
    measure("e. with device work queued", a, bytes);
    hipDeviceSynchronize();
    double t0 = now();
    memcpy(pin, c, bytes);
    double t1 = now();
    hipMemcpy(d, pin, bytes, hipMemcpyHostToDevice);
    double t2 = now();
    hipMemcpy(d, b, bytes, hipMemcpyHostToDevice);
    double t3 = now();
    printf("16 MiB: host memcpy to pinned %.2f ms, pinned H2D %.2f ms, pageable H2D %.2f ms\n", (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3);
    return 0;
}
