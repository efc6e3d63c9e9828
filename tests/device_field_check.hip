// Device-side check of the integer back end's instruction sequences (field.h, Field<false>: the gfx950 instruction wrappers are
// only compiled for the device, so the host-side tests/field_check.cpp cannot see them).  Every thread draws its own inputs and
// compares with 128-bit arithmetic: mul_hi_approx / mul_lazy4 / mul_lazy with twiddles in VGPRs and in SGPRs, fix4, canon_any,
// the forward and inverse butterflies under IntBounds' schedules.  Test infrastructure: built by seal_amd/csrc/Makefile
// (target devcheck) into seal_amd/lib/, run by tests/test_gpu_parity.py::test_device_field_check.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "field.h"

using namespace sealhip;
typedef unsigned __int128 u128;
typedef Field<false> F;

struct Rng
{
    uint64_t s;
    __device__ uint64_t next()
    {
        s ^= s << 13;
        s ^= s >> 7;
        s ^= s << 17;
        return s * 0x2545F4914F6CDD1Dull;
    }
};

#define DEXPECT(cond, code)            \
    do                                 \
    {                                  \
        if (!(cond))                   \
        {                              \
            atomicAdd(&fails[0], 1u);  \
            atomicMax(&fails[1], code); \
        }                              \
    } while (0)

__device__ __forceinline__ ShoupOp make_tw(uint64_t w, uint64_t q)
{
    return ShoupOp{ w, (uint64_t)((((u128)w) << 64) / q) };
}

__global__ void check_kernel(const ModDesc *mods, const ShoupOp *utw, unsigned nmods, unsigned iters, unsigned *fails)
{
    const unsigned mi = blockIdx.y;
    const ModDesc md = ld_uniform_mod(&mods[mi]);
    const F::Mod m = F::make_mod(md, FpDesc{});
    const uint64_t q = m.q;
    const bool tight = (q >> 58) != 0, hi32 = (q >> 40) != 0;
    const u128 lim = tight ? (u128)16 * q : (u128)64 * q;
    Rng rng{ 0x9E3779B97F4A7C15ull * (blockIdx.x * blockDim.x + threadIdx.x + 1) + mi };
    for (unsigned it = 0; it < iters; it++)
    {
        // one wave-uniform twiddle per (block, iteration) through the scalar path, one per-lane twiddle through the vector path
        const ShoupOpU uw = ld_uniform(utw, (mi * 64 + ((blockIdx.x + it) & 63)));
        const uint64_t wv = (it & 8) ? q - 1 - (rng.next() & 3) : rng.next() % q;
        const ShoupOp vw = make_tw(wv, q);
        uint64_t x;
        switch (it % 6)
        {
        case 0: x = (uint64_t)(lim - 1 - (rng.next() & 1023)); break;
        case 1: x = rng.next() % (4 * q); break;
        case 2: x = 2 * q * (1 + rng.next() % 7) - (rng.next() & 3); break;
        case 3: x = 2 * q * (rng.next() % 8) + (rng.next() & 3); break;
        case 4: x = rng.next() & 1023; break;
        default: x = (uint64_t)((u128)rng.next() % lim); break;
        }
        if ((u128)x < lim)
        {
            uint64_t y = x;
            F::fix4<false>(y, m);
            DEXPECT(y < 4 * q && y % q == x % q, 1u);
            DEXPECT(F::canon_any<false>(x, m) == x % q, 2u);
            if (hi32)
            {
                y = x;
                F::fix4<true>(y, m);
                DEXPECT(y < 4 * q && y % q == x % q, 3u);
                DEXPECT(F::canon_any<true>(x, m) == x % q, 4u);
            }
        }
        const uint64_t xx = (it & 16) ? ~(uint64_t)0 - (rng.next() & 1023) : (it & 32) ? rng.next() : x;
        {
            const uint64_t hq = F::mul_hi_approx(xx, vw), he = (uint64_t)(((u128)xx * vw.wq) >> 64);
            DEXPECT(hq <= he && he - hq <= 2, 5u);
            const uint64_t r4 = F::mul_lazy4(xx, vw, m), r2 = F::mul_lazy(xx, vw, m);
            DEXPECT(r4 < 4 * q && r4 % q == (uint64_t)(((u128)xx * vw.w) % q), 6u);
            DEXPECT(r2 < 2 * q && r2 % q == r4 % q, 7u);
            const uint64_t hu = F::mul_hi_approx(xx, uw), heu = (uint64_t)(((u128)xx * uw.wq) >> 64);
            DEXPECT(hu <= heu && heu - hu <= 2, 8u);
            const uint64_t r4u = F::mul_lazy4(xx, uw, m);
            DEXPECT(r4u < 4 * q && r4u % q == (uint64_t)(((u128)xx * uw.w) % q), 9u);
        }
        // forward stages under IntBounds' schedule, alternating the two twiddle sources
        {
            uint64_t X = (it & 1) ? 4 * q - 1 - (rng.next() & 7) : rng.next() % (4 * q), Y = (it & 2) ? 4 * q - 1 - (rng.next() & 7) : rng.next() % (4 * q);
            uint64_t Xg = X, Yg = Y;
            int B = 4;
            for (int sgl = 0; sgl < 16; sgl++)
            {
                const bool fixb = tight ? IntBounds<0>::fwd_fix_before(B) : IntBounds<1>::fwd_fix_before(B);
                if (fixb)
                {
                    if (tight)
                    {
                        F::fix4<true>(X, m);
                        F::fix4<true>(Y, m);
                    }
                    else
                    {
                        F::fix4<false>(X, m);
                        F::fix4<false>(Y, m);
                    }
                }
                B = tight ? IntBounds<0>::fwd_after_stage(B) : IntBounds<1>::fwd_after_stage(B);
                if (sgl & 1)
                {
                    F::bfly_fwd(X, Y, uw, m);
                    F::bfly_fwd_guarded(Xg, Yg, uw, m);
                }
                else
                {
                    F::bfly_fwd(X, Y, vw, m);
                    F::bfly_fwd_guarded(Xg, Yg, vw, m);
                }
                DEXPECT((u128)X < (u128)B * q && (u128)Y < (u128)B * q, 10u);
                DEXPECT(X % q == Xg % q && Y % q == Yg % q, 11u);
                const uint64_t t = X; X = Y; Y = t;
                const uint64_t tg = Xg; Xg = Yg; Yg = tg;
            }
        }
        // inverse butterflies at every admissible exponent
        {
            const int E = (int)(rng.next() % (tight ? 4 : 6));
            const u128 c = (u128)q << E;
            uint64_t X = (uint64_t)((it & 1) ? c - 1 - (rng.next() & 7) : (u128)rng.next() % c), Y = (uint64_t)((it & 2) ? c - 1 - (rng.next() & 7) : (u128)rng.next() % c);
            uint64_t Xg = X % (2 * q), Yg = Y % (2 * q), Xu = X, Yu = Y, Xh = Xg, Yh = Yg;
            F::bfly_inv_lazy(X, Y, vw, (uint64_t)c, m);
            F::bfly_inv(Xg, Yg, vw, m);
            DEXPECT((u128)X < 2 * c && Y < 4 * q, 12u);
            DEXPECT(X % q == Xg % q && Y % q == Yg % q, 13u);
            F::bfly_inv_lazy(Xu, Yu, uw, (uint64_t)c, m);
            F::bfly_inv(Xh, Yh, uw, m);
            DEXPECT((u128)Xu < 2 * c && Yu < 4 * q, 14u);
            DEXPECT(Xu % q == Xh % q && Yu % q == Yh % q, 15u);
        }
    }
}

int main()
{
    const uint64_t iq[] = { (1ull << 60) - (1ull << 18) + 1, (1ull << 59) + (1ull << 17) + 1, 1152921504606830593ull,
                            (1ull << 58) + (1ull << 17) * 3 + 1, (1ull << 58) - (1ull << 17) * 3 + 1,
                            (1ull << 55) - (1ull << 17) * 5 + 1, (1ull << 50) + (1ull << 17) + 1, (1ull << 40) + (1ull << 17) * 7 + 1,
                            (1ull << 33) + (1ull << 17) + 1, (1ull << 30) - (1ull << 17) + 1, (1ull << 20) + 1, 786433ull, 65537ull };
    const unsigned nm = sizeof(iq) / sizeof(iq[0]);
    std::vector<ModDesc> mods;
    std::vector<ShoupOp> utw;
    uint64_t s = 88172645463325252ull;
    for (uint64_t q : iq)
    {
        const u128 ratio = (~(u128)0) / q;
        mods.push_back(ModDesc{ q, 2 * q, (uint64_t)ratio, (uint64_t)(ratio >> 64) });
        for (int i = 0; i < 64; i++)
        {
            s ^= s << 13; s ^= s >> 7; s ^= s << 17;
            const uint64_t w = i == 0 ? q - 1 : i == 1 ? 1 : s % q;
            utw.push_back(ShoupOp{ w, (uint64_t)((((u128)w) << 64) / q) });
        }
    }
    ModDesc *dm; ShoupOp *dt; unsigned *df;
    if (hipMalloc(&dm, mods.size() * sizeof(ModDesc)) != hipSuccess || hipMalloc(&dt, utw.size() * sizeof(ShoupOp)) != hipSuccess || hipMalloc(&df, 8) != hipSuccess)
    {
        printf("device_field_check: no device memory\n");
        return 2;
    }
    hipMemcpy(dm, mods.data(), mods.size() * sizeof(ModDesc), hipMemcpyHostToDevice);
    hipMemcpy(dt, utw.data(), utw.size() * sizeof(ShoupOp), hipMemcpyHostToDevice);
    hipMemset(df, 0, 8);
    hipLaunchKernelGGL(check_kernel, dim3(64, nm), dim3(256), 0, 0, dm, dt, nm, 384u, df);
    unsigned fails[2] = { 1, 0 };
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(fails, df, 8, hipMemcpyDeviceToHost) != hipSuccess)
    {
        printf("device_field_check: kernel failed: %s\n", hipGetErrorString(hipGetLastError()));
        return 2;
    }
    if (fails[0])
    {
        printf("device_field_check: %u failures, highest failing check %u\n", fails[0], fails[1]);
        return 1;
    }
    printf("device_field_check ok (%u primes x %u threads x 384 rounds)\n", nm, 64 * 256);
    return 0;
}
