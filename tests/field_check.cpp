// TEST ONLY: exactness and range checks of the double-precision residue arithmetic of
// seal_amd/csrc/field.h against 128-bit integer arithmetic (built with g++ against the emulator's
// stand-in hip_runtime.h; the same source the kernels include).
#include "field.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
using namespace sealhip;
typedef unsigned __int128 u128;
typedef __int128 i128;

static int fails = 0;
#define EXPECT(c, ...) do { if (!(c)) { if (fails < 10) { printf("FAIL %s:%d: ", __FILE__, __LINE__); printf(__VA_ARGS__); printf("\n"); } fails++; } } while (0)

static uint64_t canon(i128 v, uint64_t q)
{
    i128 r = v % (i128)q;
    if (r < 0) r += q;
    return (uint64_t)r;
}

int main()
{
    std::mt19937_64 rng(0x5EA1);
    // primes = 1 (mod 2^17) just under each bit size (primality is irrelevant to the arithmetic identities)
    const uint64_t qs[] = { (1ull << 50) - (1ull << 17) * 3 + 1, (1ull << 49) + (1ull << 17) + 1, (1ull << 40) + (1ull << 17) * 7 + 1,
                            (1ull << 30) - (1ull << 17) + 1, (1ull << 20) + 1, 1125899906826241ull, 1125899906629633ull };
    for (uint64_t q : qs)
    {
        FpDesc m{ (double)q, 1.0 / (double)q, (double)((1ull << 32) % q), q };
        const double qd = (double)q;
        for (int it = 0; it < 400000; it++)
        {
            // x: signed, magnitude up to 8q (the largest the kernels feed a multiplication); w in [0,q)
            uint64_t w = rng() % q;
            int64_t x;
            switch (it & 7)
            {
            case 0: x = (int64_t)(rng() % q); break;
            case 1: x = -(int64_t)(rng() % q); break;
            case 2: x = (int64_t)(8 * q - 1 - (rng() & 1023)); break;
            case 3: x = -(int64_t)(8 * q - 1 - (rng() & 1023)); break;
            case 4: x = (int64_t)(q / 2 + (rng() & 3)); break;
            case 5: w = q - 1 - (rng() & 3); x = (int64_t)(8 * q - 1); break;
            default: x = (int64_t)(rng() % (8 * q)) - (int64_t)(4 * q); break;
            }
            if ((u128)(x < 0 ? -x : x) >= ((u128)1 << 53))
                continue;
            double r = fp_mulmod((double)x, (double)w, m.q, m.qinv);
            EXPECT(r == std::floor(r), "mulmod result not an integer");
            EXPECT(canon((i128)r, q) == canon((i128)x * (i128)w, q), "mulmod residue q=%llu x=%lld w=%llu", (unsigned long long)q, (long long)x, (unsigned long long)w);
            double bound = qd * (0.5 + 3.0 * std::ldexp(std::fabs((double)x) * ((double)w / qd), -53)) + 2.0;
            EXPECT(std::fabs(r) <= bound, "mulmod magnitude %.1f > %.1f", std::fabs(r), bound);
            // fix(): any |v| < 2^53 -> |result| <= q/2 + small, same residue
            int64_t v = (int64_t)(rng() >> 11) * ((rng() & 1) ? 1 : -1);
            double f = fp_fix((double)v, m.q, m.qinv);
            EXPECT(canon((i128)f, q) == canon((i128)v, q), "fix residue");
            EXPECT(std::fabs(f) <= qd / 2 + qd * std::ldexp(1.0, -40) + 1.0, "fix magnitude");
            // conversions
            uint64_t c = rng() % q;
            EXPECT(fp_from_u52(c) == (double)c, "from_u52");
            EXPECT(fp_to_canon(f, m) == canon((i128)f, q), "to_canon");
            uint64_t any = rng();
            double a = fp_from_u64(any, m);
            EXPECT(canon((i128)a, q) == any % q, "from_u64 residue");
            EXPECT(std::fabs(a) < qd + 4294967296.0, "from_u64 magnitude");
        }
    }
    // growth of four forward / inverse stages from the documented starting bounds stays below 2^53
    {
        double B = 1.0;
        for (int s = 0; s < 4; s++) B = B + 0.5 + 0.375 * B;
        if (!(B < 8.0)) { printf("forward bound %.3f\n", B); fails++; }
        // balanced tables (|w| <= q/2): B -> 1.1875 B + 0.5, seven stages from a fixed value stay below 8 q
        B = 0.5;
        for (int s = 0; s < 7; s++) B = 1.1875 * B + 0.5;
        if (!(B < 8.0)) { printf("balanced forward bound %.3f\n", B); fails++; }
    }
    // balanced multiplier: |r| <= q (1/2 + 3/16 B) for |x| <= B q, |w| <= q/2 (the bound the lean key-switch placement uses)
    for (uint64_t q : qs)
    {
        const double qd = (double)q, qinv = 1.0 / qd;
        for (int it = 0; it < 200000; it++)
        {
            const double w = (double)(int64_t)(rng() % (q / 2 + 1)) * ((rng() & 1) ? 1.0 : -1.0);
            const double Bq = (it & 1) ? 7.88 * qd : (double)(rng() % (8 * q));
            const double x = std::floor(Bq) * ((rng() & 2) ? 1.0 : -1.0);
            if (std::fabs(x) >= 9007199254740992.0)
                continue;
            const double r = fp_mulmod(x, w, qd, qinv);
            EXPECT(canon((i128)r, q) == canon((i128)x * (i128)w, q), "balanced mulmod residue");
            EXPECT(std::fabs(r) <= qd * (0.5 + 0.1875 * std::fabs(x) / qd) + 2.0, "balanced mulmod magnitude %.1f (x %.1f q %.1f)", std::fabs(r), x, qd);
        }
    }
    // ---- integer back end (field.h, Field<false>): approximate Shoup quotient, fix4() / canon_any(), the unguarded butterflies
    //      and the compile-time bounds of IntBounds
    {
        typedef Field<false> F;
        const uint64_t iq[] = { (1ull << 60) - (1ull << 18) + 1, (1ull << 59) + (1ull << 17) + 1, 1152921504606830593ull /* SEAL's first 60-bit prime */,
                                (1ull << 58) + (1ull << 17) * 3 + 1, (1ull << 58) - (1ull << 17) * 3 + 1,
                                (1ull << 55) - (1ull << 17) * 5 + 1, (1ull << 50) + (1ull << 17) + 1, (1ull << 40) + (1ull << 17) * 7 + 1,
                                (1ull << 33) + (1ull << 17) + 1, (1ull << 30) - (1ull << 17) + 1, (1ull << 20) + 1, 786433ull, 65537ull };
        for (uint64_t q : iq)
        {
            const u128 ratio = (((u128)1 << 127) / q) * 2 + ((((u128)1 << 127) % q) * 2 >= q ? 1 : 0); // floor(2^128 / q)
            ModDesc md{ q, 2 * q, (uint64_t)ratio, (uint64_t)(ratio >> 64) };
            EXPECT((u128)md.ratio_hi * q <= (((u128)1) << 64) - 1 + q, "ratio");
            const F::Mod m = F::make_mod(md, FpDesc{});
            const bool tight = (q >> 58) != 0;           // class 0: a word holds 16 q
            const bool hi32 = (q >> 40) != 0;            // the high-word estimates need q >= 2^40
            const u128 lim = tight ? (u128)16 * q : (u128)64 * q;
            EXPECT(lim <= ((u128)1 << 64), "class limit fits a word");
            for (int it = 0; it < 300000; it++)
            {
                uint64_t x;
                switch (it % 6)
                {
                case 0: x = (uint64_t)(lim - 1 - (rng() & 1023)); break;          // just below the limit
                case 1: x = rng() % (4 * q); break;
                case 2: x = 2 * q * (1 + rng() % 7) - (rng() & 3); break;         // around the multiples of 2q
                case 3: x = 2 * q * (rng() % 8) + (rng() & 3); break;
                case 4: x = (uint64_t)(rng() & 0xffffffffull); break;
                default: x = (uint64_t)((u128)rng() % lim); break;
                }
                if ((u128)x >= lim)
                    continue;
                uint64_t y = x;
                F::fix4<false>(y, m);
                EXPECT(y < 4 * q, "fix4 range q=%llu x=%llu -> %llu", (unsigned long long)q, (unsigned long long)x, (unsigned long long)y);
                EXPECT(y % q == x % q, "fix4 residue q=%llu x=%llu", (unsigned long long)q, (unsigned long long)x);
                EXPECT(F::canon_any<false>(x, m) == x % q, "canon_any q=%llu x=%llu", (unsigned long long)q, (unsigned long long)x);
                if (hi32)
                {
                    y = x;
                    F::fix4<true>(y, m);
                    EXPECT(y < 4 * q && y % q == x % q, "fix4<hi> q=%llu x=%llu -> %llu", (unsigned long long)q, (unsigned long long)x, (unsigned long long)y);
                    EXPECT(F::canon_any<true>(x, m) == x % q, "canon_any<hi> q=%llu x=%llu", (unsigned long long)q, (unsigned long long)x);
                }
                // the product: ANY 64-bit x, result below 4 q with the approximate quotient, below 2 q with the exact one
                const uint64_t w = (it & 8) ? q - 1 - (rng() & 3) : rng() % q;
                const ShoupOp tw{ w, (uint64_t)((((u128)w) << 64) / q) };
                const uint64_t xx = (it & 16) ? ~(uint64_t)0 - (rng() & 1023) : (it & 32) ? rng() : x;
                const uint64_t hq = F::mul_hi_approx(xx, tw), he = (uint64_t)(((u128)xx * tw.wq) >> 64);
                EXPECT(hq <= he && he - hq <= 2, "mul_hi_approx");
                const uint64_t r4 = F::mul_lazy4(xx, tw, m), r2 = F::mul_lazy(xx, tw, m);
                EXPECT(r4 < 4 * q && r4 % q == (uint64_t)(((u128)xx * w) % q), "mul_lazy4 q=%llu", (unsigned long long)q);
                EXPECT(r2 < 2 * q && r2 % q == r4 % q, "mul_lazy");
            }
            // forward stages under the schedule of IntBounds: values stay below the tracked bound, residues equal the guarded butterfly's
            for (int it = 0; it < 60000; it++)
            {
                uint64_t X = (it & 1) ? 4 * q - 1 - (rng() & 7) : rng() % (4 * q), Y = (it & 2) ? 4 * q - 1 - (rng() & 7) : rng() % (4 * q);
                uint64_t Xg = X, Yg = Y;
                int B = 4;
                for (int sgl = 0; sgl < 16; sgl++)
                {
                    const uint64_t w = (it & 4) ? q - 1 - (rng() & 1) : rng() % q;
                    const ShoupOp tw{ w, (uint64_t)((((u128)w) << 64) / q) };
                    const bool fixb = tight ? IntBounds<0>::fwd_fix_before(B) : IntBounds<1>::fwd_fix_before(B);
                    if (fixb)
                    {
                        F::fix4<false>(X, m);
                        F::fix4<false>(Y, m);
                    }
                    B = tight ? IntBounds<0>::fwd_after_stage(B) : IntBounds<1>::fwd_after_stage(B);
                    F::bfly_fwd(X, Y, tw, m);
                    F::bfly_fwd_guarded(Xg, Yg, tw, m);
                    EXPECT((u128)X < (u128)B * q && (u128)Y < (u128)B * q, "forward bound %d q=%llu", B, (unsigned long long)q);
                    EXPECT(X % q == Xg % q && Y % q == Yg % q, "forward residue");
                    std::swap(X, Y); // let both outputs take both roles
                    std::swap(Xg, Yg);
                }
            }
            // inverse butterflies: exponent E -> sum below 2^(E+1) q, product below 4 q, residues equal the guarded butterfly's
            for (int it = 0; it < 60000; it++)
            {
                const int E = (int)(rng() % (tight ? 4 : 6));
                const u128 c = (u128)q << E;
                uint64_t X = (uint64_t)((it & 1) ? c - 1 - (rng() & 7) : (u128)rng() % c), Y = (uint64_t)((it & 2) ? c - 1 - (rng() & 7) : (u128)rng() % c);
                uint64_t Xg = X % (2 * q), Yg = Y % (2 * q);
                const uint64_t w = (it & 4) ? q - 1 - (rng() & 1) : rng() % q;
                const ShoupOp tw{ w, (uint64_t)((((u128)w) << 64) / q) };
                F::bfly_inv_lazy(X, Y, tw, (uint64_t)c, m);
                F::bfly_inv(Xg, Yg, tw, m);
                EXPECT((u128)X < 2 * c && Y < 4 * q, "inverse bound");
                EXPECT(X % q == Xg % q && Y % q == Yg % q, "inverse residue");
            }
        }
        // the schedules themselves: no stage may pass its class's limit
        static_assert(IntBounds<0>::fwd_after(4, 3) == 16 && IntBounds<0>::fwd_after(4, 4) == 8, "tight: a fix4 after every third stage");
        static_assert(IntBounds<1>::fwd_after(4, 15) == 64 && IntBounds<1>::fwd_after(4, 16) == 8, "roomy: one fix4 in sixteen stages");
        static_assert(IntBounds<2>::fwd_after(4, 9) == 4, "wide: the guarded butterflies keep [0, 4q)");
        for (int r = 0; r < 16; r++)
            for (int idx = 0; idx <= 4; idx++)
                for (int e = 0; e <= 5; e++)
                {
                    if (e <= 4) { EXPECT(IntBounds<0>::inv_exp(r, idx, 0, e) <= IntBounds<0>::lim_exp, "tight inverse exponent"); }
                    EXPECT(IntBounds<1>::inv_exp(r, idx, 0, e) <= IntBounds<1>::lim_exp, "roomy inverse exponent");
                }
    }
    if (fails) { printf("%d failures\n", fails); return 1; }
    printf("field_check ok\n");
    return 0;
}
