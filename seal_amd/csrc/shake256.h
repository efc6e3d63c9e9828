// SHAKE256 (FIPS 202) — the reference's alternative PRNG for seeded objects (Shake256PRNG::refill_buffer,
// native/src/seal/randomgen.cpp:216-227: buffer = SHAKE256(seed (64 bytes) || counter (8 bytes)), 4096 bytes out).
// Written from the standard: Keccak-f[1600], rate 136 bytes, domain suffix 0x1F.  Host side.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>

namespace sealhip
{
    namespace keccak
    {
        inline uint64_t rotl(uint64_t x, int c)
        {
            return c ? (x << c) | (x >> (64 - c)) : x;
        }
        inline void f1600(uint64_t (&a)[25])
        {
            static const uint64_t RC[24] = {
                0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull, 0x000000000000808bull,
                0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull, 0x000000000000008aull, 0x0000000000000088ull,
                0x0000000080008009ull, 0x000000008000000aull, 0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull,
                0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800aull, 0x800000008000000aull,
                0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull
            };
            // rotation offsets r[x][y] and the pi permutation (x, y) -> (y, 2x + 3y), lane index = x + 5y
            static const int ROT[25] = { 0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14 };
            for (int round = 0; round < 24; round++)
            {
                uint64_t c[5], d[5], b[25];
                for (int x = 0; x < 5; x++)
                    c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
                for (int x = 0; x < 5; x++)
                    d[x] = c[(x + 4) % 5] ^ rotl(c[(x + 1) % 5], 1);
                for (int i = 0; i < 25; i++)
                    a[i] ^= d[i % 5];
                for (int x = 0; x < 5; x++)
                    for (int y = 0; y < 5; y++)
                        b[y + 5 * ((2 * x + 3 * y) % 5)] = rotl(a[x + 5 * y], ROT[x + 5 * y]);
                for (int y = 0; y < 5; y++)
                    for (int x = 0; x < 5; x++)
                        a[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
                a[0] ^= RC[round];
            }
        }
        inline void shake256(uint8_t *out, size_t outlen, const uint8_t *in, size_t inlen)
        {
            constexpr size_t rate = 136;
            uint64_t a[25];
            std::memset(a, 0, sizeof(a));
            uint8_t block[rate];
            while (inlen >= rate)
            {
                for (size_t i = 0; i < rate / 8; i++)
                {
                    uint64_t w;
                    std::memcpy(&w, in + 8 * i, 8);
                    a[i] ^= w;
                }
                f1600(a);
                in += rate;
                inlen -= rate;
            }
            std::memset(block, 0, rate);
            std::memcpy(block, in, inlen);
            block[inlen] ^= 0x1F;
            block[rate - 1] ^= 0x80;
            for (size_t i = 0; i < rate / 8; i++)
            {
                uint64_t w;
                std::memcpy(&w, block + 8 * i, 8);
                a[i] ^= w;
            }
            f1600(a);
            while (outlen)
            {
                const size_t take = outlen < rate ? outlen : rate;
                std::memcpy(out, a, take); // little-endian lanes
                out += take;
                outlen -= take;
                if (outlen)
                    f1600(a);
            }
        }
    } // namespace keccak
} // namespace sealhip
