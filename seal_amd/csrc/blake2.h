// BLAKE2b (RFC 7693) and the BLAKE2Xb extendable-output construction (BLAKE2X specification, section 2),
// host side, written from the specifications.
//
// Why the hot path needs a hash at all — two formats either side of it are defined in terms of BLAKE2:
//   * parms_id, the name of a level of the modulus-switching chain carried by every serialized Ciphertext /
//     KSwitchKeys, is BLAKE2b-256 of (scheme, N, q_0..q_{k-1}, t) as 64-bit words
//     (EncryptionParameters::compute_parms_id, native/src/seal/encryptionparams.cpp:117-147; util/hash.h:30-37);
//   * a *seeded* ciphertext / key stores c_1 as a 64-byte seed; c_1 is re-expanded with the reference's default
//     PRNG, which draws 4096-byte buffers blake2xb(out = buffer, in = counter (8 bytes), key = seed (64 bytes))
//     (Blake2xbPRNG::refill_buffer, native/src/seal/randomgen.cpp:204-214) and rejection-samples them
//     (sample_poly_uniform, native/src/seal/util/rlwe.cpp).
// The reference vendors the BLAKE2 authors' C code (native/src/seal/util/blake2b.c, blake2xb.c); this is an
// independent restatement checked against it word for word in tests/test_serialization.py.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>

namespace sealhip
{
    namespace blake2
    {
        struct State
        {
            uint64_t h[8];
            uint64_t t0 = 0, t1 = 0;
            uint8_t buf[128];
            size_t buflen = 0;
            size_t outlen = 0;
        };

        inline uint64_t rotr(uint64_t x, int c)
        {
            return (x >> c) | (x << (64 - c));
        }
        inline uint64_t load64(const uint8_t *p)
        {
            uint64_t v;
            std::memcpy(&v, p, 8); // little-endian host (x86-64)
            return v;
        }

        inline void compress(State &s, const uint8_t *block, bool last)
        {
            static const uint64_t IV[8] = { 0x6a09e667f3bcc908ull, 0xbb67ae8584caa73bull, 0x3c6ef372fe94f82bull, 0xa54ff53a5f1d36f1ull,
                                            0x510e527fade682d1ull, 0x9b05688c2b3e6c1full, 0x1f83d9abfb41bd6bull, 0x5be0cd19137e2179ull };
            static constexpr uint8_t SIGMA[12][16] = {
                { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15 }, { 14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3 },
                { 11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4 }, { 7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8 },
                { 9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13 }, { 2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9 },
                { 12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11 }, { 13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10 },
                { 6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5 }, { 10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0 },
                { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15 }, { 14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3 },
            };
            uint64_t m[16], v[16];
            for (int i = 0; i < 16; i++)
                m[i] = load64(block + 8 * i);
            for (int i = 0; i < 8; i++)
            {
                v[i] = s.h[i];
                v[i + 8] = IV[i];
            }
            v[12] ^= s.t0;
            v[13] ^= s.t1;
            if (last)
                v[14] = ~v[14];
            auto G = [&](int r, int i, int a, int b, int c, int d) {
                v[a] = v[a] + v[b] + m[SIGMA[r][2 * i]];
                v[d] = rotr(v[d] ^ v[a], 32);
                v[c] = v[c] + v[d];
                v[b] = rotr(v[b] ^ v[c], 24);
                v[a] = v[a] + v[b] + m[SIGMA[r][2 * i + 1]];
                v[d] = rotr(v[d] ^ v[a], 16);
                v[c] = v[c] + v[d];
                v[b] = rotr(v[b] ^ v[c], 63);
            };
#pragma GCC unroll 12
            for (int r = 0; r < 12; r++)
            {
                G(r, 0, 0, 4, 8, 12);
                G(r, 1, 1, 5, 9, 13);
                G(r, 2, 2, 6, 10, 14);
                G(r, 3, 3, 7, 11, 15);
                G(r, 4, 0, 5, 10, 15);
                G(r, 5, 1, 6, 11, 12);
                G(r, 6, 2, 7, 8, 13);
                G(r, 7, 3, 4, 9, 14);
            }
            for (int i = 0; i < 8; i++)
                s.h[i] ^= v[i] ^ v[i + 8];
        }

        // parameter block (RFC 7693 section 2.5 with the BLAKE2X fields): only the fields the two uses need
        struct Params
        {
            uint8_t digest_length = 64, key_length = 0, fanout = 1, depth = 1;
            uint32_t leaf_length = 0, node_offset = 0, xof_length = 0;
            uint8_t node_depth = 0, inner_length = 0;
        };

        inline void init(State &s, const Params &p, const void *key)
        {
            static const uint64_t IV[8] = { 0x6a09e667f3bcc908ull, 0xbb67ae8584caa73bull, 0x3c6ef372fe94f82bull, 0xa54ff53a5f1d36f1ull,
                                            0x510e527fade682d1ull, 0x9b05688c2b3e6c1full, 0x1f83d9abfb41bd6bull, 0x5be0cd19137e2179ull };
            uint8_t pb[64];
            std::memset(pb, 0, sizeof(pb));
            pb[0] = p.digest_length;
            pb[1] = p.key_length;
            pb[2] = p.fanout;
            pb[3] = p.depth;
            std::memcpy(pb + 4, &p.leaf_length, 4);
            std::memcpy(pb + 8, &p.node_offset, 4);
            std::memcpy(pb + 12, &p.xof_length, 4);
            pb[16] = p.node_depth;
            pb[17] = p.inner_length;
            for (int i = 0; i < 8; i++)
                s.h[i] = IV[i] ^ load64(pb + 8 * i);
            s.t0 = s.t1 = 0;
            s.buflen = 0;
            s.outlen = p.digest_length;
            if (p.key_length)
            {
                // the key, zero-padded to one block, is the first block of the message
                std::memset(s.buf, 0, 128);
                std::memcpy(s.buf, key, p.key_length);
                s.buflen = 128;
            }
        }

        inline void update(State &s, const void *in, size_t inlen)
        {
            const uint8_t *p = static_cast<const uint8_t *>(in);
            while (inlen)
            {
                if (s.buflen == 128)
                {
                    // a full buffer is compressed only once more input is known to follow (it is not the last block)
                    s.t0 += 128;
                    if (s.t0 < 128)
                        s.t1++;
                    compress(s, s.buf, false);
                    s.buflen = 0;
                }
                size_t take = 128 - s.buflen;
                if (take > inlen)
                    take = inlen;
                std::memcpy(s.buf + s.buflen, p, take);
                s.buflen += take;
                p += take;
                inlen -= take;
            }
        }

        inline void final(State &s, void *out)
        {
            s.t0 += s.buflen;
            if (s.t0 < s.buflen)
                s.t1++;
            std::memset(s.buf + s.buflen, 0, 128 - s.buflen);
            compress(s, s.buf, true);
            std::memcpy(out, s.h, s.outlen); // little-endian words
        }

        // blake2b(out, outlen <= 64, in, inlen, key, keylen <= 64)
        inline void blake2b(void *out, size_t outlen, const void *in, size_t inlen, const void *key = nullptr, size_t keylen = 0)
        {
            Params p;
            p.digest_length = (uint8_t)outlen;
            p.key_length = (uint8_t)keylen;
            State s;
            init(s, p, key);
            update(s, in, inlen);
            final(s, out);
        }

        // blake2xb(out, outlen < 2^32 - 1, in, inlen, key, keylen <= 64)
        inline void blake2xb(void *out, size_t outlen, const void *in, size_t inlen, const void *key, size_t keylen)
        {
            Params root;
            root.digest_length = 64;
            root.key_length = (uint8_t)keylen;
            root.xof_length = (uint32_t)outlen;
            State s;
            init(s, root, key);
            update(s, in, inlen);
            uint8_t h0[64];
            final(s, h0);
            uint8_t *o = static_cast<uint8_t *>(out);
            for (uint32_t i = 0; outlen; i++)
            {
                const size_t blk = outlen < 64 ? outlen : 64;
                Params p;
                p.digest_length = (uint8_t)blk;
                p.key_length = 0;
                p.fanout = 0;
                p.depth = 0;
                p.leaf_length = 64;
                p.node_offset = i;
                p.xof_length = root.xof_length;
                p.node_depth = 0;
                p.inner_length = 64;
                State b;
                init(b, p, nullptr);
                update(b, h0, 64);
                final(b, o);
                o += blk;
                outlen -= blk;
            }
        }
    } // namespace blake2
} // namespace sealhip
