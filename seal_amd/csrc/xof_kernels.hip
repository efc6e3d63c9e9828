// See xof_kernels.h.  BLAKE2b compression written from RFC 7693; 64-bit adds / xors / rotates only, the message schedule
// resolved at compile time so that the 16 message words stay in registers.
#include "xof_kernels.h"

namespace sealhip
{
    namespace
    {
        constexpr unsigned kBlock = 256;
        __device__ __forceinline__ uint64_t b2_iv(int i)
        {
            constexpr uint64_t iv[8] = { 0x6a09e667f3bcc908ull, 0xbb67ae8584caa73bull, 0x3c6ef372fe94f82bull, 0xa54ff53a5f1d36f1ull,
                                         0x510e527fade682d1ull, 0x9b05688c2b3e6c1full, 0x1f83d9abfb41bd6bull, 0x5be0cd19137e2179ull };
            return iv[i];
        }

        constexpr uint64_t kXofLength = 4096; // the reference PRNG's buffer (randomgen.h: buffer_size_)

        __device__ __forceinline__ uint64_t rotr64(uint64_t x, int c)
        {
            return (x >> c) | (x << (64 - c));
        }
#define SHL_B2_G(a, b, c, d, x, y)   \
    v[a] = v[a] + v[b] + (x);        \
    v[d] = rotr64(v[d] ^ v[a], 32);  \
    v[c] = v[c] + v[d];              \
    v[b] = rotr64(v[b] ^ v[c], 24);  \
    v[a] = v[a] + v[b] + (y);        \
    v[d] = rotr64(v[d] ^ v[a], 16);  \
    v[c] = v[c] + v[d];              \
    v[b] = rotr64(v[b] ^ v[c], 63);
#define SHL_B2_ROUND(s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, s10, s11, s12, s13, s14, s15) \
    SHL_B2_G(0, 4, 8, 12, m[s0], m[s1])                                                   \
    SHL_B2_G(1, 5, 9, 13, m[s2], m[s3])                                                   \
    SHL_B2_G(2, 6, 10, 14, m[s4], m[s5])                                                  \
    SHL_B2_G(3, 7, 11, 15, m[s6], m[s7])                                                  \
    SHL_B2_G(0, 5, 10, 15, m[s8], m[s9])                                                  \
    SHL_B2_G(1, 6, 11, 12, m[s10], m[s11])                                                \
    SHL_B2_G(2, 7, 8, 13, m[s12], m[s13])                                                 \
    SHL_B2_G(3, 4, 9, 14, m[s14], m[s15])

        // h <- F(h, m, t, last) (RFC 7693 section 3.2); the byte counter fits one word here
        __device__ __forceinline__ void b2_compress(uint64_t (&h)[8], const uint64_t (&m)[16], uint64_t t, bool last)
        {
            uint64_t v[16];
#pragma unroll
            for (int i = 0; i < 8; i++)
            {
                v[i] = h[i];
                v[i + 8] = b2_iv(i);
            }
            v[12] ^= t;
            if (last)
                v[14] = ~v[14];
            SHL_B2_ROUND(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15)
            SHL_B2_ROUND(14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3)
            SHL_B2_ROUND(11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4)
            SHL_B2_ROUND(7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8)
            SHL_B2_ROUND(9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13)
            SHL_B2_ROUND(2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9)
            SHL_B2_ROUND(12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11)
            SHL_B2_ROUND(13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10)
            SHL_B2_ROUND(6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5)
            SHL_B2_ROUND(10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0)
            SHL_B2_ROUND(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15)
            SHL_B2_ROUND(14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3)
#pragma unroll
            for (int i = 0; i < 8; i++)
                h[i] ^= v[i] ^ v[i + 8];
        }
#undef SHL_B2_ROUND
#undef SHL_B2_G

        // h <- the 64 bytes (8 words) number `piece` of the PRNG stream of `seed`: buffer = piece / 64 is the PRNG's counter,
        // node = piece % 64 the position inside the 4096-byte buffer
        __device__ __forceinline__ void stream_piece(const uint64_t *seed, uint64_t piece, uint64_t (&h)[8])
        {
            const uint64_t buffer = piece >> 6;
            const uint64_t node = piece & 63;
            // root hash h0 = BLAKE2b-512(key = seed, message = counter), parameter block: digest 64, key 64, fanout 1, depth 1,
            // xof_length 4096.  Keyed: the key padded to one block is the first message block.
            uint64_t m[16];
#pragma unroll
            for (int i = 0; i < 8; i++)
                h[i] = b2_iv(i);
            h[0] ^= 64ull | (64ull << 8) | (1ull << 16) | (1ull << 24);
            h[1] ^= kXofLength << 32;
#pragma unroll
            for (int i = 0; i < 8; i++)
            {
                m[i] = seed[i];
                m[i + 8] = 0;
            }
            b2_compress(h, m, 128, false);
#pragma unroll
            for (int i = 1; i < 8; i++)
                m[i] = 0;
            m[0] = buffer;
            b2_compress(h, m, 128 + 8, true);

            // piece `node` of the buffer = BLAKE2b-512(h0), parameter block: digest 64, key 0, fanout 0, depth 0, leaf_length 64,
            // node_offset = node, xof_length 4096, node_depth 0, inner_length 64
#pragma unroll
            for (int i = 0; i < 8; i++)
                m[i] = h[i];
#pragma unroll
            for (int i = 0; i < 8; i++)
                h[i] = b2_iv(i);
            h[0] ^= 64ull | (64ull << 32);
            h[1] ^= node | (kXofLength << 32);
            h[2] ^= 64ull << 8;
            b2_compress(h, m, 64, true);
        }

        __global__ void __launch_bounds__(kBlock) blake2xb_uniform_kernel(
            const ModDesc *mods, const XofJob *jobs, unsigned *reject, unsigned n_log, unsigned K)
        {
            const size_t words = (size_t)K << n_log;
            const size_t piece = blockIdx.x * (size_t)kBlock + threadIdx.x; // 64 bytes of the stream = 8 words
            if (piece >= words / 8)
                return;
            const XofJob &job = jobs[blockIdx.y];
            uint64_t h[8];
            stream_piece(job.seed, piece, h);

            // sample_poly_uniform's acceptance test and reduction; 8 consecutive words lie in one RNS component
            const size_t w0 = piece * 8;
            const ModDesc md = mods[(unsigned)(w0 >> n_log)];
            const uint64_t max_multiple = ~0ull - barrett64(~0ull, md) - 1;
            unsigned rejected = 0;
#pragma unroll
            for (int t = 0; t < 8; t++)
            {
                if (h[t] >= max_multiple)
                    rejected |= 1u << t;
                else
                    h[t] = barrett64(h[t], md);
                job.dst[w0 + t] = h[t];
            }
            if (rejected)
                atomicOr(reject + blockIdx.y * (words / 32) + w0 / 32, rejected << (w0 % 32));
        }

        // ---- SHAKE256 (FIPS 202): Keccak-f[1600], rate 136 bytes, domain suffix 0x1F
        __device__ __forceinline__ uint64_t rotl64(uint64_t x, int c)
        {
            return c ? (x << c) | (x >> (64 - c)) : x;
        }
        __device__ __forceinline__ void keccak_f1600(uint64_t (&a)[25])
        {
            constexpr uint64_t RC[24] = {
                0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull, 0x000000000000808bull,
                0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull, 0x000000000000008aull, 0x0000000000000088ull,
                0x0000000080008009ull, 0x000000008000000aull, 0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull,
                0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800aull, 0x800000008000000aull,
                0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull
            };
            // rotation offsets of lane x + 5y; pi moves lane (x, y) to (y, 2x + 3y)
            constexpr int ROT[25] = { 0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14 };
#pragma unroll 1
            for (int round = 0; round < 24; round++)
            {
                uint64_t c[5], d[5], b[25];
#pragma unroll
                for (int x = 0; x < 5; x++)
                    c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
#pragma unroll
                for (int x = 0; x < 5; x++)
                    d[x] = c[(x + 4) % 5] ^ rotl64(c[(x + 1) % 5], 1);
#pragma unroll
                for (int x = 0; x < 5; x++)
#pragma unroll
                    for (int y = 0; y < 5; y++)
                        b[y + 5 * ((2 * x + 3 * y) % 5)] = rotl64(a[x + 5 * y] ^ d[x], ROT[x + 5 * y]);
#pragma unroll
                for (int y = 0; y < 5; y++)
#pragma unroll
                    for (int x = 0; x < 5; x++)
                        a[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
                a[0] ^= RC[round];
            }
        }

        // one thread = one 4096-byte buffer of the PRNG stream = 512 words of the polynomial
        __global__ void __launch_bounds__(64) shake256_uniform_kernel(
            const ModDesc *mods, const XofJob *jobs, unsigned *reject, unsigned n_log, unsigned K)
        {
            const size_t words = (size_t)K << n_log;
            const size_t buffer = blockIdx.x * (size_t)64 + threadIdx.x;
            if (buffer >= words / 512)
                return;
            const XofJob &job = jobs[blockIdx.y];
            uint64_t a[25];
#pragma unroll
            for (int i = 0; i < 25; i++)
                a[i] = 0;
            // absorb seed (64 bytes) || counter (8 bytes): 72 bytes < rate; pad: 0x1F after the message, 0x80 at byte 135
#pragma unroll
            for (int i = 0; i < 8; i++)
                a[i] = job.seed[i];
            a[8] = buffer;
            a[9] = 0x1Full;
            a[16] ^= 0x8000000000000000ull;
            keccak_f1600(a);
            size_t w = buffer * 512, left = 512;
            unsigned *map = reject + blockIdx.y * (words / 32);
            while (left)
            {
                const unsigned take = left < 17 ? (unsigned)left : 17u; // 17 lanes = 136 bytes per squeeze
#pragma unroll
                for (unsigned t = 0; t < 17; t++)
                {
                    if (t < take)
                    {
                        // (a buffer of 512 words may straddle two RNS components only when N < 512: the component is per word)
                        const ModDesc md = mods[(unsigned)((w + t) >> n_log)];
                        const uint64_t max_multiple = ~0ull - barrett64(~0ull, md) - 1;
                        uint64_t v = a[t];
                        if (v >= max_multiple)
                            atomicOr(map + (w + t) / 32, 1u << ((w + t) % 32));
                        else
                            v = barrett64(v, md);
                        job.dst[w + t] = v;
                    }
                }
                w += take;
                left -= take;
                if (left)
                    keccak_f1600(a);
            }
        }

        // the raw stream: pieces first_piece .. first_piece + pieces - 1 -> out[8 * pieces]
        __global__ void __launch_bounds__(kBlock) blake2xb_stream_kernel(XofSeed seed, uint64_t first_piece, size_t pieces, uint64_t *out)
        {
            const size_t p = blockIdx.x * (size_t)kBlock + threadIdx.x;
            if (p >= pieces)
                return;
            uint64_t h[8];
            stream_piece(seed.w, first_piece + p, h);
#pragma unroll
            for (int t = 0; t < 8; t++)
                out[p * 8 + t] = h[t];
        }

        // The Encryptor's small samplers over a stream already in HBM (util/rlwe.cpp:24-43, 120-150; serial.h for the ternary draw):
        // thread k < n_ternary: coefficient k of a ternary polynomial from the 4 bytes at 4 k; the others: coefficient of a
        // centred-binomial polynomial from the 6 bytes at cbd_offset + 6 (k - n_ternary).  A ternary draw the reference would
        // redraw (g * 3 mod 2^32 == 0: probability 2^-32) shifts the rest of the stream: it raises *redraw and the caller
        // repeats the sampling on the host.
        __global__ void __launch_bounds__(kBlock) small_from_stream_kernel(
            const uint8_t *stream, size_t n_ternary, size_t cbd_offset, size_t n_cbd, int8_t *small, unsigned *redraw)
        {
            const size_t k = blockIdx.x * (size_t)kBlock + threadIdx.x;
            if (k < n_ternary)
            {
                const uint8_t *b = stream + 4 * k;
                const uint32_t g = (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24);
                const uint64_t product = (uint64_t)g * 3u;
                if ((uint32_t)product < 1u)
                    atomicOr(redraw, 1u);
                small[k] = (int8_t)((int)(product >> 32) - 1);
            }
            else if (k < n_ternary + n_cbd)
            {
                const uint8_t *b = stream + cbd_offset + 6 * (k - n_ternary);
                small[k] = (int8_t)(__popc(b[0]) + __popc(b[1]) + __popc(b[2] & 0x1Fu) - __popc(b[3]) - __popc(b[4]) - __popc(b[5] & 0x1Fu));
            }
        }

        __global__ void __launch_bounds__(kBlock) apply_patches_kernel(const XofPatch *patches, size_t count)
        {
            const size_t i = blockIdx.x * (size_t)kBlock + threadIdx.x;
            if (i < count)
                *patches[i].dst = patches[i].value;
        }
    } // namespace

    hipError_t k_blake2xb_uniform(const ModDesc *mods, const XofJob *jobs, unsigned njobs, unsigned *reject, unsigned n_log, unsigned K,
                                  hipStream_t s)
    {
        const size_t pieces = ((size_t)K << n_log) / 8;
        if (!pieces || !njobs)
            return hipSuccess;
        hipLaunchKernelGGL(blake2xb_uniform_kernel, dim3((unsigned)((pieces + kBlock - 1) / kBlock), njobs), dim3(kBlock), 0, s, mods, jobs,
                           reject, n_log, K);
        return hipGetLastError();
    }
    hipError_t k_shake256_uniform(const ModDesc *mods, const XofJob *jobs, unsigned njobs, unsigned *reject, unsigned n_log, unsigned K,
                                  hipStream_t s)
    {
        const size_t buffers = ((size_t)K << n_log) / 512;
        if (!buffers || !njobs)
            return hipSuccess;
        hipLaunchKernelGGL(shake256_uniform_kernel, dim3((unsigned)((buffers + 63) / 64), njobs), dim3(64), 0, s, mods, jobs, reject, n_log, K);
        return hipGetLastError();
    }
    hipError_t k_blake2xb_stream(const XofSeed &seed, uint64_t first_piece, size_t pieces, uint64_t *out, hipStream_t s)
    {
        if (!pieces)
            return hipSuccess;
        hipLaunchKernelGGL(blake2xb_stream_kernel, dim3((unsigned)((pieces + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, seed, first_piece,
                           pieces, out);
        return hipGetLastError();
    }
    hipError_t k_small_from_stream(const uint8_t *stream, size_t n_ternary, size_t cbd_offset, size_t n_cbd, int8_t *small, unsigned *redraw,
                                   hipStream_t s)
    {
        const size_t work = n_ternary + n_cbd;
        if (!work)
            return hipSuccess;
        hipLaunchKernelGGL(small_from_stream_kernel, dim3((unsigned)((work + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, stream, n_ternary,
                           cbd_offset, n_cbd, small, redraw);
        return hipGetLastError();
    }
    hipError_t k_apply_patches(const XofPatch *patches, size_t count, hipStream_t s)
    {
        if (!count)
            return hipSuccess;
        hipLaunchKernelGGL(apply_patches_kernel, dim3((unsigned)((count + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, patches, count);
        return hipGetLastError();
    }
} // namespace sealhip
