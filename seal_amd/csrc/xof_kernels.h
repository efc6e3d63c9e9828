// sample_poly_uniform on the device ("encrypt needs an on-device XOF (Blake2xb) to stay bit-exact with seeded
// runs").  The reference's default PRNG (Blake2xbPRNG::refill_buffer, native/src/seal/randomgen.cpp:204-214) produces its stream
// in 4096-byte buffers, buffer b = blake2xb(out 4096 bytes, in = b as 8 bytes, key = the 64-byte seed); BLAKE2Xb (BLAKE2X
// specification, section 2) derives each 64-byte piece of such a buffer from the buffer's root hash h0 with ONE compression:
// piece i = BLAKE2b(h0; node_offset = i, xof_length = 4096).  Every 64 bytes of the stream are therefore an independent
// function of (seed, position): one thread per piece (two compressions for h0 - key block + counter block - and one for the
// piece), no state carried between threads.  sample_poly_uniform (util/rlwe.cpp:152-194) takes K*N words of the stream,
// replaces a word >= max_multiple = 2^64 - 1 - ((2^64 - 1) mod q_j) - 1 by the NEXT words of the stream (after the K*N) in
// coefficient order, and reduces mod q_j: the kernel reduces the accepted words and flags the rejected ones; the few
// replacements (probability ~ q / 2^64 per word) are a sequential walk the host does (xof.cpp) before a scatter kernel
// patches them in.
#pragma once
#include "context.h"

namespace sealhip
{
    struct XofJob
    {
        uint64_t seed[8];
        uint64_t *dst; // [K][N] words on the device
        uint64_t prng_type = 1; // 1 = Blake2xbPRNG, 2 = Shake256PRNG (randomgen.h: prng_type)
    };
    struct XofPatch
    {
        uint64_t *dst;
        uint64_t value;
    };
    // jobs: device array of njobs; reject: device bitmap, njobs * (K*N/32) words, zeroed by the caller: bit w of job j = word w
    // of the polynomial was rejected (its dst word is left unreduced).  Requires K*N*8 to be a multiple of 4096.
    hipError_t k_blake2xb_uniform(const ModDesc *mods, const XofJob *jobs, unsigned njobs, unsigned *reject, unsigned n_log, unsigned K,
                                  hipStream_t s);
    // The same for Shake256PRNG seeds (randomgen.cpp:216-227: buffer b = SHAKE256(seed || b as 8 bytes), 4096 bytes).  A buffer is
    // one sponge: its 31 permutations are sequential, the buffers are independent - one thread per 4096-byte buffer (Keccak-f[1600]
    // written from FIPS 202, the 25 lanes in registers).
    hipError_t k_shake256_uniform(const ModDesc *mods, const XofJob *jobs, unsigned njobs, unsigned *reject, unsigned n_log, unsigned K,
                                  hipStream_t s);
    // the raw PRNG stream of `seed`: 64-byte pieces first_piece .. first_piece + pieces - 1 (piece p = bytes 64 p .. of the stream)
    struct XofSeed
    {
        uint64_t w[8];
    };
    hipError_t k_blake2xb_stream(const XofSeed &seed, uint64_t first_piece, size_t pieces, uint64_t *out, hipStream_t s);
    // sample_poly_ternary / sample_poly_cbd as signed bytes from a stream in HBM: small[0 .. n_ternary) from the 4-byte draws at
    // stream + 4 k, small[n_ternary .. n_ternary + n_cbd) from the 6-byte draws at stream + cbd_offset + 6 k; *redraw is set when a
    // ternary draw would have been redrawn by the reference (the caller then samples on the host)
    hipError_t k_small_from_stream(const uint8_t *stream, size_t n_ternary, size_t cbd_offset, size_t n_cbd, int8_t *small, unsigned *redraw,
                                   hipStream_t s);
    hipError_t k_apply_patches(const XofPatch *patches, size_t count, hipStream_t s);
} // namespace sealhip
