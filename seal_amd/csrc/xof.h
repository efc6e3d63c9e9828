// Host driver of the device-side sample_poly_uniform (xof_kernels.h): launches the BLAKE2Xb kernel over a set of independent
// (seed, destination) polynomials, walks the rejected words in the reference's order and patches their replacements in.
#pragma once
#include "context.h"
#include "xof_kernels.h"
#include <vector>

namespace sealhip
{
    // true when a polynomial of K*N words can be expanded on the device (the stream is consumed in whole PRNG buffers)
    inline bool xof_device_ok(uint8_t prng_type, size_t K, size_t N)
    {
        return (prng_type == 1 || prng_type == 2) && K && N >= 8 && (K * N * 8) % 4096 == 0;
    }
    // dst_j = sample_poly_uniform(Blake2xbPRNG(seed_j) or Shake256PRNG(seed_j), XofJob::prng_type) over the first K primes of the context, every job [K][N] words in HBM.
    // Synchronous (returns when the words are in place).
    void sample_uniform_device(const Context &ctx, size_t K, const std::vector<XofJob> &jobs);
} // namespace sealhip
