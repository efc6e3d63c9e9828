// Host-side number theory used to build the device tables (no device code here).
// Each routine restates the reference routine it names so that a context built here
// carries exactly the constants a SEALContext would:
//   get_primes                    native/src/seal/util/numth.cpp:278-311
//   CoeffModulus::Create          native/src/seal/modulus.cpp:143-184
//   PlainModulus::Batching        native/src/seal/modulus.h:540
//   try_minimal_primitive_root    native/src/seal/util/numth.cpp:386-413  (the minimum is deterministic)
//   try_invert_uint_mod           native/src/seal/util/numth.h
#pragma once
#include "modarith.h"
#include <cstddef>
#include <stdexcept>
#include <vector>

namespace sealhip
{
    namespace host
    {
        typedef unsigned __int128 u128;

        inline uint64_t mulmod(uint64_t a, uint64_t b, uint64_t m)
        {
            return (uint64_t)((u128)a * b % m);
        }
        uint64_t powmod(uint64_t a, uint64_t e, uint64_t m);
        // a^-1 mod m (m need not be prime, e.g. 2^32 or 2N); throws if not invertible.
        uint64_t invmod(uint64_t a, uint64_t m);
        // cryptographically secure random bytes from the operating system (the role of seal::random_bytes, randomgen.cpp:24-60)
        void random_bytes(void *dst, size_t count);
        bool is_prime(uint64_t n);
        int bit_count(uint64_t v);

        // Primes p = 1 (mod factor) with exactly `bit_size` bits, descending from 2^bit_size.
        std::vector<uint64_t> get_primes(uint64_t factor, int bit_size, size_t count);
        std::vector<uint64_t> coeff_modulus_create(size_t poly_modulus_degree, const std::vector<int> &bit_sizes);
        uint64_t plain_modulus_batching(size_t poly_modulus_degree, int bit_size);

        // Smallest primitive `degree`-th root of unity mod prime q (degree a power of two).
        bool minimal_primitive_root(uint64_t degree, uint64_t q, uint64_t &root);

        ModDesc make_mod(uint64_t q);
        ShoupOp make_shoup(uint64_t w, uint64_t q);

        inline uint32_t reverse_bits(uint32_t x, int bits)
        {
            uint32_t r = 0;
            for (int i = 0; i < bits; i++)
                r |= ((x >> i) & 1u) << (bits - 1 - i);
            return r;
        }

        // Little-endian multi-precision product of 64-bit values and its significant bit count
        // (RNSBase::base_prod / get_significant_bit_count_uint as used at rns.cpp:612).
        std::vector<uint64_t> product(const std::vector<uint64_t> &values);
        int significant_bits(const std::vector<uint64_t> &v);
    } // namespace host
} // namespace sealhip
