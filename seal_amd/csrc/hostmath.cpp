#include "hostmath.h"
#include <cstdio>
#include <map>

namespace sealhip
{
    namespace host
    {
        uint64_t powmod(uint64_t a, uint64_t e, uint64_t m)
        {
            uint64_t r = 1 % m;
            a %= m;
            while (e)
            {
                if (e & 1)
                    r = mulmod(r, a, m);
                a = mulmod(a, a, m);
                e >>= 1;
            }
            return r;
        }

        uint64_t invmod(uint64_t a, uint64_t m)
        {
            // extended Euclid on signed 128-bit to stay exact for 64-bit moduli
            __int128 t = 0, nt = 1;
            __int128 r = m, nr = a % m;
            while (nr != 0)
            {
                __int128 q = r / nr;
                __int128 tmp = t - q * nt;
                t = nt;
                nt = tmp;
                tmp = r - q * nr;
                r = nr;
                nr = tmp;
            }
            if (r != 1)
                throw std::invalid_argument("value is not invertible");
            if (t < 0)
                t += m;
            return (uint64_t)t;
        }

        void random_bytes(void *dst, size_t count)
        {
            FILE *f = std::fopen("/dev/urandom", "rb");
            const size_t got = f ? std::fread(dst, 1, count, f) : 0;
            if (f)
                std::fclose(f);
            if (got != count)
                throw std::runtime_error("no entropy source (/dev/urandom)");
        }

        int bit_count(uint64_t v)
        {
            return v ? 64 - __builtin_clzll(v) : 0;
        }

        bool is_prime(uint64_t n)
        {
            if (n < 2)
                return false;
            static const uint64_t small[] = { 2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37 };
            for (uint64_t p : small)
            {
                if (n == p)
                    return true;
                if (n % p == 0)
                    return false;
            }
            uint64_t d = n - 1;
            int r = 0;
            while ((d & 1) == 0)
            {
                d >>= 1;
                r++;
            }
            // This base set is a deterministic Miller-Rabin certificate for every n < 2^64.
            for (uint64_t a : small)
            {
                uint64_t x = powmod(a, d, n);
                if (x == 1 || x == n - 1)
                    continue;
                bool comp = true;
                for (int i = 1; i < r; i++)
                {
                    x = mulmod(x, x, n);
                    if (x == n - 1)
                    {
                        comp = false;
                        break;
                    }
                }
                if (comp)
                    return false;
            }
            return true;
        }

        std::vector<uint64_t> get_primes(uint64_t factor, int bit_size, size_t count)
        {
            std::vector<uint64_t> out;
            uint64_t value = ((uint64_t(1) << bit_size) - 1) / factor * factor + 1;
            uint64_t lower = uint64_t(1) << (bit_size - 1);
            while (count > 0 && value > lower)
            {
                if (is_prime(value))
                {
                    out.push_back(value);
                    count--;
                }
                value -= factor;
            }
            if (count > 0)
                throw std::logic_error("failed to find enough qualifying primes");
            return out;
        }

        std::vector<uint64_t> coeff_modulus_create(size_t n, const std::vector<int> &bit_sizes)
        {
            std::map<int, size_t> counts;
            for (int b : bit_sizes)
            {
                if (b > 60 || b < 2)
                    throw std::invalid_argument("bit_sizes is invalid");
                counts[b]++;
            }
            std::map<int, std::vector<uint64_t>> table;
            for (auto &kv : counts)
                table[kv.first] = get_primes(2 * (uint64_t)n, kv.first, kv.second);
            std::vector<uint64_t> out;
            for (int b : bit_sizes)
            {
                out.push_back(table[b].back());
                table[b].pop_back();
            }
            return out;
        }

        uint64_t plain_modulus_batching(size_t n, int bit_size)
        {
            return coeff_modulus_create(n, { bit_size })[0];
        }

        bool minimal_primitive_root(uint64_t degree, uint64_t q, uint64_t &root)
        {
            if ((q - 1) % degree != 0)
                return false;
            uint64_t quotient = (q - 1) / degree;
            // any primitive degree-th root: g^((q-1)/degree) with (..)^(degree/2) == -1
            uint64_t r = 0;
            bool found = false;
            for (uint64_t g = 2; g < 2000 && !found; g++)
            {
                r = powmod(g, quotient, q);
                if (powmod(r, degree >> 1, q) == q - 1)
                    found = true;
            }
            if (!found)
                return false;
            // all primitive roots are the odd powers of r; take the minimum
            uint64_t sq = mulmod(r, r, q);
            uint64_t cur = r, best = r;
            for (uint64_t i = 0; i < degree; i += 2)
            {
                if (cur < best)
                    best = cur;
                cur = mulmod(cur, sq, q);
            }
            root = best;
            return true;
        }

        ModDesc make_mod(uint64_t q)
        {
            ModDesc m;
            m.q = q;
            m.two_q = q << 1;
            // floor(2^128 / q) as two words
            u128 num_hi = ((u128)1 << 64); // 2^64
            uint64_t hi = (uint64_t)(num_hi / q);
            u128 rem = num_hi % q;
            uint64_t lo = (uint64_t)((rem << 64) / q);
            m.ratio_hi = hi;
            m.ratio_lo = lo;
            return m;
        }

        ShoupOp make_shoup(uint64_t w, uint64_t q)
        {
            ShoupOp s;
            s.w = w;
            s.wq = (uint64_t)(((u128)w << 64) / q);
            return s;
        }

        std::vector<uint64_t> product(const std::vector<uint64_t> &values)
        {
            std::vector<uint64_t> acc(1, 1);
            for (uint64_t v : values)
            {
                uint64_t carry = 0;
                for (size_t i = 0; i < acc.size(); i++)
                {
                    u128 t = (u128)acc[i] * v + carry;
                    acc[i] = (uint64_t)t;
                    carry = (uint64_t)(t >> 64);
                }
                if (carry)
                    acc.push_back(carry);
            }
            return acc;
        }

        int significant_bits(const std::vector<uint64_t> &v)
        {
            for (size_t i = v.size(); i-- > 0;)
                if (v[i])
                    return (int)(i * 64) + bit_count(v[i]);
            return 0;
        }
    } // namespace host
} // namespace sealhip
