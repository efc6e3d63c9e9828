#pragma once
// Shared by objects.cpp and the evaluator*.cpp files: small host helpers (each translation unit gets its own copy).
#include "evaluator.h"
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <limits>

namespace sealhip
{
    namespace
    {
        void ck(hipError_t e, const char *what)
        {
            if (e != hipSuccess)
                throw std::runtime_error(std::string("HIP failure in ") + what + ": " + hipGetErrorString(e));
        }

        // util::are_close<double> (util/common.h:574-578)
        bool are_close(double a, double b)
        {
            double scale_factor = std::max({ std::fabs(a), std::fabs(b), 1.0 });
            return std::fabs(a - b) < std::numeric_limits<double>::epsilon() * scale_factor;
        }

        // util::naf (util/numth.h:22-42)
        std::vector<int> naf(int value)
        {
            std::vector<int> res;
            bool sign = value < 0;
            value = std::abs(value);
            for (int i = 0; value; i++)
            {
                int zi = (value & 1) ? 2 - (value & 3) : 0;
                value = (value - zi) >> 1;
                if (zi)
                    res.push_back((sign ? -zi : zi) * (1 << i));
            }
            return res;
        }

        // balance_correction_factors (evaluator.cpp:50-117).  Two BGV operands carry correction factors c1, c2 (units mod t);
        // before they can be added both are scaled to a common factor f = e1*c1 = e2*c2 (mod t), and the scalars should be
        // small as centred residues because they multiply the noise.  With rho = c2 / c1 (mod t) the admissible pairs are
        // exactly the lattice points e1 = rho * e2 (mod t), and the short ones appear among the remainders of Euclid's
        // algorithm on (t, rho): every step yields r = s * rho (mod t).  The walk starts from (rho, 1) and a later step
        // replaces the choice only when its centred 1-norm is STRICTLY smaller and r is a unit - the reference's tie rule,
        // which decides the result words and is therefore kept.
        void balance_correction_factors(uint64_t factor1, uint64_t factor2, uint64_t t, uint64_t &f, uint64_t &e1, uint64_t &e2)
        {
            const uint64_t c1 = factor1 % t, c2 = factor2 % t;
            if (c1 == 0 || std::__gcd(c1, t) != 1)
                throw std::logic_error("invalid correction factor1");
            const uint64_t rho = host::mulmod(host::invmod(c1, t), c2, t);
            const auto residue = [t](int64_t v) { // v mod t in [0, t)
                const uint64_t m = static_cast<uint64_t>(v < 0 ? -v : v) % t;
                return (v < 0 && m) ? t - m : m;
            };
            const auto centred_abs = [t](uint64_t x) { // |x| as the centred representative of x mod t
                return static_cast<int64_t>(x > t / 2 ? t - x : x);
            };
            struct Row
            {
                int64_t r, s; // r = s * rho (mod t)
            };
            Row above{ static_cast<int64_t>(t), 0 }, here{ static_cast<int64_t>(rho), 1 };
            e1 = rho;
            e2 = 1;
            int64_t best = centred_abs(e1) + centred_abs(e2);
            while (here.r != 0)
            {
                const int64_t quot = above.r / here.r;
                const Row below{ above.r - quot * here.r, above.s - quot * here.s };
                above = here;
                here = below;
                const uint64_t r = residue(here.r), sc = residue(here.s);
                if (r == 0 || std::__gcd(r, t) != 1)
                    continue;
                const int64_t norm = centred_abs(r) + centred_abs(sc);
                if (norm < best)
                {
                    best = norm;
                    e1 = r;
                    e2 = sc;
                }
            }
            f = host::mulmod(e1, c1, t);
        }

        NttBatch plain_batch(uint64_t *data, size_t outer_stride, unsigned ncomp, unsigned nouter, unsigned prime_first)
        {
            NttBatch b{};
            b.data = data;
            b.outer_stride = outer_stride;
            b.ncomp = ncomp;
            b.nouter = nouter;
            b.comp_prime = nullptr;
            b.prime_first = prime_first;
            b.src = nullptr;
            return b;
        }
    } // namespace

    // deferred key-switch tails: counters behind SealHip_TailStats (defined in evaluator.cpp)
    extern std::atomic<uint64_t> g_tail_folded, g_tail_plain, g_tail_dropped;
} // namespace sealhip
