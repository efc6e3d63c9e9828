#include "xof.h"
#include "pool.h"
#include "serial.h"
#include <cstring>
#include <stdexcept>

namespace sealhip
{
    namespace
    {
        void ck(hipError_t e, const char *what)
        {
            if (e != hipSuccess)
                throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
        }
    } // namespace

    void sample_uniform_device(const Context &ctx, size_t K, const std::vector<XofJob> &jobs)
    {
        const size_t n = ctx.n(), words = K * n, njobs = jobs.size();
        if (!njobs)
            return;
        if (!xof_device_ok((uint8_t)jobs[0].prng_type, K, n))
            throw std::logic_error("polynomial is not a whole number of PRNG buffers");
        const size_t map_words = words / 32; // unsigned words of the rejection bitmap per job
        const size_t job_words = (njobs * sizeof(XofJob) + 7) / 8;
        Scratch djobs(job_words), dmap((njobs * map_words * 4 + 7) / 8);
        ck(hipMemcpyAsync(djobs.p, jobs.data(), njobs * sizeof(XofJob), hipMemcpyHostToDevice, nullptr), "upload seeds");
        ck(hipMemsetAsync(dmap.p, 0, njobs * map_words * 4, nullptr), "clear bitmap");
        // jobs of one kind form runs (in practice all jobs of a call share the PRNG type): one launch per run
        for (size_t j0 = 0; j0 < njobs;)
        {
            size_t j1 = j0 + 1;
            while (j1 < njobs && jobs[j1].prng_type == jobs[j0].prng_type)
                j1++;
            const XofJob *dj = reinterpret_cast<const XofJob *>(djobs.p) + j0;
            unsigned *dm = reinterpret_cast<unsigned *>(dmap.p) + j0 * map_words;
            if (jobs[j0].prng_type == 2)
                ck(k_shake256_uniform(ctx.dev_mods(), dj, (unsigned)(j1 - j0), dm, (unsigned)ctx.log_n(), (unsigned)K, nullptr), "shake256");
            else
                ck(k_blake2xb_uniform(ctx.dev_mods(), dj, (unsigned)(j1 - j0), dm, (unsigned)ctx.log_n(), (unsigned)K, nullptr), "blake2xb");
            j0 = j1;
        }
        std::vector<uint32_t> map(njobs * map_words);
        ck(hipMemcpy(map.data(), dmap.p, map.size() * 4, hipMemcpyDeviceToHost), "download bitmap");

        // the replacements: per polynomial the stream continues after its K*N words; rejected positions take the next accepted
        // words in (component, coefficient) order - the loop of sample_poly_uniform (util/rlwe.cpp:170-190)
        std::vector<XofPatch> patches;
        const uint64_t *primes = ctx.coeff_modulus().data();
        for (size_t j = 0; j < njobs; j++)
        {
            const uint32_t *bits = map.data() + j * map_words;
            serial::Prng prng((uint8_t)jobs[j].prng_type, jobs[j].seed);
            prng.counter = words * 8 / 4096;
            prng.parallel = false;
            for (size_t g = 0; g < map_words; g++)
            {
                uint32_t b = bits[g];
                while (b)
                {
                    const unsigned bit = (unsigned)__builtin_ctz(b);
                    b &= b - 1;
                    const size_t w = g * 32 + bit;
                    const uint64_t q = primes[w / n];
                    const uint64_t max_multiple = ~0ull - (~0ull % q) - 1;
                    uint64_t rand;
                    do
                        prng.generate(sizeof(rand), reinterpret_cast<uint8_t *>(&rand));
                    while (rand >= max_multiple);
                    patches.push_back(XofPatch{ jobs[j].dst + w, rand % q });
                }
            }
        }
        if (!patches.empty())
        {
            Scratch dp(patches.size() * sizeof(XofPatch) / 8);
            ck(hipMemcpy(dp.p, patches.data(), patches.size() * sizeof(XofPatch), hipMemcpyHostToDevice), "upload replacements");
            ck(k_apply_patches(reinterpret_cast<const XofPatch *>(dp.p), patches.size(), nullptr), "patch");
            ck(hipStreamSynchronize(nullptr), "xof sync");
        }
    }
} // namespace sealhip
