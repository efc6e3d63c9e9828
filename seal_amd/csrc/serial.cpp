// Parser / writer of the reference's wire format for Ciphertext and KSwitchKeys: see serial.h for the layout and the
// reference lines each step follows.
#include "serial.h"
#include "blake2.h"
#include "shake256.h"
#include <cmath>
#include <cstring>
#include <limits>
#include <list>
#include <algorithm>
#include <atomic>
#include <dlfcn.h>
#include <zlib.h>
#include <stdexcept>
#include <thread>

namespace sealhip
{
    namespace serial
    {
        // UniformRandomGenerator::generate over refill_buffer (randomgen.cpp:179-227): 4096-byte buffers, buffer i = XOF(seed, counter = i)
        void Prng::refill()
        {
            if (type == 1)
                blake2::blake2xb(buf, sizeof(buf), &counter, sizeof(counter), seed, sizeof(seed));
            else
            {
                uint64_t ext[9];
                std::memcpy(ext, seed, sizeof(seed));
                ext[8] = counter;
                keccak::shake256(buf, sizeof(buf), reinterpret_cast<const uint8_t *>(ext), sizeof(ext));
            }
            counter++;
            head = 0;
        }
        void Prng::generate(size_t bytes, uint8_t *dst)
        {
            // Bulk requests (sample_poly_uniform's K*N words, the 4 N / 6 N bytes of the small samplers) are mostly whole 4096-byte
            // blocks, block i = XOF(seed, counter + i): independent, so after what is left in the buffer they are produced straight
            // into the destination on several host threads (BLAKE2Xb makes ~0.3 GB/s per core; a C5 polynomial is 7.9 MB).  The
            // stream - and every later draw - is unchanged.
            constexpr size_t kBlock = sizeof(buf);
            if (parallel && bytes >= 32 * kBlock)
            {
                const size_t left = kBlock - head;
                std::memcpy(dst, buf + head, left);
                head = kBlock;
                dst += left;
                bytes -= left;
                const size_t blocks = bytes / kBlock;
                unsigned nthreads = std::thread::hardware_concurrency();
                nthreads = nthreads > 16 ? 16 : (nthreads < 1 ? 1 : nthreads);
                if (nthreads > blocks / 8)
                    nthreads = (unsigned)(blocks / 8 ? blocks / 8 : 1);
                const uint64_t counter0 = counter;
                const uint8_t xof = type;
                uint64_t sd[8];
                std::memcpy(sd, seed, sizeof(sd));
                auto work = [&](unsigned tix) {
                    uint64_t ext[9];
                    std::memcpy(ext, sd, sizeof(sd));
                    for (size_t i = tix; i < blocks; i += nthreads)
                    {
                        const uint64_t c = counter0 + i;
                        if (xof == 1)
                            blake2::blake2xb(dst + i * kBlock, kBlock, &c, sizeof(c), sd, sizeof(sd));
                        else
                        {
                            ext[8] = c;
                            keccak::shake256(dst + i * kBlock, kBlock, reinterpret_cast<const uint8_t *>(ext), sizeof(ext));
                        }
                    }
                };
                std::vector<std::thread> pool;
                for (unsigned tix = 1; tix < nthreads; tix++)
                    pool.emplace_back(work, tix);
                work(0);
                for (auto &th : pool)
                    th.join();
                counter += blocks;
                dst += blocks * kBlock;
                bytes -= blocks * kBlock;
            }
            while (bytes)
            {
                if (head == sizeof(buf))
                    refill();
                size_t take = sizeof(buf) - head;
                if (take > bytes)
                    take = bytes;
                std::memcpy(dst, buf + head, take);
                head += take;
                dst += take;
                bytes -= take;
            }
        }
        // sample_poly_uniform (util/rlwe.cpp): bulk fill, then per component reject rand >= max_multiple (drawing the replacement
        // from the same stream, in coefficient order) and reduce
        void sample_poly_uniform(Prng &prng, const uint64_t *primes, size_t K, size_t N, uint64_t *dst)
        {
            prng.generate(K * N * sizeof(uint64_t), reinterpret_cast<uint8_t *>(dst));
            for (size_t j = 0; j < K; j++)
            {
                const uint64_t q = primes[j];
                const uint64_t max_random = ~0ull;
                const uint64_t max_multiple = max_random - (max_random % q) - 1;
                for (size_t k = 0; k < N; k++)
                {
                    uint64_t rand = dst[k];
                    while (rand >= max_multiple)
                        prng.generate(sizeof(uint64_t), reinterpret_cast<uint8_t *>(&rand));
                    dst[k] = rand % q;
                }
                dst += N;
            }
        }
        void sample_small_ternary(Prng &prng, size_t N, int8_t *dst)
        {
            // RandomToStandardAdapter::operator() = 4 bytes of the stream as one uint32_t (randomtostd.h:43-57): N of them in one
            // draw, a rejected one (g * 3 mod 2^32 == 0, i.e. g == 0) replaced by the next 4 bytes of the stream as the reference's
            // loop does - everything after it shifts by one draw
            std::vector<uint32_t> g(N);
            prng.generate(N * sizeof(uint32_t), reinterpret_cast<uint8_t *>(g.data()));
            size_t next = 0;
            auto draw = [&]() {
                uint32_t v;
                if (next < N)
                    v = g[next++];
                else
                    prng.generate(sizeof(v), reinterpret_cast<uint8_t *>(&v));
                return v;
            };
            for (size_t k = 0; k < N; k++)
            {
                uint64_t product;
                do
                    product = (uint64_t)draw() * 3u;
                while ((uint32_t)product < 1u /* threshold = (2^32 - 3) mod 3 */);
                dst[k] = (int8_t)((int)(product >> 32) - 1); // 0, 1, 2 -> coefficient -1, 0, 1
            }
        }
        // sample_poly_cbd (util/rlwe.cpp): centred binomial noise of standard deviation 3.2 - 6 bytes per coefficient, the
        // difference of two 21-bit Hamming weights
        void sample_small_cbd(Prng &prng, size_t N, int8_t *dst)
        {
            std::vector<uint8_t> x(6 * N);
            prng.generate(x.size(), x.data());
            for (size_t k = 0; k < N; k++)
            {
                const uint8_t *b = x.data() + 6 * k;
                dst[k] = (int8_t)(__builtin_popcount(b[0]) + __builtin_popcount(b[1]) + __builtin_popcount(b[2] & 0x1F) - __builtin_popcount(b[3]) -
                                  __builtin_popcount(b[4]) - __builtin_popcount(b[5] & 0x1F));
            }
        }
        namespace
        {
            // the small value replicated into every RNS component, negative values as q_j + v (what k_expand_small does on the device)
            void replicate(const int8_t *small, const uint64_t *primes, size_t K, size_t N, uint64_t *dst)
            {
                for (size_t j = 0; j < K; j++)
                    for (size_t k = 0; k < N; k++)
                        dst[j * N + k] = small[k] < 0 ? primes[j] - (uint64_t)(-small[k]) : (uint64_t)small[k];
            }
        } // namespace
        void sample_poly_ternary(Prng &prng, const uint64_t *primes, size_t K, size_t N, uint64_t *dst)
        {
            std::vector<int8_t> small(N);
            sample_small_ternary(prng, N, small.data());
            replicate(small.data(), primes, K, N, dst);
        }
        void sample_poly_cbd(Prng &prng, const uint64_t *primes, size_t K, size_t N, uint64_t *dst)
        {
            std::vector<int8_t> small(N);
            sample_small_cbd(prng, N, small.data());
            replicate(small.data(), primes, K, N, dst);
        }

        namespace
        {
            struct Header
            {
                uint16_t magic;
                uint8_t header_size, version_major, version_minor, compr_mode;
                uint16_t reserved;
                uint64_t size;
            };
            static_assert(sizeof(Header) == 16, "SEALHeader is 16 bytes");

            // an istream over a memory buffer, reduced to what Serialization::Load needs: a failed read is the
            // reference's ios_base::failure -> runtime_error("I/O error")
            struct Reader
            {
                const uint8_t *base;
                size_t size, pos = 0;
                bool seekable = true;                            // false inside an inflated payload (see framed)
                size_t inflate_limit = size_t(1) << 40;          // bound on one decompressed payload
                std::list<std::vector<uint8_t>> inflated;        // decompressed payloads (stable addresses)
                void skip(size_t n)
                {
                    if (pos > size || n > size - pos)
                        throw std::runtime_error("I/O error");
                    pos += n;
                }
                void read(void *dst, size_t n)
                {
                    if (pos > size || n > size - pos)
                        throw std::runtime_error("I/O error");
                    std::memcpy(dst, base + pos, n);
                    pos += n;
                }
                template <typename T>
                T get()
                {
                    T v;
                    read(&v, sizeof(T));
                    return v;
                }
            };

            struct Version
            {
                uint8_t major, minor;
            };

            // Serialization::IsCompatibleVersion (serialization.h:144-165)
            bool compatible_version(const Header &h)
            {
                if (h.version_major == kVersionMajor && h.version_minor <= kVersionMinor)
                    return true;
                return h.version_major == 3 && h.version_minor >= 4;
            }
            // Serialization::IsValidHeader (serialization.h:172-191)
            bool valid_header(const Header &h)
            {
                return h.magic == kMagic && h.header_size == kHeaderSize && compatible_version(h) && compr_mode_supported(h.compr_mode);
            }

            // libzstd through its stable C ABI, loaded on demand (the image ships libzstd.so.1 without headers)
            struct Zstd
            {
                struct Buf
                {
                    void *p;
                    size_t size, pos;
                };
                void *(*createDStream)() = nullptr;
                size_t (*freeDStream)(void *) = nullptr;
                size_t (*decompressStream)(void *, Buf *, Buf *) = nullptr;
                size_t (*compress)(void *, size_t, const void *, size_t, int) = nullptr;
                size_t (*compressBound)(size_t) = nullptr;
                unsigned (*isError)(size_t) = nullptr;
                bool ok = false;
                Zstd()
                {
                    void *h = dlopen("libzstd.so.1", RTLD_NOW | RTLD_LOCAL);
                    if (!h)
                        return;
                    createDStream = reinterpret_cast<void *(*)()>(dlsym(h, "ZSTD_createDStream"));
                    freeDStream = reinterpret_cast<size_t (*)(void *)>(dlsym(h, "ZSTD_freeDStream"));
                    decompressStream = reinterpret_cast<size_t (*)(void *, Buf *, Buf *)>(dlsym(h, "ZSTD_decompressStream"));
                    compress = reinterpret_cast<size_t (*)(void *, size_t, const void *, size_t, int)>(dlsym(h, "ZSTD_compress"));
                    compressBound = reinterpret_cast<size_t (*)(size_t)>(dlsym(h, "ZSTD_compressBound"));
                    isError = reinterpret_cast<unsigned (*)(size_t)>(dlsym(h, "ZSTD_isError"));
                    ok = createDStream && freeDStream && decompressStream && compress && compressBound && isError;
                }
            };
            const Zstd &zstd()
            {
                static const Zstd z;
                return z;
            }

            // the compressed payload of an outermost object -> its member bytes.  `limit` bounds the output (a hostile stream
            // must not inflate without end: the reference inflates on demand into the parser for the same reason)
            // A corrupt or truncated stream does not throw here: what could be inflated is returned and *failed is set, so that the
            // parser behaves like the reference's on-demand inflating stream buffer (ztools.cpp:200-300): it fails where the data
            // runs out ("I/O error") unless a member check rejects the object earlier.
            std::vector<uint8_t> decompress(const uint8_t *src, size_t n, uint8_t mode, size_t limit, bool *failed)
            {
                *failed = false;
                std::vector<uint8_t> out(std::min<size_t>(limit, std::max<size_t>(4 * n, 1 << 16)));
                size_t produced = 0;
                auto grow = [&]() {
                    if (out.size() >= limit)
                        return false;
                    out.resize(std::min<size_t>(limit, out.size() * 2));
                    return true;
                };
                if (mode == 1)
                {
                    z_stream zs;
                    std::memset(&zs, 0, sizeof(zs));
                    if (inflateInit(&zs) != Z_OK)
                    {
                        *failed = true;
                        return {};
                    }
                    zs.next_in = const_cast<Bytef *>(src);
                    size_t fed = 0;
                    int rc = Z_OK;
                    while (rc != Z_STREAM_END)
                    {
                        if (zs.avail_in == 0 && fed < n)
                        {
                            const size_t chunk = std::min<size_t>(n - fed, 1u << 30);
                            zs.next_in = const_cast<Bytef *>(src + fed);
                            zs.avail_in = (uInt)chunk;
                            fed += chunk;
                        }
                        if (produced == out.size() && !grow())
                        {
                            *failed = true;
                            break;
                        }
                        const size_t room = std::min<size_t>(out.size() - produced, 1u << 30);
                        zs.next_out = out.data() + produced;
                        zs.avail_out = (uInt)room;
                        rc = inflate(&zs, Z_NO_FLUSH);
                        produced += room - zs.avail_out;
                        if ((rc != Z_OK && rc != Z_STREAM_END && !(rc == Z_BUF_ERROR && (zs.avail_out == 0 || fed < n))) ||
                            (rc == Z_BUF_ERROR && zs.avail_out != 0 && fed >= n) /* truncated */)
                        {
                            *failed = true;
                            break;
                        }
                    }
                    inflateEnd(&zs);
                }
                else
                {
                    const Zstd &z = zstd();
                    void *ds = z.createDStream();
                    if (!ds)
                    {
                        *failed = true;
                        return {};
                    }
                    Zstd::Buf in{ const_cast<uint8_t *>(src), n, 0 };
                    size_t rc = 1;
                    while (rc != 0)
                    {
                        if (produced == out.size() && !grow())
                        {
                            *failed = true;
                            break;
                        }
                        Zstd::Buf ob{ out.data(), out.size(), produced };
                        rc = z.decompressStream(ds, &ob, &in);
                        const bool progressed = ob.pos != produced;
                        produced = ob.pos;
                        if (z.isError(rc) || (rc != 0 && in.pos == in.size && !progressed && produced < out.size()))
                        {
                            *failed = true;
                            break;
                        }
                    }
                    z.freeDStream(ds);
                }
                out.resize(produced);
                return out;
            }

            // Serialization::Load (serialization.cpp:341-553) around `members(reader, version)`
            template <class Fn>
            size_t framed(Reader &r, Fn members)
            {
                const size_t start = r.pos;
                Header h;
                r.read(&h, sizeof(h));
                if (!compatible_version(h))
                    throw std::logic_error("incompatible version");
                if (!valid_header(h) || h.size < sizeof(Header))
                    throw std::logic_error("loaded SEALHeader is invalid");
                // the two size checks run on seekable streams only (serialization.cpp:383-412, 437): the reference parses a
                // compressed payload from an inflating, non-seekable stream buffer, where a short payload surfaces as "I/O error"
                if (r.seekable && h.size > r.size - start)
                    throw std::invalid_argument("SEALHeader.size exceeds available input");
                if (h.compr_mode != 0)
                {
                    // serialization.cpp:430-515: the payload is one compressed stream of the member bytes; the members are parsed
                    // from the inflated buffer (which the images may point into: it is kept alive by the reader's owner)
                    const size_t payload = (size_t)h.size - sizeof(Header);
                    // a nested object inside an inflated payload is not covered by the seekable-only size check above: the
                    // reference's inflating stream buffer simply runs dry there ("I/O error"); here the payload is a pointer
                    // range, so the bound has to be explicit whatever the reader is
                    if (payload > r.size - r.pos)
                        throw std::runtime_error("I/O error");
                    bool failed = false;
                    r.inflated.emplace_back(decompress(r.base + r.pos, payload, h.compr_mode, r.inflate_limit, &failed));
                    Reader inner{ r.inflated.back().data(), r.inflated.back().size() };
                    inner.seekable = false;
                    inner.inflate_limit = r.inflate_limit;
                    members(inner, Version{ h.version_major, h.version_minor });
                    if (failed)
                        throw std::logic_error("stream decompression failed");
                    for (auto &b : inner.inflated)
                        r.inflated.emplace_back(std::move(b));
                    r.pos += payload;
                    return (size_t)h.size;
                }
                members(r, Version{ h.version_major, h.version_minor });
                if (r.seekable && h.size != r.pos - start)
                    throw std::logic_error("invalid data size");
                return (size_t)h.size;
            }

            // is_metadata_valid_for(const Ciphertext &, context, allow_pure_key_levels) (valcheck.cpp:20-78)
            bool metadata_valid(const Context &ctx, const Level *lvl, uint64_t n, uint64_t K, uint64_t size, double scale,
                                uint64_t correction_factor, bool allow_pure_key_levels)
            {
                if (!lvl)
                    return false;
                if (!allow_pure_key_levels && lvl->chain_index > ctx.first_level().chain_index)
                    return false;
                if (K != lvl->K || n != ctx.n())
                    return false;
                if ((size < 2 && size != 0) || size > 6) // SEAL_CIPHERTEXT_SIZE_MIN / _MAX (util/defines.h)
                    return false;
                const Scheme s = ctx.scheme();
                const bool positive_normal = std::isnormal(scale) && scale > 0;
                if ((scale != 1.0 && (s == Scheme::bfv || s == Scheme::bgv)) || (!positive_normal && s == Scheme::ckks))
                    return false;
                if ((correction_factor != 1 && (s == Scheme::bfv || s == Scheme::ckks)) ||
                    ((correction_factor == 0 || correction_factor >= ctx.plain_modulus()) && s == Scheme::bgv))
                    return false;
                return true;
            }
            // the coefficient range part of is_data_valid_for (valcheck.cpp:310-346)
            typedef uint64_t unaligned_u64 __attribute__((aligned(1)));
            bool data_in_range(const Context &ctx, const CiphertextImage &c)
            {
                const size_t n = ctx.n();
                const unaligned_u64 *p = reinterpret_cast<const unaligned_u64 *>(c.stored);
                size_t left = c.stored_words; // words still to come from the stored piece; then the expanded one
                for (uint64_t i = 0; i < c.size; i++)
                    for (unsigned j = 0; j < c.level->K; j++)
                    {
                        if (left == 0 && c.pending_words)
                            return true; // the rest is expanded on the device, reduced by construction
                        if (left == 0)
                        {
                            p = reinterpret_cast<const unaligned_u64 *>(c.expanded.data());
                            left = c.expanded.size();
                        }
                        const uint64_t q = ctx.coeff_modulus()[j];
                        uint64_t over = 0;
                        for (size_t k = 0; k < n; k++)
                            over |= (uint64_t)(p[k] >= q);
                        if (over)
                            return false;
                        p += n;
                        left -= n;
                    }
                return true;
            }

            // a seed expansion postponed by the caller (KSwitchKeys: one independent PRNG per digit, expanded on several threads)
            struct ExpandJob
            {
                Prng prng;
                size_t K, N;
                uint64_t *dst;
            };

            // Ciphertext::load_members (ciphertext.cpp:230-403)
            void ciphertext_members(const Context &ctx, Reader &r, Version v, CiphertextImage &out, std::vector<ExpandJob> *deferred = nullptr,
                                    bool device_expand = false)
            {
                uint64_t parms_id[4];
                r.read(parms_id, sizeof(parms_id));
                const uint8_t ntt_byte = r.get<uint8_t>();
                const uint64_t size64 = r.get<uint64_t>(), n64 = r.get<uint64_t>(), K64 = r.get<uint64_t>();
                const double scale = r.get<double>();
                uint64_t correction_factor = 1;
                if (v.major == 4)
                    correction_factor = r.get<uint64_t>();
                const Level *lvl = ctx.level_by_parms_id(parms_id);
                // pure key levels are allowed here: the same members serialize a PublicKey
                if (!metadata_valid(ctx, lvl, n64, K64, size64, scale, correction_factor, true))
                    throw std::logic_error("ciphertext data is invalid");
                out.level = lvl;
                out.is_ntt_form = ntt_byte != 0;
                out.size = size64;
                out.scale = scale;
                out.correction_factor = correction_factor;
                out.was_seeded = false;
                const uint64_t total = size64 * n64 * K64;
                // DynArray::load with in_size_bound = total, strict (dynarray.h:692-735)
                out.expanded.clear();
                framed(r, [&](Reader &rr, Version) {
                    const uint64_t count = rr.get<uint64_t>();
                    if (count > total)
                        throw std::logic_error("unexpected size");
                    out.stored = rr.base + rr.pos;
                    out.stored_words = (size_t)count;
                    rr.skip((size_t)count * sizeof(uint64_t));
                });
                const uint64_t seeded_count = n64 * K64;
                if (out.stored_words == seeded_count)
                {
                    // only c_0 was stored: c_1 is expanded from the seed that follows (ciphertext.cpp:118-151)
                    if (size64 != 2)
                        throw std::logic_error("ciphertext data is invalid");
                    if (!(v.major == 4 || (v.major == 3 && v.minor >= 6)))
                        throw std::logic_error("incompatible version"); // the 3.4 / 3.5 samplers are not restated here
                    Prng prng;
                    framed(r, [&](Reader &rr, Version) {
                        prng.type = rr.get<uint8_t>();
                        if (prng.type != 1 && prng.type != 2)
                            throw std::logic_error("prng_type is invalid");
                        rr.read(prng.seed, sizeof(prng.seed));
                    });
                    out.pending_words = 0;
                    // (the device kernel's conditions, xof.h: xof_device_ok)
                    if (device_expand && n64 >= 8 && seeded_count && (seeded_count * 8) % 4096 == 0)
                    {
                        out.pending_words = (size_t)seeded_count;
                        out.pending_type = prng.type;
                        std::memcpy(out.pending_seed, prng.seed, sizeof(prng.seed));
                    }
                    else
                        out.expanded.resize((size_t)seeded_count);
                    if (out.pending_words)
                        ;
                    else if (deferred)
                        deferred->push_back(ExpandJob{ prng, (size_t)K64, (size_t)n64, out.expanded.data() });
                    else
                        sample_poly_uniform(prng, ctx.coeff_modulus().data(), (size_t)K64, (size_t)n64, out.expanded.data());
                    out.was_seeded = true;
                }
                // is_buffer_valid (valcheck.cpp:180-196)
                if (out.word_count() != total)
                    throw std::logic_error("ciphertext data is invalid");
                // BGV stores coefficient form and transforms on load; the coefficients are range-checked first
                // (ciphertext.cpp:384-396), for unsafe_load as well
                if (ctx.scheme() == Scheme::bgv && !out.is_ntt_form && out.word_count() != 0)
                {
                    if (!metadata_valid(ctx, lvl, n64, K64, size64, scale, correction_factor, false) || !data_in_range(ctx, out))
                        throw std::logic_error("ciphertext data is invalid");
                }
            }

            void check_input(const uint8_t *in, size_t size)
            {
                // Serialization::Load(load_members, in, size) (serialization.cpp:559-575)
                if (!in)
                    throw std::invalid_argument("in cannot be null");
                if (size < sizeof(Header))
                    throw std::invalid_argument("insufficient size");
                if (size > (size_t)std::numeric_limits<std::streamsize>::max())
                    throw std::invalid_argument("size is too large");
            }
        } // namespace

        void CiphertextImage::copy_words(uint64_t *dst) const
        {
            if (stored_words)
                std::memcpy(dst, stored, stored_words * 8);
            if (!expanded.empty())
                std::memcpy(dst + stored_words, expanded.data(), expanded.size() * 8);
        }

        void expand_seed_blake2xb(const uint64_t *seed, const uint64_t *primes, size_t K, size_t N, uint64_t *destination)
        {
            Prng prng(1, seed);
            sample_poly_uniform(prng, primes, K, N, destination);
        }

        size_t load_ciphertext(const Context &ctx, const uint8_t *in, size_t size, bool check_data, CiphertextImage &out, bool device_expand)
        {
            check_input(in, size);
            Reader r{ in, size };
            r.inflate_limit = 6 * ctx.key_level().K * ctx.n() * 8 + 4096; // SEAL_CIPHERTEXT_SIZE_MAX polynomials at the key level
            CiphertextImage img;
            const size_t bytes = framed(r, [&](Reader &rr, Version v) { ciphertext_members(ctx, rr, v, img, nullptr, device_expand); });
            img.inflated = std::move(r.inflated);
            if (check_data)
            {
                // Ciphertext::load = unsafe_load + is_valid_for (ciphertext.h:533-545; valcheck.cpp): data levels only,
                // every coefficient reduced
                if (!metadata_valid(ctx, img.level, ctx.n(), img.level->K, img.size, img.scale, img.correction_factor, false) ||
                    !data_in_range(ctx, img))
                    throw std::logic_error("ciphertext data is invalid");
            }
            out = std::move(img);
            return bytes;
        }

        size_t load_kswitchkeys(const Context &ctx, const uint8_t *in, size_t size, bool check_data, KSwitchKeysImage &out, bool device_expand)
        {
            check_input(in, size);
            Reader r{ in, size };
            {
                // bound on one inflated payload: at most n key slots (GaloisKeys) of first_K digits, each a size-2 ciphertext at
                // the key level with its own headers; a stream that inflates beyond what any valid object holds is refused
                // before the host allocates for it
                const size_t n = ctx.n(), first_K = ctx.first_level().K, L = ctx.key_level().K;
                const size_t one_ct = 2 * L * n * 8 + 256;
                size_t limit = 64 + n * 8;
                const size_t keys_max = n * first_K; // cannot overflow: n <= 2^17, first_K <= 64
                limit += keys_max * one_ct;
                // and no valid key stream (uniform residues do not compress) inflates by more than deflate's maximum ratio
                const size_t by_ratio = std::max<size_t>(size_t(1) << 20, size > (size_t(1) << 50) ? ~size_t(0) : size * 1100);
                r.inflate_limit = std::min(std::min(limit, by_ratio), size_t(1) << 40);
            }
            KSwitchKeysImage img;
            uint64_t parms_id[4] = { 0, 0, 0, 0 };
            std::vector<ExpandJob> jobs; // the heap buffers they point into do not move when the images are moved
            const size_t bytes = framed(r, [&](Reader &rr, Version) {
                // KSwitchKeys::load_members (kswitchkeys.cpp:92-180)
                rr.read(parms_id, sizeof(parms_id));
                const uint64_t dim1 = rr.get<uint64_t>();
                if (dim1 > ctx.n())
                    throw std::logic_error("KSwitchKeys outer dimension is invalid");
                const uint64_t max_dim2 = ctx.first_level().K;
                img.keys.reserve((size_t)dim1);
                for (uint64_t i = 0; i < dim1; i++)
                {
                    const uint64_t dim2 = rr.get<uint64_t>();
                    if (dim2 > max_dim2)
                        throw std::logic_error("KSwitchKeys inner dimension is invalid");
                    img.keys.emplace_back();
                    img.keys.back().reserve((size_t)dim2);
                    for (uint64_t j = 0; j < dim2; j++)
                    {
                        CiphertextImage key;
                        framed(rr, [&](Reader &r3, Version v) { ciphertext_members(ctx, r3, v, key, &jobs, device_expand); });
                        img.keys.back().emplace_back(std::move(key));
                    }
                }
            });
            // seeded keys: every digit has its own seed, so the expansions (BLAKE2Xb runs at ~0.3 GB/s per host core; a C5 key is
            // 126 MB of it) are independent and spread over the host cores
            if (!jobs.empty())
            {
                unsigned nthreads = std::thread::hardware_concurrency();
                if (nthreads > 16)
                    nthreads = 16;
                if (nthreads > jobs.size())
                    nthreads = (unsigned)jobs.size();
                if (nthreads < 1)
                    nthreads = 1;
                std::atomic<size_t> next{ 0 };
                for (auto &j : jobs)
                    j.prng.parallel = nthreads == 1; // one level of threading: across the digits here, inside a draw otherwise
                auto work = [&]() {
                    for (size_t i = next++; i < jobs.size(); i = next++)
                        sample_poly_uniform(jobs[i].prng, ctx.coeff_modulus().data(), jobs[i].K, jobs[i].N, jobs[i].dst);
                };
                std::vector<std::thread> pool;
                for (unsigned t = 1; t < nthreads; t++)
                    pool.emplace_back(work);
                work();
                for (auto &t : pool)
                    t.join();
            }
            // What the device representation needs regardless of `check_data` (the reference's unsafe_load defers these to
            // the first use, where they surface as invalid_argument / logic_error): key-level, NTT-form, size-2 digits
            const Level *key_level = &ctx.key_level();
            bool structure_ok = ctx.level_by_parms_id(parms_id) == key_level;
            for (auto &a : img.keys)
            {
                if (!a.empty() && a.size() != ctx.first_level().K)
                    structure_ok = false;
                for (auto &b : a)
                    if (b.level != key_level || !b.is_ntt_form || b.size != 2)
                        structure_ok = false;
            }
            if (!structure_ok)
                throw std::logic_error("KSwitchKeys data is invalid");
            if (check_data)
            {
                // KSwitchKeys::load = unsafe_load + is_valid_for (kswitchkeys.h:236-246; valcheck.cpp)
                for (auto &a : img.keys)
                    for (auto &b : a)
                        if (!data_in_range(ctx, b))
                            throw std::logic_error("KSwitchKeys data is invalid");
            }
            img.inflated = std::move(r.inflated);
            out = std::move(img);
            return bytes;
        }

        size_t load_plaintext(const Context &ctx, const uint8_t *in, size_t size, bool check_data, PlaintextImage &out)
        {
            check_input(in, size);
            Reader r{ in, size };
            r.inflate_limit = ctx.key_level().K * ctx.n() * 8 + 4096;
            PlaintextImage img;
            // is_metadata_valid_for(const Plaintext &, context, allow_pure_key_levels) (valcheck.cpp:80-133)
            auto metadata_ok = [&](bool allow_pure_key_levels) {
                if (img.level)
                {
                    if (!allow_pure_key_levels && img.level->chain_index > ctx.first_level().chain_index)
                        return false;
                    if ((uint64_t)img.level->K * ctx.n() != img.coeff_count)
                        return false;
                }
                else if (img.coeff_count > ctx.n())
                    return false;
                if (ctx.scheme() == Scheme::ckks && !(std::isnormal(img.scale) && img.scale > 0))
                    return false;
                return true;
            };
            const size_t bytes = framed(r, [&](Reader &rr, Version) {
                // Plaintext::load_members (plaintext.cpp)
                uint64_t parms_id[4];
                rr.read(parms_id, sizeof(parms_id));
                img.coeff_count = rr.get<uint64_t>();
                img.scale = rr.get<double>();
                const bool ntt_form = parms_id[0] || parms_id[1] || parms_id[2] || parms_id[3];
                img.level = ntt_form ? ctx.level_by_parms_id(parms_id) : nullptr;
                if ((ntt_form && !img.level) || !metadata_ok(true))
                    throw std::logic_error("plaintext data is invalid");
                uint64_t count = 0;
                framed(rr, [&](Reader &r3, Version) {
                    count = r3.get<uint64_t>();
                    if (count > img.coeff_count)
                        throw std::logic_error("unexpected size");
                    img.stored = r3.base + r3.pos;
                    r3.skip((size_t)count * sizeof(uint64_t));
                });
                if (count != img.coeff_count) // is_buffer_valid
                    throw std::logic_error("plaintext data is invalid");
            });
            // Plaintext::load = unsafe_load + is_valid_for (is_data_valid_for, valcheck.cpp:348-396)
            if (check_data && !(metadata_ok(false) && plaintext_in_range(ctx, img)))
                throw std::logic_error("plaintext data is invalid");
            img.inflated = std::move(r.inflated);
            out = std::move(img);
            return bytes;
        }

        bool plaintext_in_range(const Context &ctx, const PlaintextImage &img)
        {
            const unaligned_u64 *p = reinterpret_cast<const unaligned_u64 *>(img.stored);
            if (img.level)
            {
                for (unsigned j = 0; j < img.level->K; j++)
                {
                    const uint64_t q = ctx.coeff_modulus()[j];
                    uint64_t over = 0;
                    for (size_t k = 0; k < ctx.n(); k++)
                        over |= (uint64_t)(p[k] >= q);
                    if (over)
                        return false;
                    p += ctx.n();
                }
                return true;
            }
            const uint64_t t = ctx.plain_modulus();
            for (uint64_t k = 0; k < img.coeff_count; k++)
                if (p[k] >= t)
                    return false;
            return true;
        }

        size_t load_encryption_parameters(const uint8_t *in, size_t size, uint8_t &scheme, uint64_t &poly_modulus_degree,
                                          std::vector<uint64_t> &coeff_modulus, uint64_t &plain_modulus)
        {
            if (!in)
                throw std::invalid_argument("in cannot be null");
            if (size < sizeof(Header))
                throw std::invalid_argument("insufficient size");
            Reader r{ in, size };
            return framed(r, [&](Reader &m, Version) {
                const uint8_t sch = m.get<uint8_t>();
                if (sch > 3) // EncryptionParameters(uint8_t scheme) (encryptionparams.h:166-175)
                    throw std::invalid_argument("unsupported scheme");
                const uint64_t n = m.get<uint64_t>();
                if (n > 131072) // SEAL_POLY_MOD_DEGREE_MAX
                    throw std::logic_error("poly_modulus_degree is invalid");
                const uint64_t k = m.get<uint64_t>();
                if (k > kMaxComps) // SEAL_COEFF_MOD_COUNT_MAX
                    throw std::logic_error("coeff_modulus is invalid");
                // Modulus::load -> Modulus::set_value (modulus.cpp:71-93): at most 61 bits and not 1
                auto modulus = [&](Reader &mm) {
                    uint64_t v = 0;
                    framed(mm, [&](Reader &inner, Version) { v = inner.get<uint64_t>(); });
                    if ((v >> 61) != 0 || v == 1)
                        throw std::invalid_argument("value can be at most 61-bit and cannot be 1");
                    return v;
                };
                std::vector<uint64_t> q;
                for (uint64_t i = 0; i < k; i++)
                    q.push_back(modulus(m));
                const uint64_t t = modulus(m);
                // set_poly_modulus_degree / set_coeff_modulus / set_plain_modulus (encryptionparams.h:190-262)
                if (sch == 0 && (n || k))
                    throw std::logic_error(n ? "poly_modulus_degree is not supported for this scheme" : "coeff_modulus is not supported for this scheme");
                if (sch != 0 && k < 1)
                    throw std::invalid_argument("coeff_modulus is invalid");
                if ((sch == 0 || sch == 2) && t != 0)
                    throw std::logic_error("plain_modulus is not supported for this scheme");
                scheme = sch;
                poly_modulus_degree = n;
                coeff_modulus = q;
                plain_modulus = t;
            });
        }

        size_t plaintext_save_size(uint64_t coeff_count)
        {
            return sizeof(Header) + 32 + 8 + 8 + sizeof(Header) + 8 + (size_t)coeff_count * 8;
        }

        size_t save_plaintext(const uint64_t *parms_id, uint64_t coeff_count, double scale, const uint64_t *words, uint8_t *out,
                              size_t capacity, size_t *data_offset)
        {
            if (!out)
                throw std::invalid_argument("out cannot be null");
            if (capacity < sizeof(Header))
                throw std::invalid_argument("insufficient size");
            const size_t total = plaintext_save_size(coeff_count);
            if (capacity < total)
                throw std::runtime_error("I/O error");
            uint8_t *p = out;
            auto put = [&](const void *src, size_t bytes) {
                std::memcpy(p, src, bytes);
                p += bytes;
            };
            Header h{ kMagic, kHeaderSize, kVersionMajor, kVersionMinor, 0, 0, (uint64_t)total };
            put(&h, sizeof(h));
            put(parms_id, 32);
            put(&coeff_count, 8);
            put(&scale, 8);
            Header hd{ kMagic, kHeaderSize, kVersionMajor, kVersionMinor, 0, 0, (uint64_t)(sizeof(Header) + 8 + coeff_count * 8) };
            put(&hd, sizeof(hd));
            put(&coeff_count, 8);
            if (data_offset)
                *data_offset = (size_t)(p - out);
            if (coeff_count && words)
                put(words, (size_t)coeff_count * 8);
            return total;
        }

        bool compr_mode_supported(uint8_t compr_mode)
        {
            return compr_mode == 0 || compr_mode == 1 || (compr_mode == 2 && zstd().ok);
        }
        size_t compress_bound(size_t raw_bytes, uint8_t compr_mode)
        {
            if (!compr_mode_supported(compr_mode))
                throw std::invalid_argument("unsupported compression mode");
            if (compr_mode == 0)
                return raw_bytes;
            const size_t payload = raw_bytes - sizeof(Header);
            if (compr_mode == 1)
                return sizeof(Header) + payload + (payload >> 12) + (payload >> 14) + (payload >> 25) + 64; // deflateBound
            return sizeof(Header) + zstd().compressBound(payload);
        }
        size_t compress_stream(const uint8_t *raw, size_t raw_bytes, uint8_t compr_mode, uint8_t *out, size_t capacity)
        {
            // Serialization::Save (serialization.cpp:232-340): header in the clear, the member bytes as one compressed stream
            if (!compr_mode_supported(compr_mode))
                throw std::invalid_argument("unsupported compression mode");
            if (!out)
                throw std::invalid_argument("out cannot be null");
            if (capacity < sizeof(Header))
                throw std::invalid_argument("insufficient size");
            const uint8_t *payload = raw + sizeof(Header);
            const size_t n = raw_bytes - sizeof(Header), room = capacity - sizeof(Header);
            size_t produced = 0;
            if (compr_mode == 0)
            {
                if (room < n)
                    throw std::runtime_error("I/O error");
                std::memcpy(out + sizeof(Header), payload, n);
                produced = n;
            }
            else if (compr_mode == 1)
            {
                z_stream zs;
                std::memset(&zs, 0, sizeof(zs));
                if (deflateInit(&zs, Z_DEFAULT_COMPRESSION) != Z_OK)
                    throw std::logic_error("stream compression failed");
                size_t fed = 0;
                int rc = Z_OK;
                while (rc != Z_STREAM_END)
                {
                    if (zs.avail_in == 0 && fed < n)
                    {
                        const size_t chunk = std::min<size_t>(n - fed, 1u << 30);
                        zs.next_in = const_cast<Bytef *>(payload + fed);
                        zs.avail_in = (uInt)chunk;
                        fed += chunk;
                    }
                    const size_t space = std::min<size_t>(room - produced, 1u << 30);
                    if (space == 0)
                    {
                        deflateEnd(&zs);
                        throw std::runtime_error("I/O error");
                    }
                    zs.next_out = out + sizeof(Header) + produced;
                    zs.avail_out = (uInt)space;
                    rc = deflate(&zs, fed < n ? Z_NO_FLUSH : Z_FINISH);
                    produced += space - zs.avail_out;
                    if (rc != Z_OK && rc != Z_STREAM_END && rc != Z_BUF_ERROR)
                    {
                        deflateEnd(&zs);
                        throw std::logic_error("stream compression failed");
                    }
                }
                deflateEnd(&zs);
            }
            else
            {
                const size_t rc = zstd().compress(out + sizeof(Header), room, payload, n, 3 /* ZSTD_CLEVEL_DEFAULT */);
                if (zstd().isError(rc))
                    throw std::runtime_error("I/O error");
                produced = rc;
            }
            Header h;
            std::memcpy(&h, raw, sizeof(h));
            h.compr_mode = compr_mode;
            h.size = sizeof(Header) + produced;
            std::memcpy(out, &h, sizeof(h));
            return sizeof(Header) + produced;
        }

        size_t seeded_ciphertext_save_size(uint64_t n, uint64_t K)
        {
            const size_t members = 4 * 8 + 1 + 3 * 8 + 8 + 8;
            const size_t dyn = sizeof(Header) + 8 + (size_t)(n * K) * 8;
            const size_t info = sizeof(Header) + 1 + 64;
            return sizeof(Header) + members + dyn + info;
        }

        size_t save_seeded_ciphertext(const uint64_t *parms_id, bool is_ntt_form, uint64_t n, uint64_t K, double scale,
                                      uint64_t correction_factor, const uint64_t *c0_words, uint8_t prng_type, const uint64_t *seed,
                                      uint8_t *out, size_t capacity, size_t *data_offset)
        {
            if (!out)
                throw std::invalid_argument("out cannot be null");
            if (capacity < sizeof(Header))
                throw std::invalid_argument("insufficient size");
            const size_t total = seeded_ciphertext_save_size(n, K);
            if (capacity < total)
                throw std::runtime_error("I/O error");
            uint8_t *p = out;
            auto put = [&](const void *src, size_t bytes) {
                std::memcpy(p, src, bytes);
                p += bytes;
            };
            Header h{ kMagic, kHeaderSize, kVersionMajor, kVersionMinor, 0, 0, (uint64_t)total };
            put(&h, sizeof(h));
            put(parms_id, 32);
            const uint8_t ntt = is_ntt_form ? 1 : 0;
            put(&ntt, 1);
            const uint64_t size = 2;
            put(&size, 8);
            put(&n, 8);
            put(&K, 8);
            put(&scale, 8);
            put(&correction_factor, 8);
            const uint64_t count = n * K;
            Header hd{ kMagic, kHeaderSize, kVersionMajor, kVersionMinor, 0, 0, (uint64_t)(sizeof(Header) + 8 + count * 8) };
            put(&hd, sizeof(hd));
            put(&count, 8);
            if (data_offset)
                *data_offset = (size_t)(p - out);
            if (c0_words)
                std::memcpy(p, c0_words, (size_t)count * 8);
            p += (size_t)count * 8;
            Header hi{ kMagic, kHeaderSize, kVersionMajor, kVersionMinor, 0, 0, (uint64_t)(sizeof(Header) + 1 + 64) };
            put(&hi, sizeof(hi));
            put(&prng_type, 1);
            put(seed, 64);
            return total;
        }

        size_t ciphertext_save_size(uint64_t size, uint64_t n, uint64_t K)
        {
            // Ciphertext::save_size(compr_mode_type::none) (ciphertext.cpp:153-186): members + the DynArray's own frame
            const size_t members = 4 * 8 + 1 + 3 * 8 + 8 + 8;
            const size_t dyn = sizeof(Header) + 8 + (size_t)(size * n * K) * 8;
            return sizeof(Header) + members + dyn;
        }

        size_t save_ciphertext(const uint64_t *parms_id, bool is_ntt_form, uint64_t size, uint64_t n, uint64_t K, double scale,
                               uint64_t correction_factor, const uint64_t *words, uint8_t *out, size_t capacity, size_t *data_offset)
        {
            // Serialization::Save(save_members, raw_size, out, size, compr_mode) (serialization.cpp:232-340, 541-557)
            if (!out)
                throw std::invalid_argument("out cannot be null");
            if (capacity < sizeof(Header))
                throw std::invalid_argument("insufficient size");
            const size_t total = ciphertext_save_size(size, n, K);
            if (capacity < total)
                throw std::runtime_error("I/O error"); // the reference's ArrayPutBuffer overflows: ios failure
            uint8_t *p = out;
            auto put = [&](const void *src, size_t bytes) {
                std::memcpy(p, src, bytes);
                p += bytes;
            };
            Header h{ kMagic, kHeaderSize, kVersionMajor, kVersionMinor, 0, 0, (uint64_t)total };
            put(&h, sizeof(h));
            put(parms_id, 32);
            const uint8_t ntt = is_ntt_form ? 1 : 0;
            put(&ntt, 1);
            put(&size, 8);
            put(&n, 8);
            put(&K, 8);
            put(&scale, 8);
            put(&correction_factor, 8);
            const uint64_t count = size * n * K;
            Header hd{ kMagic, kHeaderSize, kVersionMajor, kVersionMinor, 0, 0, (uint64_t)(sizeof(Header) + 8 + count * 8) };
            put(&hd, sizeof(hd));
            put(&count, 8);
            if (data_offset)
                *data_offset = (size_t)(p - out);
            if (count && words)
                put(words, (size_t)count * 8);
            return total;
        }
    } // namespace serial
} // namespace sealhip
