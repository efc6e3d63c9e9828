// See ckks_encoder.h.  Reference: native/src/seal/ckks.h, ckks.cpp, util/croots.cpp, util/rns.cpp (RNSBase).
#include "ckks_encoder.h"
#include "hostmath.h"
#include <algorithm>
#include <cmath>
#include <complex>
#include <cstring>

namespace sealhip
{
    namespace
    {
        void ck(hipError_t e, const char *what)
        {
            if (e != hipSuccess)
                throw std::runtime_error(std::string("HIP failure in ") + what + ": " + hipGetErrorString(e));
        }
        // util::ComplexRoots (croots.cpp:12-70)
        struct ComplexRoots
        {
            size_t degree;
            std::vector<std::complex<double>> roots;
            explicit ComplexRoots(size_t degree_of_roots) : degree(degree_of_roots), roots(degree_of_roots / 8 + 1)
            {
                constexpr double PI_ = 3.1415926535897932384626433832795028842;
                // std::polar(1.0, theta) = (cos theta, sin theta).  The reference is compiled with GCC, which fuses the two calls
                // into one sincos(); glibc's sincos and its separate sin / cos are not bit-identical for every argument, and
                // clang (hipcc's host compiler) does not fuse - so the fused call is made explicitly here.
                for (size_t i = 0; i <= degree / 8; i++)
                {
                    const double theta = 2 * PI_ * static_cast<double>(i) / static_cast<double>(degree);
                    double sn, cs;
                    ::sincos(theta, &sn, &cs);
                    roots[i] = std::complex<double>(1.0 * cs, 1.0 * sn);
                }
            }
            std::complex<double> get_root(size_t index) const
            {
                index &= degree - 1;
                auto mirror = [](std::complex<double> a) { return std::complex<double>{ a.imag(), a.real() }; };
                if (index <= degree / 8)
                    return roots[index];
                if (index <= degree / 4)
                    return mirror(roots[degree / 4 - index]);
                if (index <= degree / 2)
                    return -std::conj(get_root(degree / 2 - index));
                if (index <= 3 * degree / 4)
                    return -get_root(index - degree / 2);
                return std::conj(get_root(degree - index));
            }
        };
        uint32_t reverse_bits(uint64_t v, int bits)
        {
            uint64_t r = 0;
            for (int b = 0; b < bits; b++)
                r |= ((v >> b) & 1) << (bits - 1 - b);
            return (uint32_t)r;
        }
    } // namespace

    CKKSEncoder::CKKSEncoder(const Context &context) : context_(context)
    {
        if (context.scheme() != Scheme::ckks)
            throw std::invalid_argument("unsupported scheme");
        const size_t n = context.n();
        const int logn = context.log_n();
        slots_ = n >> 1;
        std::vector<uint32_t> map(n);
        const uint64_t m = (uint64_t)n << 1;
        uint64_t pos = 1;
        for (size_t i = 0; i < slots_; i++)
        {
            map[i] = reverse_bits((pos - 1) >> 1, logn);
            map[slots_ | i] = reverse_bits((m - pos - 1) >> 1, logn);
            pos = (pos * 3) & (m - 1);
        }
        std::vector<std::complex<double>> rp(n), irp(n);
        if (m >= 8)
        {
            ComplexRoots cr((size_t)m);
            for (size_t i = 1; i < n; i++)
            {
                rp[i] = cr.get_root(reverse_bits(i, logn));
                irp[i] = std::conj(cr.get_root((size_t)reverse_bits(i - 1, logn) + 1));
            }
        }
        else if (m == 4)
        {
            rp[1] = { 0, 1 };
            irp[1] = { 0, -1 };
        }
        ck(hipMalloc(reinterpret_cast<void **>(&map_), n * 4), "hipMalloc ckks map");
        ck(hipMalloc(reinterpret_cast<void **>(&roots_), n * 16), "hipMalloc ckks roots");
        ck(hipMalloc(reinterpret_cast<void **>(&inv_roots_), n * 16), "hipMalloc ckks roots");
        ck(hipMemcpy(map_, map.data(), n * 4, hipMemcpyHostToDevice), "upload ckks map");
        ck(hipMemcpy(roots_, rp.data(), n * 16, hipMemcpyHostToDevice), "upload ckks roots");
        ck(hipMemcpy(inv_roots_, irp.data(), n * 16, hipMemcpyHostToDevice), "upload ckks roots");
    }
    CKKSEncoder::~CKKSEncoder()
    {
        (void)hipFree(map_);
        (void)hipFree(roots_);
        (void)hipFree(inv_roots_);
        for (auto &kv : consts_)
            (void)hipFree(kv.second.dev);
    }

    uint64_t *build_crt_constants(const Context &context, const Level &lvl)
    {
        const unsigned K = lvl.K;
        std::vector<uint64_t> q(context.coeff_modulus().begin(), context.coeff_modulus().begin() + K);
        std::vector<uint64_t> block((size_t)K * K + 2 * K + 2 * K, 0);
        for (unsigned j = 0; j < K; j++)
        {
            std::vector<uint64_t> others;
            for (unsigned i = 0; i < K; i++)
                if (i != j)
                    others.push_back(q[i]);
            std::vector<uint64_t> pp = others.empty() ? std::vector<uint64_t>{ 1 } : host::product(others);
            for (size_t w = 0; w < pp.size() && w < K; w++)
                block[(size_t)j * K + w] = pp[w];
            uint64_t pm = 1 % q[j];
            for (unsigned i = 0; i < K; i++)
                if (i != j)
                    pm = host::mulmod(pm, q[i] % q[j], q[j]);
            const ShoupOp ip = host::make_shoup(host::invmod(pm, q[j]), q[j]);
            block[(size_t)K * K + 2 * K + 2 * j] = ip.w;
            block[(size_t)K * K + 2 * K + 2 * j + 1] = ip.wq;
        }
        std::vector<uint64_t> Q = host::product(q);
        Q.resize(K, 0);
        // (Q + 1) >> 1
        std::vector<uint64_t> q1(K), half(K);
        uint64_t carry = 1;
        for (unsigned w = 0; w < K; w++)
        {
            q1[w] = Q[w] + carry;
            carry = (carry && q1[w] == 0) ? 1 : 0;
        }
        for (unsigned w = 0; w < K; w++)
            half[w] = (q1[w] >> 1) | (w + 1 < K ? q1[w + 1] << 63 : (carry << 63));
        for (unsigned w = 0; w < K; w++)
        {
            block[(size_t)K * K + w] = Q[w];
            block[(size_t)K * K + K + w] = half[w];
        }
        uint64_t *dev = nullptr;
        ck(hipMalloc(reinterpret_cast<void **>(&dev), block.size() * 8), "hipMalloc crt constants");
        ck(hipMemcpy(dev, block.data(), block.size() * 8, hipMemcpyHostToDevice), "upload crt constants");
        return dev;
    }

    const CKKSEncoder::LevelConst &CKKSEncoder::level_const(const Level &lvl) const
    {
        std::lock_guard<std::mutex> lock(mu_);
        auto it = consts_.find(lvl.chain_index);
        if (it != consts_.end())
            return it->second;
        LevelConst lc;
        lc.dev = build_crt_constants(context_, lvl);
        return consts_.emplace(lvl.chain_index, lc).first->second;
    }

    void CKKSEncoder::encode(const double *values, size_t count, bool is_complex, const uint64_t *parms_id, double scale, Plaintext &dest) const
    {
        // ckks.h:458-680
        const Level *lvl = parms_id ? context_.level_by_parms_id(parms_id) : nullptr;
        if (!lvl)
            throw std::invalid_argument("parms_id is not valid for encryption parameters");
        if (!values && count > 0)
            throw std::invalid_argument("values cannot be null");
        if (count > slots_)
            throw std::invalid_argument("values_size is too large");
        if (&dest.context() != &context_)
            throw std::invalid_argument("destination belongs to another context");
        if (!std::isnormal(scale) || scale <= 0 || (static_cast<int>(std::log2(scale)) + 1 >= lvl->total_coeff_modulus_bit_count))
            throw std::invalid_argument("scale out of bounds");
        for (size_t i = 0; i < (is_complex ? 2 * count : count); i++)
            if (!std::isfinite(values[i]))
                throw std::invalid_argument("values must be finite");
        const size_t n = context_.n(), K = lvl->K;
        const unsigned n_log = (unsigned)context_.log_n();
        std::vector<double> in(2 * count, 0.0);
        for (size_t i = 0; i < count; i++)
        {
            in[2 * i] = is_complex ? values[2 * i] : values[i];
            in[2 * i + 1] = is_complex ? values[2 * i + 1] : 0.0;
        }
        Scratch vin(count ? 2 * count : 1), conj(2 * n), aux(2);
        double2 *cv = reinterpret_cast<double2 *>(conj.p);
        ck(hipStreamSynchronize(nullptr), "encode sync");
        ck(hipMemsetAsync(cv, 0, n * 16, nullptr), "zero values");
        if (count)
        {
            ck(hipMemcpy(vin.p, in.data(), 2 * count * 8, hipMemcpyHostToDevice), "upload values");
            ck(k_ckks_place(map_, reinterpret_cast<const double2 *>(vin.p), cv, n_log, (unsigned)count, nullptr), "place values");
        }
        // fft_handler_.transform_from_rev(conj_values, logn, inv_root_powers_, &fix) with fix = scale / n
        const double fix = scale / static_cast<double>(n);
        ck(hipMemcpy(aux.p, &fix, 8, hipMemcpyHostToDevice), "upload fix");
        for (unsigned g = 0; g + 1 < n_log; g++)
            ck(k_fft_gs_stage(cv, inv_roots_, n_log, g, 1, nullptr, nullptr), "fft stage");
        ck(k_fft_gs_stage(cv, inv_roots_, n_log, n_log - 1, 1, reinterpret_cast<const double *>(aux.p), nullptr), "fft last stage");
        // the largest coefficient decides the arithmetic width (ckks.h:525-548)
        ck(hipMemsetAsync(aux.p + 1, 0, 8, nullptr), "zero max");
        ck(k_max_abs_real(cv, n, reinterpret_cast<unsigned long long *>(aux.p + 1), nullptr), "max coefficient");
        double max_coeff;
        ck(hipMemcpy(&max_coeff, aux.p + 1, 8, hipMemcpyDeviceToHost), "download max");
        if (std::isnan(max_coeff) || !std::isfinite(max_coeff))
            throw std::invalid_argument("encoded values are too large");
        const double mc = std::max<>(max_coeff, 1.0);
        const int max_coeff_bit_count = static_cast<int>(std::ceil(std::log2(mc))) + 1; // util::safe_ceil_log2_int
        if (max_coeff_bit_count >= lvl->total_coeff_modulus_bit_count)
            throw std::invalid_argument("encoded values are too large");
        uint64_t *slab = DevicePool::global().alloc_words(K * n);
        try
        {
            ck(k_ckks_decompose(context_.dev_mods(), cv, slab, n_log, (unsigned)K, 1, max_coeff_bit_count <= 64 ? 64 : (max_coeff_bit_count <= 128 ? 128 : 0), nullptr), "decompose");
            NttBatch b{};
            b.data = slab;
            b.outer_stride = K * n;
            b.ncomp = (unsigned)K;
            b.nouter = 1;
            b.prime_first = 0;
            ck(ntt_forward(context_.ntt_tables(), b, 0, nullptr), "ntt plaintext");
            ck(hipStreamSynchronize(nullptr), "encode sync");
        }
        catch (...)
        {
            DevicePool::global().free_words(slab);
            throw;
        }
        dest.adopt(slab, K * n, K * n);
        dest.set_level(lvl);
        dest.scale() = scale;
    }

    void CKKSEncoder::fill_constant(const Level &lvl, const std::vector<uint64_t> &residues, double scale, Plaintext &dest) const
    {
        if (&dest.context() != &context_)
            throw std::invalid_argument("destination belongs to another context");
        const size_t n = context_.n(), K = lvl.K;
        std::vector<uint64_t> words(K * n);
        for (size_t j = 0; j < K; j++)
            std::fill_n(words.begin() + j * n, n, residues[j]);
        uint64_t *slab = DevicePool::global().alloc_words(K * n);
        hipError_t e = hipStreamSynchronize(nullptr);
        if (e == hipSuccess)
            e = hipMemcpy(slab, words.data(), K * n * 8, hipMemcpyHostToDevice);
        if (e != hipSuccess)
        {
            DevicePool::global().free_words(slab);
            ck(e, "upload constant plaintext");
        }
        dest.adopt(slab, K * n, K * n);
        dest.set_level(&lvl);
        dest.scale() = scale;
    }
    void CKKSEncoder::encode_value(double value, const uint64_t *parms_id, double scale, Plaintext &dest) const
    {
        // ckks.cpp:72-205
        const Level *lvl = parms_id ? context_.level_by_parms_id(parms_id) : nullptr;
        if (!lvl)
            throw std::invalid_argument("parms_id is not valid for encryption parameters");
        if (!std::isfinite(scale) || scale <= 0 || (static_cast<int>(std::log2(scale)) >= lvl->total_coeff_modulus_bit_count))
            throw std::invalid_argument("scale out of bounds");
        if (!std::isfinite(value))
            throw std::invalid_argument("value must be finite");
        value *= scale;
        if (!std::isfinite(value))
            throw std::invalid_argument("encoded value is too large");
        const int coeff_bit_count = (std::fabs(value) < 1.0) ? 2 : (static_cast<int>(std::log2(std::fabs(value))) + 2);
        if (coeff_bit_count >= lvl->total_coeff_modulus_bit_count)
            throw std::invalid_argument("encoded value is too large");
        const double two_pow_64 = std::pow(2.0, 64);
        double coeffd = std::round(value);
        const bool is_negative = std::signbit(coeffd);
        coeffd = std::fabs(coeffd);
        std::vector<uint64_t> residues(lvl->K);
        for (unsigned j = 0; j < lvl->K; j++)
        {
            const uint64_t q = context_.coeff_modulus()[j];
            uint64_t r;
            if (coeff_bit_count <= 64)
                r = static_cast<uint64_t>(std::fabs(coeffd)) % q;
            else if (coeff_bit_count > 128)
            {
                // ckks.cpp:165-196: the double cut into 64-bit words (fmod / division by 2^64, exact), reduced modulo q
                // (RNSBase::decompose); Horner from the top word
                std::vector<uint64_t> words;
                for (double c = coeffd; c >= 1 && words.size() < lvl->K; c /= two_pow_64)
                    words.push_back(static_cast<uint64_t>(std::fmod(c, two_pow_64)));
                unsigned __int128 acc = 0;
                for (size_t w = words.size(); w-- > 0;)
                    acc = ((acc << 64) | words[w]) % q;
                r = (uint64_t)acc;
            }
            else
            {
                const unsigned __int128 v = ((unsigned __int128) static_cast<uint64_t>(coeffd / two_pow_64) << 64) |
                                            static_cast<uint64_t>(std::fmod(coeffd, two_pow_64));
                r = (uint64_t)(v % q);
            }
            residues[j] = is_negative ? (r ? q - r : 0) : r;
        }
        fill_constant(*lvl, residues, scale, dest);
    }
    void CKKSEncoder::encode_integer(int64_t value, const uint64_t *parms_id, Plaintext &dest) const
    {
        // ckks.cpp:207-250
        const Level *lvl = parms_id ? context_.level_by_parms_id(parms_id) : nullptr;
        if (!lvl)
            throw std::invalid_argument("parms_id is not valid for encryption parameters");
        const uint64_t mag = value < 0 ? (uint64_t)0 - (uint64_t)value : (uint64_t)value;
        const int coeff_bit_count = (mag ? 64 - __builtin_clzll(mag) : 0) + 2;
        if (coeff_bit_count >= lvl->total_coeff_modulus_bit_count)
            throw std::invalid_argument("encoded value is too large");
        std::vector<uint64_t> residues(lvl->K);
        for (unsigned j = 0; j < lvl->K; j++)
        {
            const uint64_t q = context_.coeff_modulus()[j];
            uint64_t tmp = static_cast<uint64_t>(value);
            if (value < 0)
                tmp += q; // wraps modulo 2^64, as the reference's line does
            residues[j] = tmp % q;
        }
        fill_constant(*lvl, residues, 1.0, dest);
    }

    void CKKSEncoder::decode(const Plaintext &plain, double *values, bool want_complex) const
    {
        // ckks.h:683-789
        if (&plain.context() != &context_ || (plain.is_ntt_form() && plain.coeff_count() != plain.level()->K * context_.n()))
            throw std::invalid_argument("plain is not valid for encryption parameters");
        if (!plain.is_ntt_form())
            throw std::invalid_argument("plain is not in NTT form");
        if (!values)
            throw std::invalid_argument("destination cannot be null");
        const Level &lvl = *plain.level();
        if (lvl.chain_index > context_.first_level().chain_index)
            throw std::invalid_argument("plain is not valid for encryption parameters");
        if (!std::isnormal(plain.scale()) || plain.scale() <= 0 ||
            (static_cast<int>(std::log2(plain.scale())) >= lvl.total_coeff_modulus_bit_count))
            throw std::invalid_argument("scale out of bounds");
        const size_t n = context_.n(), K = lvl.K;
        const unsigned n_log = (unsigned)context_.log_n();
        const double inv_scale = double(1.0) / plain.scale();
        const LevelConst &lc = level_const(lvl);
        Scratch copy(K * n), res(2 * n), out(2 * slots_);
        ck(hipStreamSynchronize(nullptr), "decode sync");
        ck(hipMemcpyAsync(copy.p, plain.data(), K * n * 8, hipMemcpyDeviceToDevice, nullptr), "copy plain");
        NttBatch b{};
        b.data = copy.p;
        b.outer_stride = K * n;
        b.ncomp = (unsigned)K;
        b.nouter = 1;
        b.prime_first = 0;
        ck(ntt_inverse(context_.ntt_tables(), b, 0, nullptr), "intt plaintext");
        double2 *rv = reinterpret_cast<double2 *>(res.p);
        ck(k_ckks_compose_scale(context_.dev_mods(), copy.p, lc.dev, reinterpret_cast<const ShoupOp *>(lc.dev + K * K + 2 * K), lc.dev + K * K,
                                lc.dev + K * K + K, inv_scale, rv, n_log, (unsigned)K, 1, nullptr),
           "crt compose");
        // fft_handler_.transform_to_rev(res, logn, root_powers_)
        for (int g = (int)n_log - 1; g >= 0; g--)
            ck(k_fft_ct_stage(rv, roots_, n_log, (unsigned)g, 1, nullptr), "fft stage");
        ck(k_ckks_gather(map_, rv, reinterpret_cast<double2 *>(out.p), n_log, nullptr), "gather slots");
        std::vector<double> host(2 * slots_);
        ck(hipMemcpy(host.data(), out.p, 2 * slots_ * 8, hipMemcpyDeviceToHost), "download values");
        if (want_complex)
            std::memcpy(values, host.data(), 2 * slots_ * 8);
        else
            for (size_t i = 0; i < slots_; i++)
                values[i] = host[2 * i]; // from_complex<double>: the real part
    }
} // namespace sealhip
