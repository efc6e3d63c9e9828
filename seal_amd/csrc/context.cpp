#include "context.h"
#include "blake2.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace sealhip
{
    namespace
    {
        using namespace host;

        void check_hip(hipError_t e, const char *what)
        {
            if (e != hipSuccess)
                throw std::runtime_error(std::string("HIP failure in ") + what + ": " + hipGetErrorString(e));
        }

        // (prod_{k != skip} base[k]) mod m
        uint64_t punctured_prod_mod(const std::vector<uint64_t> &base, size_t skip, uint64_t m)
        {
            uint64_t r = 1 % m;
            for (size_t k = 0; k < base.size(); k++)
                if (k != skip)
                    r = mulmod(r, base[k] % m, m);
            return r;
        }
        uint64_t prod_mod(const std::vector<uint64_t> &base, uint64_t m)
        {
            return punctured_prod_mod(base, base.size(), m);
        }

        // tiny bump allocator for building one level's constant block on the host
        struct Block
        {
            std::vector<uint64_t> words;
            size_t put(const void *p, size_t bytes)
            {
                size_t off = (words.size() + 1) & ~size_t(1); // 16-byte align
                words.resize(off + (bytes + 7) / 8);
                std::memcpy(words.data() + off, p, bytes);
                return off;
            }
            template <typename T>
            size_t put(const std::vector<T> &v)
            {
                static T dummy{};
                return put(v.empty() ? (const void *)&dummy : (const void *)v.data(), v.size() * sizeof(T));
            }
        };
    } // namespace

    // the checks SEALContext::validate makes BEFORE its security verdict (context.cpp:142-218): scheme, modulus count and sizes,
    // primality, degree.  Host arithmetic only - SEALContext_Create runs them ahead of its early security check so that a parameter
    // set that is both malformed and too large reports what the reference reports
    void Context::check_basic_parameters(Scheme scheme, size_t n, const std::vector<uint64_t> &primes)
    {
        if (scheme != Scheme::bfv && scheme != Scheme::ckks && scheme != Scheme::bgv)
            throw std::invalid_argument("invalid_scheme");
        if (primes.empty() || primes.size() > kMaxComps)
            throw std::invalid_argument("invalid_coeff_modulus_size");
        for (uint64_t q : primes)
        {
            if ((q >> 60) || !(q >> 1))
                throw std::invalid_argument("invalid_coeff_modulus_bit_count");
            if (!is_prime(q))
                throw std::invalid_argument("invalid_coeff_modulus_non_prime");
        }
        if (n < 2 || n > 131072 || (n & (n - 1)))
            throw std::invalid_argument("invalid_poly_modulus_degree");
    }

    Context::Context(
        Scheme scheme, size_t n, const std::vector<uint64_t> &coeff_modulus, uint64_t plain_modulus,
        bool expand_mod_chain)
        : scheme_(scheme), n_(n), plain_modulus_(plain_modulus), primes_(coeff_modulus)
    {
        // ---- validation, in the order of SEALContext::validate (context.cpp:142-460)
        check_basic_parameters(scheme, n, primes_);
        log_n_ = bit_count(n) - 1;
        for (size_t i = 0; i < primes_.size(); i++)
            for (size_t j = i + 1; j < primes_.size(); j++)
                if (primes_[i] == primes_[j])
                    throw std::invalid_argument("failed_creating_rns_base");
        for (uint64_t q : primes_)
            if ((q - 1) % (2 * n) != 0)
                throw std::invalid_argument("invalid_coeff_modulus_no_ntt");
        if (scheme == Scheme::bfv || scheme == Scheme::bgv)
        {
            if ((plain_modulus >> 60) || !(plain_modulus >> 1))
                throw std::invalid_argument("invalid_plain_modulus_bit_count");
            for (uint64_t q : primes_)
                if (std::__gcd(q, plain_modulus) != 1)
                    throw std::invalid_argument("invalid_plain_modulus_coprimality");
            if (primes_.size() == 1 && plain_modulus >= primes_[0])
                throw std::invalid_argument("invalid_plain_modulus_too_large");
            using_batching_ = is_prime(plain_modulus) && (plain_modulus - 1) % (2 * n) == 0;
        }
        else
        {
            if (plain_modulus != 0)
                throw std::invalid_argument("invalid_plain_modulus_nonzero");
            using_batching_ = true;
        }

        // ---- chain (context.cpp:495-575): key level, then drop the last prime repeatedly
        size_t L = primes_.size();
        std::vector<unsigned> ks;
        ks.push_back((unsigned)L);
        if (L > 1)
        {
            ks.push_back((unsigned)L - 1);
            using_keyswitching_ = true;
            if (expand_mod_chain)
                for (unsigned k = (unsigned)L - 2; k >= 1; k--)
                {
                    // BFV/BGV: a level is only valid while plain_modulus < total modulus
                    if (scheme != Scheme::ckks && k == 1 && plain_modulus >= primes_[0])
                        break;
                    ks.push_back(k);
                }
        }
        levels_.resize(ks.size());
        for (size_t i = 0; i < ks.size(); i++)
        {
            levels_[i].K = ks[i];
            levels_[i].chain_index = ks.size() - 1 - i;
        }

        build_pool_and_tables();
        for (auto &l : levels_)
            build_level(l);
    }

    Context::~Context()
    {
        for (auto &l : levels_)
            if (l.dev_block)
                (void)hipFree(l.dev_block);
        if (d_mods_)
            (void)hipFree(d_mods_);
        if (d_fwd_)
            (void)hipFree(d_fwd_);
        if (d_inv_)
            (void)hipFree(d_inv_);
        if (d_ninv_)
            (void)hipFree(d_ninv_);
        if (d_fpd_)
            (void)hipFree(d_fpd_);
        if (d_fwd_d_)
            (void)hipFree(d_fwd_d_);
        if (d_inv_d_)
            (void)hipFree(d_inv_d_);
        if (d_ninv_d_)
            (void)hipFree(d_ninv_d_);
    }

    const Level *Context::level_by_chain_index(size_t chain_index) const
    {
        for (auto &l : levels_)
            if (l.chain_index == chain_index)
                return &l;
        return nullptr;
    }
    const Level *Context::level_by_parms_id(const uint64_t *id) const
    {
        for (auto &l : levels_)
            if (!std::memcmp(l.parms_id, id, sizeof(parms_id_type)))
                return &l;
        return nullptr;
    }
    const Level *Context::next_level(const Level &l) const
    {
        return l.chain_index == 0 ? nullptr : level_by_chain_index(l.chain_index - 1);
    }
    void Context::set_parms_id(size_t chain_index, const uint64_t *id)
    {
        for (auto &l : levels_)
            if (l.chain_index == chain_index)
                std::memcpy(l.parms_id, id, sizeof(parms_id_type));
    }

    void Context::build_pool_and_tables()
    {
        // BEHZ auxiliary primes (rns.cpp:605-632): the largest level decides how many we need.
        // get_primes returns them in descending order: m_sk, gamma, then B.
        size_t max_bsk_mtilde = 0;
        if (scheme_ == Scheme::bfv)
        {
            for (auto &l : levels_)
            {
                std::vector<uint64_t> q(primes_.begin(), primes_.begin() + l.K);
                int total_bits = significant_bits(product(q));
                l.total_coeff_modulus_bit_count = total_bits;
                size_t nb = l.K;
                if (32 + bit_count(plain_modulus_) + total_bits >= 61 * (int)l.K + 61)
                    nb++;
                max_bsk_mtilde = std::max(max_bsk_mtilde, nb + 2);
            }
        }
        else
        {
            for (auto &l : levels_)
            {
                std::vector<uint64_t> q(primes_.begin(), primes_.begin() + l.K);
                l.total_coeff_modulus_bit_count = significant_bits(product(q));
            }
        }
        pool_ = primes_;
        if (max_bsk_mtilde)
        {
            auto aux = get_primes(2 * (uint64_t)n_, 61, max_bsk_mtilde);
            pool_.insert(pool_.end(), aux.begin(), aux.end());
        }
        if (scheme_ != Scheme::ckks && using_batching_)
        {
            // the plain modulus gets NTT tables of its own (BatchEncoder): last entry of the pool
            plain_prime_ = (int)pool_.size();
            pool_.push_back(plain_modulus_);
        }

        // parms_id of a level = BLAKE2b-256 over (scheme, N, q_0 .. q_{K-1}, t) as 64-bit words
        // (EncryptionParameters::compute_parms_id, encryptionparams.cpp:117-147; HashFunction::hash, util/hash.h:30-37):
        // the same 256 bits the reference computes, so serialized ciphertexts and keys name their level identically
        for (auto &l : levels_)
        {
            std::vector<uint64_t> words;
            words.push_back((uint64_t)scheme_);
            words.push_back((uint64_t)n_);
            words.insert(words.end(), primes_.begin(), primes_.begin() + l.K);
            words.push_back(plain_modulus_);
            blake2::blake2b(l.parms_id, sizeof(parms_id_type), words.data(), words.size() * sizeof(uint64_t));
        }

        size_t np = pool_.size();
        h_mods_.resize(np);
        roots_.assign(np, 0);
        std::vector<ShoupOp> fwd(np * n_), inv(np * n_), ninv(np * 2);
        for (size_t p = 0; p < np; p++)
        {
            uint64_t q = pool_[p];
            h_mods_[p] = make_mod(q);
            if (p == primes_.size() + 1 && (int)p != plain_prime_)
                continue; // gamma: never transformed
            uint64_t root;
            if (!minimal_primitive_root(2 * (uint64_t)n_, q, root))
                throw std::invalid_argument("invalid_coeff_modulus_no_ntt");
            roots_[p] = root;
            ShoupOp *f = fwd.data() + p * n_;
            ShoupOp *iv = inv.data() + p * n_;
            // fwd[bitrev(i)] = psi^i  (ntt.cpp:273-278);  inv[j] = fwd[j]^-1 (our layout)
            uint64_t power = 1;
            uint64_t inv_root = invmod(root, q);
            uint64_t ipower = 1;
            for (size_t i = 0; i < n_; i++)
            {
                uint32_t r = reverse_bits((uint32_t)i, log_n_);
                f[r] = make_shoup(power, q);
                iv[r] = make_shoup(ipower, q);
                power = mulmod(power, root, q);
                ipower = mulmod(ipower, inv_root, q);
            }
            uint64_t ni = invmod((uint64_t)n_ % q, q);
            ninv[2 * p] = make_shoup(ni, q);
            ninv[2 * p + 1] = make_shoup(mulmod(ni, n_ > 1 ? iv[1].w : 1, q), q);
        }
        // double-precision twins of the tables for primes below 2^kFpMaxBits (field.h); only the
        // two-pass engine (ntt2_kernels.hip) reads them, so they are built for its sizes only
        h_fpd_.assign(np, FpDesc{ 0.0, 0.0, 0.0, 0 });
        const bool want_fp = log_n_ >= 13 && log_n_ <= 16 && !std::getenv("SEALHIP_NO_FP");
        std::vector<double> fwd_d(want_fp ? np * n_ : 1), inv_d(want_fp ? np * n_ : 1), ninv_d(np * 2);
        for (size_t p = 0; want_fp && p < np; p++)
        {
            uint64_t q = pool_[p];
            if (!roots_[p] || (q >> kFpMaxBits))
                continue;
            h_fpd_[p].q = (double)q;
            h_fpd_[p].qinv = 1.0 / (double)q;
            h_fpd_[p].two32 = (double)((uint64_t(1) << 32) % q);
            h_fpd_[p].qi = q;
            // BALANCED representatives in (-q/2, q/2]: a multiplier of magnitude <= q/2 halves the quotient-estimate term of the
            // butterfly bound (field.h: |r| <= q (1/2 + 3/16 B) instead of 3/8 B), which is what lets the key-switch kernels
            // run seven stages between two fix() calls
            const auto balanced = [q](uint64_t w) { return w > q / 2 ? -(double)(q - w) : (double)w; };
            for (size_t i = 0; i < n_; i++)
            {
                fwd_d[p * n_ + i] = balanced(fwd[p * n_ + i].w);
                inv_d[p * n_ + i] = balanced(inv[p * n_ + i].w);
            }
            ninv_d[2 * p] = balanced(ninv[2 * p].w);
            ninv_d[2 * p + 1] = balanced(ninv[2 * p + 1].w);
        }
        check_hip(hipMalloc(&d_fpd_, np * sizeof(FpDesc)), "hipMalloc fpd");
        check_hip(hipMalloc(&d_fwd_d_, fwd_d.size() * 8), "hipMalloc fwd_d");
        check_hip(hipMalloc(&d_inv_d_, inv_d.size() * 8), "hipMalloc inv_d");
        check_hip(hipMalloc(&d_ninv_d_, ninv_d.size() * 8), "hipMalloc ninv_d");
        check_hip(hipMemcpy(d_fpd_, h_fpd_.data(), np * sizeof(FpDesc), hipMemcpyHostToDevice), "upload fpd");
        check_hip(hipMemcpy(d_fwd_d_, fwd_d.data(), fwd_d.size() * 8, hipMemcpyHostToDevice), "upload fwd_d");
        check_hip(hipMemcpy(d_inv_d_, inv_d.data(), inv_d.size() * 8, hipMemcpyHostToDevice), "upload inv_d");
        check_hip(hipMemcpy(d_ninv_d_, ninv_d.data(), ninv_d.size() * 8, hipMemcpyHostToDevice), "upload ninv_d");
        tables_.fpd = d_fpd_;
        h_fp_flag_.resize(np);
        for (size_t p = 0; p < np; p++)
            h_fp_flag_[p] = h_fpd_[p].qi != 0;
        tables_.fp_host = h_fp_flag_.data();
        tables_.fwd_d = d_fwd_d_;
        tables_.inv_d = d_inv_d_;
        tables_.ninv_d = d_ninv_d_;
        check_hip(hipMalloc(&d_mods_, np * sizeof(ModDesc)), "hipMalloc mods");
        check_hip(hipMalloc(&d_fwd_, fwd.size() * sizeof(ShoupOp)), "hipMalloc fwd tables");
        check_hip(hipMalloc(&d_inv_, inv.size() * sizeof(ShoupOp)), "hipMalloc inv tables");
        check_hip(hipMalloc(&d_ninv_, ninv.size() * sizeof(ShoupOp)), "hipMalloc ninv");
        check_hip(hipMemcpy(d_mods_, h_mods_.data(), np * sizeof(ModDesc), hipMemcpyHostToDevice), "upload mods");
        check_hip(hipMemcpy(d_fwd_, fwd.data(), fwd.size() * sizeof(ShoupOp), hipMemcpyHostToDevice), "upload fwd");
        check_hip(hipMemcpy(d_inv_, inv.data(), inv.size() * sizeof(ShoupOp), hipMemcpyHostToDevice), "upload inv");
        check_hip(hipMemcpy(d_ninv_, ninv.data(), ninv.size() * sizeof(ShoupOp), hipMemcpyHostToDevice), "upload ninv");
        tables_.mods = d_mods_;
        tables_.fwd = d_fwd_;
        tables_.inv = d_inv_;
        tables_.ninv = d_ninv_;
        tables_.log_n = log_n_;
    }

    void Context::build_level(Level &lvl)
    {
        const unsigned K = lvl.K;
        std::vector<uint64_t> q(primes_.begin(), primes_.begin() + K);
        Block blk;
        struct Off
        {
            size_t inv_q_last = 0, round_fix = 0, half_mod_q = 0, q_last_mod_q = 0, delta_mod_q = 0, upper_half_inc = 0, bsk_prime = 0, inv_punct_q = 0, m_tilde_mod_q = 0, q_to_bsk = 0, q_to_mtilde = 0,
                   prod_q_mod_bsk = 0, inv_mtilde_mod_bsk = 0, inv_prod_q_mod_bsk = 0, inv_punct_b = 0, b_to_q = 0,
                   b_to_msk = 0, prod_b_mod_q = 0, t_mod_q = 0, t_mod_bsk = 0, mt_inv_punct_q = 0, q_to_bsk_lift = 0, prod_q_lift = 0,
                   t_inv_punct_q = 0, q_to_bsk_floor = 0, t_floor_bsk = 0, two64_bsk = 0, two64_q = 0, neg_base_q = 0, b_to_q3 = 0, b_to_msk3 = 0, dec_inv_punct_q = 0, dec_q_to_t = 0, dec_prod_t_gamma = 0,
                   dec_q_to_gamma = 0;
        } off;

        // q_last^-1 mod q_i  (rns.cpp:769-776)
        {
            std::vector<ShoupOp> v;
            for (unsigned i = 0; i + 1 < K; i++)
                v.push_back(make_shoup(invmod(q[K - 1] % q[i], q[i]), q[i]));
            off.inv_q_last = blk.put(v);
            std::vector<uint64_t> fix, hm;
            uint64_t half = q[K - 1] >> 1;
            for (unsigned i = 0; i + 1 < K; i++)
            {
                hm.push_back(half % q[i]);
                fix.push_back(q[i] - half % q[i]);
            }
            off.round_fix = blk.put(fix);
            off.half_mod_q = blk.put(hm);
            std::vector<uint64_t> qlm;
            for (unsigned i = 0; i + 1 < K; i++)
                qlm.push_back(q[K - 1] % q[i]);
            off.q_last_mod_q = blk.put(qlm);
            if (scheme_ == Scheme::bgv)
                lvl.dev.inv_q_last_mod_t = invmod(q[K - 1] % plain_modulus_, plain_modulus_);
            if (scheme_ != Scheme::ckks)
            {
                // floor(Q/t) = (Q - (Q mod t)) / t exactly, and Q = 0 mod q_i, hence floor(Q/t) = -(Q mod t) t^-1 (mod q_i)
                const uint64_t t = plain_modulus_;
                const uint64_t r = prod_mod(q, t);
                std::vector<uint64_t> dl, inc;
                for (unsigned i = 0; i < K; i++)
                {
                    const uint64_t rm = r % q[i];
                    dl.push_back(mulmod(rm ? q[i] - rm : 0, invmod(t % q[i], q[i]), q[i]));
                    inc.push_back((q[i] - t % q[i]) % q[i]);
                }
                off.delta_mod_q = blk.put(dl);
                off.upper_half_inc = blk.put(inc);
                lvl.dev.q_mod_t = r;
                lvl.dev.plain_upper_half_threshold = (t + 1) >> 1;
            }
            lvl.dev.q_last = q[K - 1];
            lvl.dev.half_q_last = half;
        }
        lvl.dev.K = K;

        if (scheme_ != Scheme::ckks)
        {
            // decryption constants (rns.cpp:617-650, 680-690, 738-765): base conversion q -> {t} (BGV, exact) and
            // q -> {t, gamma} (BFV)
            const uint64_t t = plain_modulus_;
            std::vector<ShoupOp> ipq;
            std::vector<uint64_t> q2t;
            for (unsigned i = 0; i < K; i++)
            {
                ipq.push_back(make_shoup(invmod(punctured_prod_mod(q, i, q[i]), q[i]), q[i]));
                q2t.push_back(punctured_prod_mod(q, i, t));
            }
            off.dec_inv_punct_q = blk.put(ipq);
            off.dec_q_to_t = blk.put(q2t);
            if (scheme_ == Scheme::bfv)
            {
                const unsigned gp = aux_first() + 1;
                const uint64_t gamma = pool_[gp];
                std::vector<ShoupOp> ptg;
                std::vector<uint64_t> q2g;
                for (unsigned i = 0; i < K; i++)
                {
                    ptg.push_back(make_shoup(mulmod(t % q[i], gamma % q[i], q[i]), q[i]));
                    q2g.push_back(punctured_prod_mod(q, i, gamma));
                }
                off.dec_prod_t_gamma = blk.put(ptg);
                off.dec_q_to_gamma = blk.put(q2g);
                lvl.dev.gamma_prime = gp;
                const uint64_t qt = prod_mod(q, t), qg = prod_mod(q, gamma);
                lvl.dev.dec_neg_inv_q_mod_t = (t - invmod(qt, t)) % t;
                lvl.dev.dec_neg_inv_q_mod_gamma = (gamma - invmod(qg, gamma)) % gamma;
                lvl.dev.dec_inv_gamma_mod_t = invmod(gamma % t, t);
            }
        }

        const bool behz = scheme_ == Scheme::bfv;
        if (behz)
        {
            // bases (rns.cpp:605-648)
            size_t nb = K;
            if (32 + bit_count(plain_modulus_) + lvl.total_coeff_modulus_bit_count >= 61 * (int)K + 61)
                nb++;
            const unsigned aux = aux_first();
            const uint64_t m_sk = pool_[aux];
            const uint64_t m_tilde = uint64_t(1) << 32;
            std::vector<uint64_t> B(pool_.begin() + aux + 2, pool_.begin() + aux + 2 + nb);
            std::vector<uint64_t> Bsk = B;
            Bsk.push_back(m_sk);
            std::vector<uint32_t> bsk_prime;
            for (size_t i = 0; i < nb; i++)
                bsk_prime.push_back(aux + 2 + (uint32_t)i);
            bsk_prime.push_back(aux);
            lvl.bsk = Bsk;
            lvl.dev.nB = (unsigned)nb;
            lvl.dev.nBsk = (unsigned)nb + 1;
            lvl.dev.m_tilde = m_tilde;
            lvl.dev.msk_prime = aux;
            off.bsk_prime = blk.put(bsk_prime);

            std::vector<ShoupOp> inv_punct_q, m_tilde_mod_q, t_mod_q, inv_mtilde_mod_bsk, inv_prod_q_mod_bsk,
                inv_punct_b, t_mod_bsk;
            std::vector<uint64_t> q_to_bsk, q_to_mtilde, prod_q_mod_bsk, b_to_q, b_to_msk, prod_b_mod_q;
            for (unsigned i = 0; i < K; i++)
            {
                inv_punct_q.push_back(make_shoup(invmod(punctured_prod_mod(q, i, q[i]), q[i]), q[i]));
                m_tilde_mod_q.push_back(make_shoup(m_tilde % q[i], q[i]));
                t_mod_q.push_back(make_shoup(plain_modulus_ % q[i], q[i]));
                q_to_mtilde.push_back(punctured_prod_mod(q, i, m_tilde));
                prod_b_mod_q.push_back(prod_mod(B, q[i]));
            }
            for (size_t j = 0; j < Bsk.size(); j++)
            {
                for (unsigned i = 0; i < K; i++)
                    q_to_bsk.push_back(punctured_prod_mod(q, i, Bsk[j]));
                uint64_t pq = prod_mod(q, Bsk[j]);
                prod_q_mod_bsk.push_back(pq);
                inv_prod_q_mod_bsk.push_back(make_shoup(invmod(pq, Bsk[j]), Bsk[j]));
                inv_mtilde_mod_bsk.push_back(make_shoup(invmod(m_tilde % Bsk[j], Bsk[j]), Bsk[j]));
                t_mod_bsk.push_back(make_shoup(plain_modulus_ % Bsk[j], Bsk[j]));
            }
            for (size_t i = 0; i < nb; i++)
            {
                inv_punct_b.push_back(make_shoup(invmod(punctured_prod_mod(B, i, B[i]), B[i]), B[i]));
                b_to_msk.push_back(punctured_prod_mod(B, i, m_sk));
            }
            for (unsigned j = 0; j < K; j++)
                for (size_t i = 0; i < nb; i++)
                    b_to_q.push_back(punctured_prod_mod(B, i, q[j]));
            lvl.dev.inv_prod_b_mod_msk = make_shoup(invmod(prod_mod(B, m_sk), m_sk), m_sk);
            // -prod(q)^-1 mod m~  (rns.cpp:727-733)
            uint64_t pqm = prod_mod(q, m_tilde);
            lvl.dev.neg_inv_prod_q_mod_mtilde = (m_tilde - invmod(pqm, m_tilde)) % m_tilde;

            off.inv_punct_q = blk.put(inv_punct_q);
            off.m_tilde_mod_q = blk.put(m_tilde_mod_q);
            off.q_to_bsk = blk.put(q_to_bsk);
            off.q_to_mtilde = blk.put(q_to_mtilde);
            off.prod_q_mod_bsk = blk.put(prod_q_mod_bsk);
            off.inv_mtilde_mod_bsk = blk.put(inv_mtilde_mod_bsk);
            off.inv_prod_q_mod_bsk = blk.put(inv_prod_q_mod_bsk);
            off.inv_punct_b = blk.put(inv_punct_b);
            off.b_to_q = blk.put(b_to_q);
            off.b_to_msk = blk.put(b_to_msk);
            off.prod_b_mod_q = blk.put(prod_b_mod_q);
            off.t_mod_q = blk.put(t_mod_q);
            off.t_mod_bsk = blk.put(t_mod_bsk);
            // products of the constants above (LevelDev: "multiplied together")
            std::vector<ShoupOp> mt_ipq, t_ipq, t_floor;
            std::vector<uint64_t> lift_m, lift_pq, floor_m;
            for (unsigned i = 0; i < K; i++)
            {
                mt_ipq.push_back(make_shoup(mulmod(m_tilde % q[i], inv_punct_q[i].w, q[i]), q[i]));
                t_ipq.push_back(make_shoup(mulmod(plain_modulus_ % q[i], inv_punct_q[i].w, q[i]), q[i]));
            }
            for (size_t j = 0; j < Bsk.size(); j++)
            {
                const uint64_t pj = Bsk[j];
                const uint64_t im = inv_mtilde_mod_bsk[j].w;
                uint64_t fl = inv_prod_q_mod_bsk[j].w;
                if (j < nb)
                    fl = mulmod(fl, inv_punct_b[j].w, pj);
                for (unsigned i = 0; i < K; i++)
                {
                    lift_m.push_back(mulmod(q_to_bsk[j * K + i], im, pj));
                    floor_m.push_back(mulmod(q_to_bsk[j * K + i], fl, pj));
                }
                lift_pq.push_back(mulmod(prod_q_mod_bsk[j], im, pj));
                t_floor.push_back(make_shoup(mulmod(plain_modulus_ % pj, fl, pj), pj));
            }
            // the matrices the dot products of behz_kernels.hip read, cut into three 21-bit limbs: {limb0 | limb1 << 32, limb2}
            const auto split21 = [](const std::vector<uint64_t> &v) {
                std::vector<uint64_t> out;
                for (uint64_t r : v)
                {
                    out.push_back((r & 0x1FFFFFull) | (((r >> 21) & 0x1FFFFFull) << 32));
                    out.push_back(r >> 42);
                }
                return out;
            };
            lift_m = split21(lift_m);
            floor_m = split21(floor_m);
            off.b_to_q3 = blk.put(split21(b_to_q));
            off.b_to_msk3 = blk.put(split21(b_to_msk));
            off.mt_inv_punct_q = blk.put(mt_ipq);
            off.q_to_bsk_lift = blk.put(lift_m);
            off.prod_q_lift = blk.put(lift_pq);
            off.t_inv_punct_q = blk.put(t_ipq);
            off.q_to_bsk_floor = blk.put(floor_m);
            off.t_floor_bsk = blk.put(t_floor);
            std::vector<ShoupOp> two64_b, two64_qv;
            const auto two64 = [](uint64_t p) { return make_shoup((uint64_t)((((unsigned __int128)1) << 64) % p), p); };
            for (size_t j = 0; j < Bsk.size(); j++)
                two64_b.push_back(two64(Bsk[j]));
            for (unsigned i = 0; i < K; i++)
                two64_qv.push_back(two64(q[i]));
            off.two64_bsk = blk.put(two64_b);
            off.two64_q = blk.put(two64_qv);
            std::vector<uint64_t> neg_base;
            for (unsigned i = 0; i < K; i++)
                neg_base.push_back(q[i] * ((((uint64_t)1 << 60) + q[i] - 1) / q[i])); // below 2^60 + q_i <= 2^61
            off.neg_base_q = blk.put(neg_base);
        }

        uint64_t *d = nullptr;
        check_hip(hipMalloc(&d, blk.words.size() * 8 + 16), "hipMalloc level block");
        check_hip(hipMemcpy(d, blk.words.data(), blk.words.size() * 8, hipMemcpyHostToDevice), "upload level block");
        lvl.dev_block = d;
        lvl.dev.inv_q_last_mod_q = reinterpret_cast<const ShoupOp *>(d + off.inv_q_last);
        lvl.dev.round_fix = d + off.round_fix;
        lvl.dev.half_mod_q = d + off.half_mod_q;
        lvl.dev.q_last_mod_q = d + off.q_last_mod_q;
        if (scheme_ != Scheme::ckks)
        {
            lvl.dev.delta_mod_q = d + off.delta_mod_q;
            lvl.dev.upper_half_inc = d + off.upper_half_inc;
            lvl.dev.dec_inv_punct_q = reinterpret_cast<const ShoupOp *>(d + off.dec_inv_punct_q);
            lvl.dev.dec_q_to_t = d + off.dec_q_to_t;
            if (scheme_ == Scheme::bfv)
            {
                lvl.dev.dec_prod_t_gamma_mod_q = reinterpret_cast<const ShoupOp *>(d + off.dec_prod_t_gamma);
                lvl.dev.dec_q_to_gamma = d + off.dec_q_to_gamma;
            }
        }
        if (behz)
        {
            lvl.dev.bsk_prime = reinterpret_cast<const uint32_t *>(d + off.bsk_prime);
            lvl.dev.inv_punct_q = reinterpret_cast<const ShoupOp *>(d + off.inv_punct_q);
            lvl.dev.m_tilde_mod_q = reinterpret_cast<const ShoupOp *>(d + off.m_tilde_mod_q);
            lvl.dev.q_to_bsk = d + off.q_to_bsk;
            lvl.dev.q_to_mtilde = d + off.q_to_mtilde;
            lvl.dev.prod_q_mod_bsk = d + off.prod_q_mod_bsk;
            lvl.dev.inv_mtilde_mod_bsk = reinterpret_cast<const ShoupOp *>(d + off.inv_mtilde_mod_bsk);
            lvl.dev.inv_prod_q_mod_bsk = reinterpret_cast<const ShoupOp *>(d + off.inv_prod_q_mod_bsk);
            lvl.dev.inv_punct_b = reinterpret_cast<const ShoupOp *>(d + off.inv_punct_b);
            lvl.dev.b_to_q = d + off.b_to_q;
            lvl.dev.b_to_msk = d + off.b_to_msk;
            lvl.dev.prod_b_mod_q = d + off.prod_b_mod_q;
            lvl.dev.t_mod_q = reinterpret_cast<const ShoupOp *>(d + off.t_mod_q);
            lvl.dev.t_mod_bsk = reinterpret_cast<const ShoupOp *>(d + off.t_mod_bsk);
            lvl.dev.mt_inv_punct_q = reinterpret_cast<const ShoupOp *>(d + off.mt_inv_punct_q);
            lvl.dev.q_to_bsk_lift = d + off.q_to_bsk_lift;
            lvl.dev.prod_q_lift = d + off.prod_q_lift;
            lvl.dev.t_inv_punct_q = reinterpret_cast<const ShoupOp *>(d + off.t_inv_punct_q);
            lvl.dev.q_to_bsk_floor = d + off.q_to_bsk_floor;
            lvl.dev.t_floor_bsk = reinterpret_cast<const ShoupOp *>(d + off.t_floor_bsk);
            lvl.dev.two64_bsk = reinterpret_cast<const ShoupOp *>(d + off.two64_bsk);
            lvl.dev.two64_q = reinterpret_cast<const ShoupOp *>(d + off.two64_q);
            lvl.dev.neg_base_q = d + off.neg_base_q;
            lvl.dev.b_to_q3 = d + off.b_to_q3;
            lvl.dev.b_to_msk3 = d + off.b_to_msk3;
        }
    }
} // namespace sealhip
