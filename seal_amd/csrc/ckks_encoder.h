// seal::CKKSEncoder on the device (native/src/seal/ckks.h, ckks.cpp): vectors of N/2 complex (or real) numbers
// <-> NTT-form plaintexts.  Host: the index map, the complex root tables (util::ComplexRoots, croots.cpp: std::polar over an
// eighth of the circle and its symmetries - computed here with the same expressions and the same libm) and the multi-precision
// CRT constants.  Device: the FFT, the rounding / decomposition, the NTT engine, the CRT composition (ckks_kernels.h).
#pragma once
#include "ckks_kernels.h"
#include "evaluator.h"

namespace sealhip
{
    // Multi-precision CRT constants of a level, one device block of K*K + 4K words (RNSBase::initialize, rns.cpp:212-257;
    // total_coeff_modulus / upper_half_threshold, context.cpp:300-330): punct [K][K] (Q / q_j as K words) | Q [K] | (Q + 1) / 2 [K] |
    // (Q / q_j)^-1 mod q_j as K Shoup pairs.  The caller owns the block (hipFree).
    uint64_t *build_crt_constants(const Context &context, const Level &lvl);

    class CKKSEncoder
    {
    public:
        explicit CKKSEncoder(const Context &context); // ckks.cpp:16-70
        ~CKKSEncoder();
        CKKSEncoder(const CKKSEncoder &) = delete;
        CKKSEncoder &operator=(const CKKSEncoder &) = delete;
        size_t slot_count() const { return slots_; }
        // CKKSEncoder::encode(values, parms_id, scale, destination) (ckks.h:458-680): values = count <= N/2 complex numbers as
        // (re, im) pairs, or real numbers when is_complex is false
        void encode(const double *values, size_t count, bool is_complex, const uint64_t *parms_id, double scale, Plaintext &destination) const;
        // CKKSEncoder::encode(double value, ...) / encode(int64_t value, ...) (ckks.cpp:72-250): the value in every slot = a
        // constant polynomial, whose NTT form is that constant in every position
        void encode_value(double value, const uint64_t *parms_id, double scale, Plaintext &destination) const;
        void encode_integer(int64_t value, const uint64_t *parms_id, Plaintext &destination) const;
        // CKKSEncoder::decode (ckks.h:683-789): N/2 complex numbers as (re, im) pairs, or their real parts
        void decode(const Plaintext &plain, double *values, bool want_complex) const;

    private:
        void fill_constant(const Level &lvl, const std::vector<uint64_t> &residues, double scale, Plaintext &destination) const;
        struct LevelConst
        {
            uint64_t *dev = nullptr; // punct [K][K] | q_words [K] | half_words [K] | inv_punct [K] Shoup pairs
        };
        const LevelConst &level_const(const Level &lvl) const;
        const Context &context_;
        size_t slots_;
        uint32_t *map_ = nullptr;
        double2 *roots_ = nullptr, *inv_roots_ = nullptr;
        mutable std::mutex mu_;
        mutable std::map<size_t, LevelConst> consts_;
    };
} // namespace sealhip
