// Negacyclic NTT / INTT for gfx950 — device-side building blocks and launch interface.
//
// Replaces the reference's DWTHandler::transform_to_rev / transform_from_rev
// (native/src/seal/util/dwthandler.h:94-356) and the ntt_negacyclic_harvey* wrappers
// (native/src/seal/util/ntt.cpp:394-475).  Same mathematical transform — root psi =
// NTTTables::get_root(), natural-order input, bit-reversed output (SURVEY §8(a') "NTT") — but
// decomposed for the GPU: a transform of 2^n points is one or two "passes"; a pass executes D
// consecutive radix-2 stages for a tile held in LDS, as ceil(D/R) register phases of R stages on
// 2^R coefficients per thread (radix-2^R blocks), exchanging through LDS between phases.  The
// first pass of a two-pass transform works on strided columns (coalesced along the contiguous
// low index), the last pass on contiguous rows; the merged psi-power twiddles of the reference's
// bit-reversed table are what lets the two passes compose without an extra twiddle multiply.
#pragma once
#include "field.h"

namespace sealhip
{
    // Per-prime table set resident in HBM.  fwd[i] = psi^bitrev_n(i), inv[i] = fwd[i]^-1
    // (our own layout for the inverse: same index as the forward stage that it undoes).
    struct NttTables
    {
        const ModDesc *mods;  // [nprimes]
        const ShoupOp *fwd;   // [nprimes][N]
        const ShoupOp *inv;   // [nprimes][N]
        const ShoupOp *ninv;  // [nprimes][2]: {N^-1, N^-1 * inv[1]}
        // double-precision back end (field.h): the same tables as doubles for primes below 2^50;
        // fpd[p].qi == 0 marks a prime that only the integer back end may use.
        const FpDesc *fpd;    // [nprimes]
        const double *fwd_d;  // [nprimes][N]
        const double *inv_d;  // [nprimes][N]
        const double *ninv_d; // [nprimes][2]
        // HOST copy of "fpd[p].qi != 0" per prime (null = unknown): lets the launcher pick kernels
        // specialised for one back end, which need far fewer registers than the mixed ones
        const unsigned char *fp_host;
        int log_n;
    };

    // Two rounding divisions folded into ONE forward transform (two-pass engine; NttBatch::tail2).  CKKS relinearize followed by
    // rescale divides by the special prime P (key-switch mod-down, evaluator.cpp:2806-2864) and then by q_last
    // (divide_and_round_q_last_ntt_inplace, rns.cpp:830-901).  With v_i = the mod-down correction of component i (from t_P, the
    // coefficient form of the special-prime sums) and u_i = the rescale correction (from t_last, the coefficient form of the
    // relinearised ciphertext's last component) the two steps give, for i < K - 1,
    //   out_i = (c_i + (S_i - NTT_i(v_i)) P^-1 - NTT_i(u_i)) q_last^-1 = (c_i + S_i P^-1 - NTT_i(v_i P^-1 + u_i)) q_last^-1   (mod q_i)
    // because the transform is linear over Z_q_i and every step is exact residue arithmetic: one transform per component
    // instead of two, and the relinearised ciphertext is never written.  The NttBatch carries the mod-down side in its src_*
    // fields (src = t_P, src_mode 2 constants of P) and epi_a = S (the key-switch sums), epi_mul = q_last^-1, epi_out0 / epi_out1 =
    // the two output planes; this struct adds the rest.  outer = 2 * item + plane (the layout of the key-switch sums).
    struct NttTail2
    {
        const uint64_t *src2_0, *src2_1; // t_last of plane 0 / 1: + (outer >> 1) * src2_stride
        size_t src2_stride;
        uint64_t src2_half, src2_q;      // q_last / 2, q_last
        const uint64_t *src2_fix;        // [ncomp] rescale rounding constants (device)
        const ShoupOp *pmul;             // [ncomp] P^-1 mod q_i (device)
        const uint64_t *c0, *c1;         // ciphertext planes: + (outer >> 1) * c_stride + comp * N
        size_t c_stride;
        int halves_added;                // both sources were produced with out_add = their half: the maps skip that step
        int a_has_c = 0;                 // epi_a holds c + S P^-1 (KsFusedArgs::fold_c0): c0 / c1 are not read
    };

    // One batched launch: transforms live at data + outer*outer_stride + comp*N, comp in
    // [0, ncomp), outer in [0, nouter); prime of a component = comp_prime[comp] (device array) or
    // prime_first + comp when comp_prime == nullptr.
    struct NttBatch
    {
        uint64_t *data;
        size_t outer_stride; // words
        unsigned ncomp;
        unsigned nouter;
        const uint32_t *comp_prime;
        unsigned prime_first;
        // Optional separate source for the first pass (forward only): src + outer*src_outer_stride
        // + (comp % src_ncomp)*N is read instead of data and mapped into the target prime:
        //   src_mode 1: x mod q                      (key-switch mod-raise, evaluator.cpp:2690-2701)
        //   src_mode 2: ((x + src_half) mod src_q) mod q + src_fix[comp]
        //               (the "+half, mod q_i, -half" rounding step fused into the load:
        //                rns.cpp:858-881 rescale, evaluator.cpp:2813-2832 key-switch mod-down)
        //   src_mode 3: x mod q + src_fix[comp]: mode 2 for a source whose producer has already added src_half (out_add below;
        //               two-pass engine only)
        const uint64_t *src;
        size_t src_outer_stride;
        unsigned src_ncomp;
        int src_mode;
        uint64_t src_half;
        uint64_t src_q;
        const uint64_t *src_fix; // [ncomp], device
        // Optional fused epilogue of the forward transform (two-pass engine only): instead of storing
        // the transform T to `data`, combine it with a resident operand and store the result,
        //   v = (A[outer][comp] - T) * mul[comp]  mod q                        (A canonical)
        //   epi 1: out[outer][comp] = v                      rescale tail, rns.cpp:890-899
        //   epi 2: ct_{outer&1}[outer>>1][comp] += v         key-switch tail, evaluator.cpp:2845-2863
        //   epi 3: ct_{outer&1}[outer>>1][comp] = A[outer][comp] - T * mul[comp]: the same tail when the key switch has left
        //          A = c + S P^-1 behind (KsFusedArgs::fold_c0)
        // A = epi_a + outer*epi_a_stride + comp*N; epi 1 writes epi_out0 + outer*epi_out_stride + comp*N,
        // epi 2 updates epi_out{0,1} + (outer>>1)*epi_out_stride + comp*N.
        int epi;
        const uint64_t *epi_a;
        size_t epi_a_stride;
        const ShoupOp *epi_mul; // [ncomp]
        uint64_t *epi_out0, *epi_out1;
        size_t epi_out_stride;
        const NttTail2 *tail2 = nullptr; // see NttTail2 (host pointer, read during the call only)
        // Inverse transforms of the two-pass engine, canonical output only: every output word v is stored as
        // (v + out_add) mod q.  Used on the single component a rounding division is about to drop, so that the "+ q/2" of the
        // rounding is added once per coefficient instead of once per target modulus (NttTail2::halves_added).
        uint64_t out_add = 0;
        // Inverse transforms of the two-pass engine: the input is the 2 x 2 tensor product of two size-2 operands, formed while it is
        // loaded (evaluator.cpp:497-541 followed by 543-547: the product is never stored in NTT form).  prod_x / prod_y = the
        // operands [2][prod_batch][ncomp][N] (canonical, item stride src_outer_stride), nouter = 3 * prod_batch: outer item
        // p * prod_batch + b is polynomial p of item b - x0 y0, x0 y1 + x1 y0, x1 y1.  `src` is not used.
        const uint64_t *prod_x = nullptr, *prod_y = nullptr;
        unsigned prod_batch = 0;
        // round 6 (CKKS multiply fused into relinearize): the launch covers outer items prod_outer0 .. prod_outer0 + nouter - 1 of the
        // 3 * prod_batch (prod_outer0 = 2 * prod_batch + b0: polynomial x1 y1 of the items from b0 on, and nothing else), and the
        // product is additionally STORED in NTT form at prod_out + (outer item in the launch) * outer stride of the product's slab
        // (prod_out_stride words; natural order, canonical) - the key switch reads it for its diagonal terms
        unsigned prod_outer0 = 0;
        uint64_t *prod_out = nullptr;
        size_t prod_out_stride = 0;
        // round 6 (rotations without the permutation kernels; inverse transform of the two-pass engine, plain `src` only): the input is
        // the NTT-domain automorphism of `src` with this Galois element - word j of a polynomial is read from position T(j) of the
        // same polynomial (galois.cpp:18-51; T keeps aligned blocks together, so the gather stays inside the lines a plain read touches)
        uint32_t src_galois_elt = 0;
        // Two-pass engine: what the HOST knows about the arithmetic class of the components when they are named through comp_prime
        // (the launcher reads the class of prime_first + comp itself, a device table it cannot): -1 unknown - the launch carries both
        // back ends and guards every integer butterfly -, 0 every prime on the integer back end (single-class kernels, the
        // unguarded butterflies where the prime's size allows), 1 every prime on the double-precision one.
        int cls_hint = -1;
    };

    // out_range: 0 = canonical [0,q); 1 = lazy ([0,4q) forward / [0,2q) inverse).
    hipError_t ntt_forward(const NttTables &t, const NttBatch &b, int out_lazy, hipStream_t stream);
    hipError_t ntt_inverse(const NttTables &t, const NttBatch &b, int out_lazy, hipStream_t stream);
} // namespace sealhip
