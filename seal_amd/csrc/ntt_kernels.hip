// Negacyclic NTT / INTT kernels for gfx950.  See ntt_kernels.h for the decomposition.
//
// Butterflies follow the reference's lazy ranges (Arithmetic<>, native/src/seal/util/ntt.h:20-67):
//   forward (Cooley-Tukey):   X,Y in [0,4q) -> X' = guard(X); T = W*Y in [0,2q); (X'+T, X'-T+2q)
//   inverse (Gentleman-Sande): X,Y in [0,2q) -> (guard(X+Y), W^-1 * (X-Y+2q))
// with N^-1 folded into the last inverse stage (dwthandler.h:273-314).  Only canonical values
// leave through the non-lazy entry points, so results equal ntt_negacyclic_harvey /
// inverse_ntt_negacyclic_harvey (ntt.cpp:408-475) bit for bit.
#include "ntt_kernels.h"
#include "ntt2_kernels.h"
#include "pool.h"

namespace sealhip
{
    namespace
    {
        struct PassArgs
        {
            uint64_t *data;
            const uint64_t *src; // forward first pass only (may be null)
            size_t outer_stride;
            size_t src_outer_stride;
            unsigned src_ncomp;
            uint64_t src_half;
            uint64_t src_q;
            const uint64_t *src_fix;
            const uint32_t *comp_prime;
            unsigned prime_first;
            const ModDesc *mods;
            const ShoupOp *tw;   // fwd or inv tables, [prime][N]
            const ShoupOp *ninv; // [prime][2]
            int log_n;           // n
            int s0;              // first global stage handled by this pass
            int log_wd;          // log2(sub-transforms per workgroup)
            int flags;           // bit0: lazy output, bit1: this pass ends the transform,
                                 // bit2: reduce source mod q on load, bit3: round-shift load
        };
        enum
        {
            F_LAZY = 1,
            F_FINAL = 2,
            F_REDUCE_SRC = 4,
            F_ROUND_SRC = 8
        };

        // Map a coefficient read from a foreign-modulus source into the target prime.
        __device__ __forceinline__ uint64_t load_map(uint64_t val, const PassArgs &a, const ModDesc &md, unsigned comp)
        {
            if (a.flags & F_ROUND_SRC)
            {
                uint64_t s = csub(val + a.src_half, a.src_q);
                return barrett64(s, md) + a.src_fix[comp];
            }
            if (a.flags & F_REDUCE_SRC)
                return barrett64(val, md);
            return val;
        }

        __device__ __forceinline__ uint64_t guard2q(uint64_t x, uint64_t two_q)
        {
            return x >= two_q ? x - two_q : x;
        }

        // LDS position of coefficient M (D bits) of sub-transform u.
        template <int D, int R, bool LAST>
        __device__ __forceinline__ unsigned lds_pos(unsigned M, unsigned u, int log_wd)
        {
            if (LAST)
            {
                // row-major per sub-transform; XOR-swizzle the low R bits with the next R bits so
                // that both "stride 2^(D-R)" (first phase) and "2^R contiguous per lane" (last
                // phase) accesses spread over all banks; pad rows by 16 words.
                constexpr unsigned E = 1u << R;
                unsigned sw = M ^ ((M >> R) & (E - 1));
                return u * ((1u << D) + 16u) + sw;
            }
            else
            {
                // coefficient-major, sub-transform (column) fastest; one 16-word pad per 2^R rows
                return (M << log_wd) + u + ((M >> R) << 4);
            }
        }
        template <int D, int R, bool LAST>
        constexpr size_t lds_words(int log_wd)
        {
            return LAST ? (size_t(1) << log_wd) * ((size_t(1) << D) + 16) : ((size_t(1) << D) << log_wd) + ((size_t(1) << (D - R)) << 4);
        }

        // ------------------------------------------------------------------------------------
        // One pass of D stages, forward.  LAST: contiguous rows (c == 0); else strided columns.
        // ------------------------------------------------------------------------------------
        template <int D, int R, bool LAST>
        __global__ void __launch_bounds__((D - R > 8) ? (1 << (D - R)) : 256) ntt_fwd_pass(PassArgs a)
        {
            constexpr int E = 1 << R;
            constexpr int P = (D + R - 1) / R;
            constexpr unsigned TPS = 1u << (D - R > 0 ? D - R : 0); // threads per sub-transform
            HIP_DYNAMIC_SHARED(uint64_t, lds)

            const int n = a.log_n;
            const int s0 = a.s0;
            const int c = n - s0 - D; // contiguous low bits below this pass (0 for LAST)
            const unsigned comp = blockIdx.y;
            const unsigned outer = blockIdx.z;
            const unsigned prime = a.comp_prime ? a.comp_prime[comp] : a.prime_first + comp;
            const ModDesc md = a.mods[prime];
            const uint64_t q = md.q, two_q = md.two_q;
            const ShoupOp *tw = a.tw + ((size_t)prime << n);
            uint64_t *base = a.data + (size_t)outer * a.outer_stride + ((size_t)comp << n);

            const unsigned tid = threadIdx.x;
            const unsigned wd_mask = (1u << a.log_wd) - 1;
            unsigned u, v;
            if (LAST)
            {
                v = tid & (TPS - 1);
                u = tid >> (D - R > 0 ? D - R : 0);
            }
            else
            {
                u = tid & wd_mask;
                v = tid >> a.log_wd;
            }
            // tile coordinates
            size_t tile_off; // word offset of (u=0, M=0) inside the transform
            unsigned hrow;   // H of this thread's sub-transform (twiddle row)
            if (LAST)
            {
                unsigned h0 = blockIdx.x << a.log_wd;
                hrow = h0 + u;
                tile_off = (size_t)h0 << D;
            }
            else
            {
                unsigned lwb_per = 1u << (c - a.log_wd);
                unsigned H = blockIdx.x / lwb_per;
                unsigned lwb = blockIdx.x % lwb_per;
                hrow = H;
                tile_off = ((size_t)H << (n - s0)) + ((size_t)lwb << a.log_wd);
            }

            uint64_t x[E];
#pragma unroll
            for (int p = 0; p < P; p++)
            {
                constexpr int dummy = 0;
                (void)dummy;
                const int o = p * R;
                const int r = (D - o) < R ? (D - o) : R;
                const bool last_phase = (o + r == D);
                const int sh = last_phase ? 0 : D - o - R;
                const unsigned hi = v >> sh;
                const unsigned lo = v & ((1u << sh) - 1);
                const unsigned mbase = (hi << (sh + R)) | lo;

                // ---- fetch
                if (p == 0)
                {
                    if (a.src)
                    {
                        const uint64_t *sb = a.src + (size_t)outer * a.src_outer_stride + ((size_t)(comp % a.src_ncomp) << n);
#pragma unroll
                        for (int e = 0; e < E; e++)
                        {
                            unsigned M = mbase | ((unsigned)e << sh);
                            size_t g = LAST ? tile_off + ((size_t)u << D) + M : tile_off + ((size_t)M << c) + u;
                            x[e] = load_map(sb[g], a, md, comp);
                        }
                    }
                    else
                    {
#pragma unroll
                        for (int e = 0; e < E; e++)
                        {
                            unsigned M = mbase | ((unsigned)e << sh);
                            size_t g = LAST ? tile_off + ((size_t)u << D) + M : tile_off + ((size_t)M << c) + u;
                            x[e] = base[g];
                        }
                    }
                }
                else
                {
                    __syncthreads();
#pragma unroll
                    for (int e = 0; e < E; e++)
                    {
                        unsigned M = mbase | ((unsigned)e << sh);
                        x[e] = lds[lds_pos<D, R, LAST>(M, u, a.log_wd)];
                    }
                }

                // ---- r radix-2 stages in registers
#pragma unroll
                for (int t = 0; t < r; t++)
                {
                    const int as = o + t;                 // stage offset inside the pass
                    const int b = D - 1 - as - sh;        // bit of e that this stage pairs
                    const unsigned trow = (1u << (s0 + as)) + (hrow << as) + (hi << (R - (b + 1)));
#pragma unroll
                    for (int e0 = 0; e0 < E; e0++)
                    {
                        if (e0 & (1 << b))
                            continue;
                        const int e1 = e0 | (1 << b);
                        const ShoupOp w = tw[trow + ((unsigned)e0 >> (b + 1))];
                        uint64_t X = guard2q(x[e0], two_q);
                        uint64_t T = mul_shoup_lazy(x[e1], w.w, w.wq, q);
                        x[e0] = X + T;
                        x[e1] = X - T + two_q;
                    }
                }

                // ---- hand over
                if (!last_phase)
                {
#pragma unroll
                    for (int e = 0; e < E; e++)
                    {
                        unsigned M = mbase | ((unsigned)e << sh);
                        lds[lds_pos<D, R, LAST>(M, u, a.log_wd)] = x[e];
                    }
                }
                else
                {
                    if ((a.flags & F_FINAL) && !(a.flags & F_LAZY))
                    {
#pragma unroll
                        for (int e = 0; e < E; e++)
                            x[e] = csub(guard2q(x[e], two_q), q);
                    }
                    if (LAST)
                    {
                        // regs -> LDS (own positions) -> coalesced global store
#pragma unroll
                        for (int e = 0; e < E; e++)
                        {
                            unsigned M = mbase | ((unsigned)e << sh);
                            lds[lds_pos<D, R, LAST>(M, u, a.log_wd)] = x[e];
                        }
                        __syncthreads();
                        const unsigned nthreads = blockDim.x;
#pragma unroll
                        for (int k = 0; k < E; k++)
                        {
                            unsigned g = k * nthreads + tid;
                            unsigned uu = g >> D, MM = g & ((1u << D) - 1);
                            base[tile_off + g] = lds[lds_pos<D, R, LAST>(MM, uu, a.log_wd)];
                        }
                    }
                    else
                    {
#pragma unroll
                        for (int e = 0; e < E; e++)
                        {
                            unsigned M = mbase | ((unsigned)e << sh);
                            base[tile_off + ((size_t)M << c) + u] = x[e];
                        }
                    }
                }
            }
        }

        // ------------------------------------------------------------------------------------
        // One pass of D stages, inverse (stages run from the bottom of the pass upwards).
        // ------------------------------------------------------------------------------------
        template <int D, int R, bool LAST>
        __global__ void __launch_bounds__((D - R > 8) ? (1 << (D - R)) : 256) ntt_inv_pass(PassArgs a)
        {
            constexpr int E = 1 << R;
            constexpr int P = (D + R - 1) / R;
            constexpr unsigned TPS = 1u << (D - R > 0 ? D - R : 0);
            HIP_DYNAMIC_SHARED(uint64_t, lds)

            const int n = a.log_n;
            const int s0 = a.s0;
            const int c = n - s0 - D;
            const unsigned comp = blockIdx.y;
            const unsigned outer = blockIdx.z;
            const unsigned prime = a.comp_prime ? a.comp_prime[comp] : a.prime_first + comp;
            const ModDesc md = a.mods[prime];
            const uint64_t q = md.q, two_q = md.two_q;
            const ShoupOp *tw = a.tw + ((size_t)prime << n);
            uint64_t *base = a.data + (size_t)outer * a.outer_stride + ((size_t)comp << n);

            const unsigned tid = threadIdx.x;
            const unsigned wd_mask = (1u << a.log_wd) - 1;
            unsigned u, v;
            if (LAST)
            {
                v = tid & (TPS - 1);
                u = tid >> (D - R > 0 ? D - R : 0);
            }
            else
            {
                u = tid & wd_mask;
                v = tid >> a.log_wd;
            }
            size_t tile_off;
            unsigned hrow;
            if (LAST)
            {
                unsigned h0 = blockIdx.x << a.log_wd;
                hrow = h0 + u;
                tile_off = (size_t)h0 << D;
            }
            else
            {
                unsigned lwb_per = 1u << (c - a.log_wd);
                unsigned H = blockIdx.x / lwb_per;
                unsigned lwb = blockIdx.x % lwb_per;
                hrow = H;
                tile_off = ((size_t)H << (n - s0)) + ((size_t)lwb << a.log_wd);
            }

            uint64_t x[E];
#pragma unroll
            for (int pp = 0; pp < P; pp++)
            {
                const int p = P - 1 - pp;
                const int o = p * R;
                const int r = (D - o) < R ? (D - o) : R;
                const bool last_phase = (o + r == D); // geometrically last = executed first here
                const int sh = last_phase ? 0 : D - o - R;
                const unsigned hi = v >> sh;
                const unsigned lo = v & ((1u << sh) - 1);
                const unsigned mbase = (hi << (sh + R)) | lo;

                // ---- fetch
                if (pp == 0)
                {
                    if (LAST)
                    {
                        // coalesced global -> LDS, then pick up own (contiguous-per-lane) positions
                        const unsigned nthreads = blockDim.x;
#pragma unroll
                        for (int k = 0; k < E; k++)
                        {
                            unsigned g = k * nthreads + tid;
                            unsigned uu = g >> D, MM = g & ((1u << D) - 1);
                            lds[lds_pos<D, R, LAST>(MM, uu, a.log_wd)] = base[tile_off + g];
                        }
                        __syncthreads();
#pragma unroll
                        for (int e = 0; e < E; e++)
                        {
                            unsigned M = mbase | ((unsigned)e << sh);
                            x[e] = lds[lds_pos<D, R, LAST>(M, u, a.log_wd)];
                        }
                    }
                    else
                    {
#pragma unroll
                        for (int e = 0; e < E; e++)
                        {
                            unsigned M = mbase | ((unsigned)e << sh);
                            x[e] = base[tile_off + ((size_t)M << c) + u];
                        }
                    }
                }
                else
                {
                    __syncthreads();
#pragma unroll
                    for (int e = 0; e < E; e++)
                    {
                        unsigned M = mbase | ((unsigned)e << sh);
                        x[e] = lds[lds_pos<D, R, LAST>(M, u, a.log_wd)];
                    }
                }

                // ---- r stages, bottom-up
#pragma unroll
                for (int tt = 0; tt < r; tt++)
                {
                    const int t = r - 1 - tt;
                    const int as = o + t;
                    const int b = D - 1 - as - sh;
                    const unsigned trow = (1u << (s0 + as)) + (hrow << as) + (hi << (R - (b + 1)));
                    const bool scale_stage = (s0 + as == 0); // global stage 0 carries N^-1
#pragma unroll
                    for (int e0 = 0; e0 < E; e0++)
                    {
                        if (e0 & (1 << b))
                            continue;
                        const int e1 = e0 | (1 << b);
                        uint64_t X = x[e0], Y = x[e1];
                        if (scale_stage)
                        {
                            const ShoupOp ni = a.ninv[2 * prime], nw = a.ninv[2 * prime + 1];
                            x[e0] = mul_shoup_lazy(X + Y, ni.w, ni.wq, q);
                            x[e1] = mul_shoup_lazy(X - Y + two_q, nw.w, nw.wq, q);
                        }
                        else
                        {
                            const ShoupOp w = tw[trow + ((unsigned)e0 >> (b + 1))];
                            x[e0] = guard2q(X + Y, two_q);
                            x[e1] = mul_shoup_lazy(X - Y + two_q, w.w, w.wq, q);
                        }
                    }
                }

                // ---- hand over
                if (pp < P - 1)
                {
#pragma unroll
                    for (int e = 0; e < E; e++)
                    {
                        unsigned M = mbase | ((unsigned)e << sh);
                        lds[lds_pos<D, R, LAST>(M, u, a.log_wd)] = x[e];
                    }
                }
                else
                {
                    if ((a.flags & F_FINAL) && !(a.flags & F_LAZY))
                    {
#pragma unroll
                        for (int e = 0; e < E; e++)
                            x[e] = csub(x[e], q);
                    }
#pragma unroll
                    for (int e = 0; e < E; e++)
                    {
                        unsigned M = mbase | ((unsigned)e << sh);
                        size_t g = LAST ? tile_off + ((size_t)u << D) + M : tile_off + ((size_t)M << c) + u;
                        base[g] = x[e];
                    }
                }
            }
        }

        // ------------------------------------------------------------------------------------
        // Small transforms (n <= 10): one workgroup per transform, radix-2 in LDS.  Used for the
        // tiny degrees of the reference's own unit tests (N = 2 ... 32) where tiles degenerate.
        // ------------------------------------------------------------------------------------
        __global__ void ntt_small(PassArgs a, int inverse)
        {
            HIP_DYNAMIC_SHARED(uint64_t, lds)
            const int n = a.log_n;
            const unsigned N = 1u << n;
            const unsigned comp = blockIdx.y, outer = blockIdx.z;
            const unsigned prime = a.comp_prime ? a.comp_prime[comp] : a.prime_first + comp;
            const ModDesc md = a.mods[prime];
            const uint64_t q = md.q, two_q = md.two_q;
            const ShoupOp *tw = a.tw + ((size_t)prime << n);
            uint64_t *base = a.data + (size_t)outer * a.outer_stride + ((size_t)comp << n);
            const uint64_t *in = base;
            if (a.src)
                in = a.src + (size_t)outer * a.src_outer_stride + ((size_t)(comp % a.src_ncomp) << n);
            for (unsigned i = threadIdx.x; i < N; i += blockDim.x)
            {
                lds[i] = a.src ? load_map(in[i], a, md, comp) : in[i];
            }
            __syncthreads();
            if (!inverse)
            {
                for (int s = 0; s < n; s++)
                {
                    unsigned gap = N >> (s + 1);
                    for (unsigned bfl = threadIdx.x; bfl < N / 2; bfl += blockDim.x)
                    {
                        unsigned grp = bfl / gap, j = bfl % gap;
                        unsigned i0 = grp * 2 * gap + j, i1 = i0 + gap;
                        ShoupOp w = tw[(1u << s) + grp];
                        uint64_t X = guard2q(lds[i0], two_q);
                        uint64_t T = mul_shoup_lazy(lds[i1], w.w, w.wq, q);
                        lds[i0] = X + T;
                        lds[i1] = X - T + two_q;
                    }
                    __syncthreads();
                }
                for (unsigned i = threadIdx.x; i < N; i += blockDim.x)
                {
                    uint64_t val = lds[i];
                    if (!(a.flags & F_LAZY))
                        val = csub(guard2q(val, two_q), q);
                    base[i] = val;
                }
            }
            else
            {
                for (int s = n - 1; s >= 0; s--)
                {
                    unsigned gap = N >> (s + 1);
                    for (unsigned bfl = threadIdx.x; bfl < N / 2; bfl += blockDim.x)
                    {
                        unsigned grp = bfl / gap, j = bfl % gap;
                        unsigned i0 = grp * 2 * gap + j, i1 = i0 + gap;
                        uint64_t X = lds[i0], Y = lds[i1];
                        if (s == 0)
                        {
                            const ShoupOp ni = a.ninv[2 * prime], nw = a.ninv[2 * prime + 1];
                            lds[i0] = mul_shoup_lazy(X + Y, ni.w, ni.wq, q);
                            lds[i1] = mul_shoup_lazy(X - Y + two_q, nw.w, nw.wq, q);
                        }
                        else
                        {
                            ShoupOp w = tw[(1u << s) + grp];
                            lds[i0] = guard2q(X + Y, two_q);
                            lds[i1] = mul_shoup_lazy(X - Y + two_q, w.w, w.wq, q);
                        }
                    }
                    __syncthreads();
                }
                for (unsigned i = threadIdx.x; i < N; i += blockDim.x)
                {
                    uint64_t val = lds[i];
                    if (!(a.flags & F_LAZY))
                        val = csub(val, q);
                    base[i] = val;
                }
            }
        }

        // ---- pass plan -----------------------------------------------------------------------
        struct Plan
        {
            int passes;   // 1 or 2 (0 = small kernel)
            int d_col;    // stages of the column pass (two-pass only)
            int d_last;   // stages of the contiguous pass
        };
        Plan make_plan(int n)
        {
            if (n < 6)
                return { 0, 0, 0 };
            if (n <= 13)
                return { 1, 0, n };
            switch (n)
            {
            case 14:
                return { 2, 7, 7 };
            case 15:
                return { 2, 7, 8 };
            case 16:
                return { 2, 8, 8 };
            default:
                return { 2, 8, 9 };
            }
        }

        template <int D, int R, bool LAST, bool INV>
        hipError_t launch_pass(PassArgs a, unsigned ncomp, unsigned nouter, hipStream_t stream)
        {
            constexpr int TPS_LOG = D - R > 0 ? D - R : 0;
            const int n = a.log_n;
            const int c = n - a.s0 - D;
            // sub-transforms available along the tile's "width" dimension
            int avail_log = LAST ? a.s0 : c;
            int log_wd = 8 - TPS_LOG; // aim at 256 threads
            if (log_wd < 0)
                log_wd = 0;
            if (log_wd > avail_log)
                log_wd = avail_log;
            a.log_wd = log_wd;
            unsigned threads = 1u << (TPS_LOG + log_wd);
            unsigned tiles = LAST ? (1u << (a.s0 - log_wd)) : (1u << (a.s0 + c - log_wd));
            size_t shmem = lds_words<D, R, LAST>(log_wd) * sizeof(uint64_t);
            dim3 grid(tiles, ncomp, nouter);
            if (INV)
                hipLaunchKernelGGL((ntt_inv_pass<D, R, LAST>), grid, dim3(threads), shmem, stream, a);
            else
                hipLaunchKernelGGL((ntt_fwd_pass<D, R, LAST>), grid, dim3(threads), shmem, stream, a);
            return hipGetLastError();
        }

        template <bool LAST, bool INV>
        hipError_t dispatch_pass(int D, const PassArgs &a, unsigned ncomp, unsigned nouter, hipStream_t stream)
        {
            switch (D)
            {
            case 6:
                return launch_pass<6, 3, LAST, INV>(a, ncomp, nouter, stream);
            case 7:
                return launch_pass<7, 4, LAST, INV>(a, ncomp, nouter, stream);
            case 8:
                return launch_pass<8, 4, LAST, INV>(a, ncomp, nouter, stream);
            case 9:
                return launch_pass<9, 3, LAST, INV>(a, ncomp, nouter, stream);
            case 10:
                return launch_pass<10, 5, LAST, INV>(a, ncomp, nouter, stream);
            case 11:
                return launch_pass<11, 4, LAST, INV>(a, ncomp, nouter, stream);
            case 12:
                return launch_pass<12, 4, LAST, INV>(a, ncomp, nouter, stream);
            case 13:
                return launch_pass<13, 4, LAST, INV>(a, ncomp, nouter, stream);
            default:
                return hipErrorInvalidValue;
            }
        }

        hipError_t run(const NttTables &t, const NttBatch &b, int out_lazy, bool inverse, hipStream_t stream)
        {
            if (b.ncomp == 0 || b.nouter == 0)
                return hipSuccess;
            PassArgs a;
            a.data = b.data;
            a.src = inverse ? nullptr : b.src;
            a.outer_stride = b.outer_stride;
            a.src_outer_stride = b.src_outer_stride;
            a.src_ncomp = b.src_ncomp ? b.src_ncomp : 1;
            a.src_half = b.src_half;
            a.src_q = b.src_q;
            a.src_fix = b.src_fix;
            a.comp_prime = b.comp_prime;
            a.prime_first = b.prime_first;
            a.mods = t.mods;
            a.tw = inverse ? t.inv : t.fwd;
            a.ninv = t.ninv;
            a.log_n = t.log_n;
            a.s0 = 0;
            a.log_wd = 0;
            const int lazy = out_lazy ? F_LAZY : 0;
            const int srcflag = !a.src ? 0 : (b.src_mode == 2 ? F_ROUND_SRC : (b.src_mode == 1 ? F_REDUCE_SRC : 0));
            const Plan plan = make_plan(t.log_n);
            const unsigned zmax = 65535;
            for (unsigned z0 = 0; z0 < b.nouter; z0 += zmax)
            {
                unsigned nz = b.nouter - z0 < zmax ? b.nouter - z0 : zmax;
                PassArgs az = a;
                az.data = a.data + (size_t)z0 * a.outer_stride;
                if (az.src)
                    az.src = a.src + (size_t)z0 * a.src_outer_stride;
                hipError_t e = hipSuccess;
                if (plan.passes == 0)
                {
                    az.flags = lazy | F_FINAL | srcflag;
                    unsigned N = 1u << t.log_n;
                    unsigned threads = N / 2 < 64 ? 64 : N / 2;
                    hipLaunchKernelGGL(ntt_small, dim3(1, b.ncomp, nz), dim3(threads), N * sizeof(uint64_t), stream, az, inverse ? 1 : 0);
                    e = hipGetLastError();
                }
                else if (plan.passes == 1)
                {
                    az.flags = lazy | F_FINAL | srcflag;
                    az.s0 = 0;
                    e = inverse ? dispatch_pass<true, true>(plan.d_last, az, b.ncomp, nz, stream)
                                : dispatch_pass<true, false>(plan.d_last, az, b.ncomp, nz, stream);
                }
                else if (!inverse)
                {
                    az.flags = srcflag;
                    az.s0 = 0;
                    e = dispatch_pass<false, false>(plan.d_col, az, b.ncomp, nz, stream);
                    if (e != hipSuccess)
                        return e;
                    az.src = nullptr;
                    az.flags = lazy | F_FINAL;
                    az.s0 = plan.d_col;
                    e = dispatch_pass<true, false>(plan.d_last, az, b.ncomp, nz, stream);
                }
                else
                {
                    az.flags = 0;
                    az.s0 = plan.d_col;
                    e = dispatch_pass<true, true>(plan.d_last, az, b.ncomp, nz, stream);
                    if (e != hipSuccess)
                        return e;
                    az.flags = lazy | F_FINAL;
                    az.s0 = 0;
                    e = dispatch_pass<false, true>(plan.d_col, az, b.ncomp, nz, stream);
                }
                if (e != hipSuccess)
                    return e;
            }
            return hipSuccess;
        }
    } // namespace

    hipError_t ntt_forward(const NttTables &t, const NttBatch &b, int out_lazy, hipStream_t stream)
    {
        if ((b.epi || b.tail2) && !ntt2_supports(t.log_n))
            return hipErrorInvalidValue;
        if (ntt2_supports(t.log_n))
        {
            // (round 5: chunks of the outer items sized to the Infinity Cache, dealt to forked streams, measured 4 - 27 % SLOWER than
            // one launch over the whole batch - profiles/r05_ntt_mall_chunks.txt - and are not here)
            // two-pass engine; the scratch block goes back to the pool in stream order
            Scratch mid(((size_t)b.nouter * b.ncomp) << t.log_n);
            // N = 2^16, large plain batches: the double-precision components run as ONE launch through a re-used ring (round 6)
            const size_t ring_words = ntt2_ring_words(t, b);
            if (ring_words)
            {
                Scratch ring(ring_words);
                return ntt2_forward(t, b, out_lazy, mid.p, stream, ring.p, ring_words);
            }
            return ntt2_forward(t, b, out_lazy, mid.p, stream);
        }
        return run(t, b, out_lazy, false, stream);
    }
    hipError_t ntt_inverse(const NttTables &t, const NttBatch &b, int out_lazy, hipStream_t stream)
    {
        if (ntt2_supports(t.log_n))
        {
            Scratch mid(((size_t)b.nouter * b.ncomp) << t.log_n);
            return ntt2_inverse(t, b, out_lazy, mid.p, stream);
        }
        if (b.src || b.out_add || b.prod_x)
            return hipErrorInvalidValue; // out-of-place input, out_add and the product source are features of the two-pass engine only
        return run(t, b, out_lazy, true, stream);
    }
} // namespace sealhip
