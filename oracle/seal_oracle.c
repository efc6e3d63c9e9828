/*
 * seal_oracle.c — TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C CPU restatement of the reference's RNS polynomial-arithmetic hot path (Microsoft SEAL
 * 4.4.3), written from the closed-form specification the survey verified against the compiled
 * reference (SURVEY.md §8(a')): exact modular arithmetic with unsigned __int128 and '%', no lazy
 * ranges, no Shoup/Barrett tricks — deliberately a different code shape from the HIP kernels it
 * checks.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load the
 * resulting libsealoracle.so; the product (seal_amd/) never does.
 *
 * PINNING: this restatement is itself checked (tests/test_oracle.py) against
 *   (1) the reference's own known-answer tests: NTTPrimitiveRootsTest / NegacyclicNTTTest
 *       (native/tests/seal/util/ntt.cpp:53-101), GaloisTool ApplyGalois/ApplyGaloisNTT
 *       (native/tests/seal/util/galois.cpp:86-120), BaseConverter / RNSTool vectors
 *       (native/tests/seal/util/rns.cpp:347-1011), and
 *   (2) golden input/output vectors produced by the REAL reference (oracle/_ref, compiled from
 *       /root/reference) and committed under tests/golden/ with their generator script, and
 *   (3) live, against oracle/_ref/libsealref.so whenever that library is present.
 *
 * Each function cites the reference code whose result it reproduces.
 */
#include "seal_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef unsigned __int128 u128;

/* ---------------------------------------------------------------- word arithmetic (uintarithsmallmod.h) */
static uint64_t mulmod(uint64_t a, uint64_t b, uint64_t m)
{
    return (uint64_t)((u128)a * b % m);
}
static uint64_t addmod(uint64_t a, uint64_t b, uint64_t m)
{
    return (uint64_t)(((u128)a + b) % m);
}
static uint64_t submod(uint64_t a, uint64_t b, uint64_t m)
{
    a %= m;
    b %= m;
    return a >= b ? a - b : a + m - b;
}
static uint64_t powmod(uint64_t a, uint64_t e, uint64_t m)
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
/* try_invert_uint_mod (numth.h): extended Euclid, works for non-prime m (2^32, 2N) */
static uint64_t invmod(uint64_t a, uint64_t m)
{
    __int128 t = 0, nt = 1, r = m, nr = a % m;
    while (nr)
    {
        __int128 q = r / nr, tmp = t - q * nt;
        t = nt;
        nt = tmp;
        tmp = r - q * nr;
        r = nr;
        nr = tmp;
    }
    if (t < 0)
        t += m;
    return (uint64_t)t;
}
static int bitcount(uint64_t v)
{
    int c = 0;
    while (v)
    {
        c++;
        v >>= 1;
    }
    return c;
}
static uint32_t bitrev(uint32_t x, int bits)
{
    uint32_t r = 0;
    for (int i = 0; i < bits; i++)
        r |= ((x >> i) & 1u) << (bits - 1 - i);
    return r;
}

/* Modulus::is_prime (modulus.cpp): Miller-Rabin; the fixed base set below is exact below 2^64 */
static int is_prime(uint64_t n)
{
    static const uint64_t bases[] = { 2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37 };
    if (n < 2)
        return 0;
    for (int i = 0; i < 12; i++)
    {
        if (n == bases[i])
            return 1;
        if (n % bases[i] == 0)
            return 0;
    }
    uint64_t d = n - 1;
    int r = 0;
    while (!(d & 1))
    {
        d >>= 1;
        r++;
    }
    for (int i = 0; i < 12; i++)
    {
        uint64_t x = powmod(bases[i], d, n);
        if (x == 1 || x == n - 1)
            continue;
        int comp = 1;
        for (int k = 1; k < r; k++)
        {
            x = mulmod(x, x, n);
            if (x == n - 1)
            {
                comp = 0;
                break;
            }
        }
        if (comp)
            return 0;
    }
    return 1;
}

/* get_primes (numth.cpp:278-311) */
int so_get_primes(uint64_t factor, int bit_size, int count, uint64_t *out)
{
    uint64_t value = (((uint64_t)1 << bit_size) - 1) / factor * factor + 1;
    uint64_t lower = (uint64_t)1 << (bit_size - 1);
    int found = 0;
    while (found < count && value > lower)
    {
        if (is_prime(value))
            out[found++] = value;
        value -= factor;
    }
    return found == count ? 0 : -1;
}

/* CoeffModulus::Create (modulus.cpp:143-184): per bit size the primes come in descending order and
 * are handed out from the back */
int so_coeff_modulus_create(uint64_t n, const int *bit_sizes, int count, uint64_t *out)
{
    for (int i = 0; i < count; i++)
        out[i] = 0;
    for (int i = 0; i < count; i++)
    {
        if (out[i])
            continue;
        int b = bit_sizes[i], cnt = 0;
        for (int j = 0; j < count; j++)
            cnt += bit_sizes[j] == b;
        uint64_t tmp[SO_MAX_PRIMES];
        if (so_get_primes(2 * n, b, cnt, tmp))
            return -1;
        int k = cnt - 1;
        for (int j = 0; j < count; j++)
            if (bit_sizes[j] == b)
                out[j] = tmp[k--];
    }
    return 0;
}

/* try_minimal_primitive_root (numth.cpp:386-413): smallest primitive degree-th root */
static uint64_t min_primitive_root(uint64_t degree, uint64_t q)
{
    uint64_t r = 0;
    for (uint64_t g = 2;; g++)
    {
        r = powmod(g, (q - 1) / degree, q);
        if (powmod(r, degree / 2, q) == q - 1)
            break;
    }
    uint64_t sq = mulmod(r, r, q), cur = r, best = r;
    for (uint64_t i = 0; i < degree; i += 2)
    {
        if (cur < best)
            best = cur;
        cur = mulmod(cur, sq, q);
    }
    return best;
}

/* ---------------------------------------------------------------- context */
typedef struct
{
    int nB, nBsk;              /* |B|, |B|+1 */
    uint64_t bsk[SO_MAX_PRIMES + 2]; /* B..., m_sk */
    uint64_t bsk_psi[SO_MAX_PRIMES + 2];
} so_behz;

struct so_ctx
{
    int scheme;
    uint64_t n;
    int logn;
    int L;
    uint64_t q[SO_MAX_PRIMES];
    uint64_t psi[SO_MAX_PRIMES];
    uint64_t t;
    so_behz behz[SO_MAX_PRIMES + 1]; /* indexed by K (number of data primes at the level) */
};

/* significant bit count of prod(q_0..q_{K-1}) (rns.cpp:612) via school multiplication */
static int prod_bits(const uint64_t *q, int K)
{
    uint64_t acc[SO_MAX_PRIMES + 1];
    int len = 1;
    acc[0] = 1;
    for (int i = 0; i < K; i++)
    {
        uint64_t carry = 0;
        for (int j = 0; j < len; j++)
        {
            u128 v = (u128)acc[j] * q[i] + carry;
            acc[j] = (uint64_t)v;
            carry = (uint64_t)(v >> 64);
        }
        if (carry)
            acc[len++] = carry;
    }
    return (len - 1) * 64 + bitcount(acc[len - 1]);
}

so_ctx *so_ctx_create(int scheme, uint64_t n, const uint64_t *primes, int count, uint64_t t)
{
    so_ctx *c = (so_ctx *)calloc(1, sizeof(so_ctx));
    c->scheme = scheme;
    c->n = n;
    c->logn = bitcount(n) - 1;
    c->L = count;
    c->t = t;
    for (int i = 0; i < count; i++)
    {
        c->q[i] = primes[i];
        c->psi[i] = min_primitive_root(2 * n, primes[i]); /* NTTTables::initialize, ntt.cpp:256 */
    }
    if (scheme == 1)
    {
        /* RNSTool::initialize (rns.cpp:605-648) for every level K = 1..L */
        for (int K = 1; K <= count; K++)
        {
            int nb = K;
            if (32 + bitcount(t) + prod_bits(c->q, K) >= 61 * K + 61)
                nb++;
            uint64_t aux[SO_MAX_PRIMES + 4];
            if (so_get_primes(2 * n, 61, nb + 2, aux))
            {
                free(c);
                return NULL;
            }
            so_behz *b = &c->behz[K];
            b->nB = nb;
            b->nBsk = nb + 1;
            for (int i = 0; i < nb; i++)
                b->bsk[i] = aux[2 + i]; /* aux[0] = m_sk, aux[1] = gamma, then B */
            b->bsk[nb] = aux[0];
            for (int i = 0; i <= nb; i++)
                b->bsk_psi[i] = min_primitive_root(2 * n, b->bsk[i]);
        }
    }
    return c;
}
void so_ctx_destroy(so_ctx *c)
{
    free(c);
}
uint64_t so_ntt_root(const so_ctx *c, int i)
{
    return c->psi[i];
}
int so_base_bsk(const so_ctx *c, int K, uint64_t *out)
{
    const so_behz *b = &c->behz[K];
    for (int i = 0; i < b->nBsk; i++)
        out[i] = b->bsk[i];
    return b->nBsk;
}

/* ---------------------------------------------------------------- NTT (ntt.cpp / dwthandler.h) */
/* NTT(a)[j] = sum_k a[k] psi^{(2 bitrev(j)+1) k}: natural in, bit-reversed out (SURVEY §8(a')).
 * Fast form: Cooley-Tukey with the psi powers taken in bit-reversed order, stage m uses
 * psi^bitrev(m+i) for group i — the result of DWTHandler::transform_to_rev (dwthandler.h:94-191)
 * followed by the final reduction of ntt_negacyclic_harvey (ntt.cpp:408-437). */
static void ntt_fwd_generic(uint64_t *a, uint64_t n, int logn, uint64_t q, uint64_t psi)
{
    uint64_t *tw = (uint64_t *)malloc(n * sizeof(uint64_t));
    uint64_t p = 1;
    for (uint64_t i = 0; i < n; i++)
    {
        tw[bitrev((uint32_t)i, logn)] = p;
        p = mulmod(p, psi, q);
    }
    uint64_t t = n >> 1;
    for (uint64_t m = 1; m < n; m <<= 1, t >>= 1)
        for (uint64_t i = 0; i < m; i++)
        {
            uint64_t w = tw[m + i];
            for (uint64_t j = 2 * i * t; j < 2 * i * t + t; j++)
            {
                uint64_t u = a[j], v = mulmod(a[j + t], w, q);
                a[j] = addmod(u, v, q);
                a[j + t] = submod(u, v, q);
            }
        }
    free(tw);
}
/* inverse: Gentleman-Sande, bit-reversed in, natural out, times N^-1
 * (transform_from_rev dwthandler.h:202-356 + inverse_ntt_negacyclic_harvey ntt.cpp:453-475) */
static void ntt_inv_generic(uint64_t *a, uint64_t n, int logn, uint64_t q, uint64_t psi)
{
    uint64_t *tw = (uint64_t *)malloc(n * sizeof(uint64_t));
    uint64_t ipsi = invmod(psi, q), p = 1;
    for (uint64_t i = 0; i < n; i++)
    {
        tw[bitrev((uint32_t)i, logn)] = p;
        p = mulmod(p, ipsi, q);
    }
    uint64_t t = 1;
    for (uint64_t m = n >> 1; m >= 1; m >>= 1, t <<= 1)
        for (uint64_t i = 0; i < m; i++)
        {
            uint64_t w = tw[m + i];
            for (uint64_t j = 2 * i * t; j < 2 * i * t + t; j++)
            {
                uint64_t u = a[j], v = a[j + t];
                a[j] = addmod(u, v, q);
                a[j + t] = mulmod(submod(u, v, q), w, q);
            }
        }
    uint64_t ninv = invmod(n % q, q);
    for (uint64_t i = 0; i < n; i++)
        a[i] = mulmod(a[i], ninv, q);
    free(tw);
}
void so_ntt_forward(const so_ctx *c, int i, uint64_t *a)
{
    ntt_fwd_generic(a, c->n, c->logn, c->q[i], c->psi[i]);
}
void so_ntt_inverse(const so_ctx *c, int i, uint64_t *a)
{
    ntt_inv_generic(a, c->n, c->logn, c->q[i], c->psi[i]);
}
void so_ntt_forward_naive(const so_ctx *c, int pi, const uint64_t *a, uint64_t *out)
{
    uint64_t q = c->q[pi], n = c->n;
    for (uint64_t j = 0; j < n; j++)
    {
        uint64_t e = 2 * (uint64_t)bitrev((uint32_t)j, c->logn) + 1;
        uint64_t w = powmod(c->psi[pi], e, q), p = 1, s = 0;
        for (uint64_t k = 0; k < n; k++)
        {
            s = addmod(s, mulmod(a[k] % q, p, q), q);
            p = mulmod(p, w, q);
        }
        out[j] = s;
    }
}

/* ---------------------------------------------------------------- element-wise (polyarithsmallmod.cpp) */
void so_dyadic(const so_ctx *c, int pi, const uint64_t *a, const uint64_t *b, uint64_t *r)
{
    for (uint64_t j = 0; j < c->n; j++)
        r[j] = mulmod(a[j], b[j], c->q[pi]); /* dyadic_product_coeffmod :226-284 */
}

/* ckks_multiply (evaluator.cpp:569-708): out[I] = sum_{a+b=I} x[a] . y[b], all in NTT form */
static void tensor(const uint64_t *mods, int K, uint64_t n, const uint64_t *x, int sx, const uint64_t *y, int sy, uint64_t *out)
{
    int dest = sx + sy - 1;
    memset(out, 0, (size_t)dest * K * n * sizeof(uint64_t));
    for (int a = 0; a < sx; a++)
        for (int b = 0; b < sy; b++)
            for (int i = 0; i < K; i++)
            {
                const uint64_t *xp = x + ((size_t)a * K + i) * n, *yp = y + ((size_t)b * K + i) * n;
                uint64_t *op = out + ((size_t)(a + b) * K + i) * n;
                for (uint64_t j = 0; j < n; j++)
                    op[j] = addmod(op[j], mulmod(xp[j] % mods[i], yp[j] % mods[i], mods[i]), mods[i]);
            }
}
void so_ckks_multiply(const so_ctx *c, int K, const uint64_t *x, int sx, const uint64_t *y, int sy, uint64_t *out)
{
    tensor(c->q, K, c->n, x, sx, y, sy, out);
}

/* ---------------------------------------------------------------- Galois (galois.cpp) */
uint32_t so_galois_elt_from_step(const so_ctx *c, int step)
{
    /* GaloisTool::get_elt_from_step :53-95, generator 3 */
    uint64_t m = 2 * c->n;
    if (step == 0)
        return (uint32_t)(m - 1);
    int neg = step < 0;
    uint64_t pos = (uint64_t)(neg ? -step : step);
    uint64_t e = neg ? (c->n >> 1) - pos : pos, elt = 1;
    while (e--)
        elt = (elt * 3) & (m - 1);
    return (uint32_t)elt;
}
void so_apply_galois(const so_ctx *c, int K, int ntt_form, uint32_t elt, const uint64_t *in, uint64_t *out)
{
    uint64_t n = c->n;
    for (int i = 0; i < K; i++)
    {
        const uint64_t *ip = in + (size_t)i * n;
        uint64_t *op = out + (size_t)i * n;
        if (ntt_form)
        {
            /* apply_galois_ntt :192-218 with table generate_table_ntt :18-51 */
            for (uint64_t j = 0; j < n; j++)
            {
                uint32_t rev = bitrev((uint32_t)(j + n), c->logn + 1);
                uint64_t raw = ((uint64_t)elt * rev) >> 1;
                op[j] = ip[bitrev((uint32_t)(raw & (n - 1)), c->logn)];
            }
        }
        else
        {
            /* apply_galois :148-190 */
            for (uint64_t j = 0; j < n; j++)
            {
                uint64_t raw = j * elt, idx = raw & (n - 1), v = ip[j];
                if ((raw >> c->logn) & 1)
                    v = v ? c->q[i] - v : 0;
                op[idx] = v;
            }
        }
    }
}

/* ---------------------------------------------------------------- modulus switching (rns.cpp:789-901) */
void so_rescale(const so_ctx *c, int K, const uint64_t *in, int size, uint64_t *out)
{
    /* divide_and_round_q_last_ntt_inplace: r = (INTT(c_l) + half) mod q_l;
       c'_i = (c_i - NTT_i((r mod q_i - half mod q_i) mod q_i)) * q_l^-1 mod q_i */
    uint64_t n = c->n, ql = c->q[K - 1], half = ql >> 1;
    uint64_t *r = (uint64_t *)malloc(n * 8), *u = (uint64_t *)malloc(n * 8);
    for (int p = 0; p < size; p++)
    {
        const uint64_t *poly = in + (size_t)p * K * n;
        memcpy(r, poly + (size_t)(K - 1) * n, n * 8);
        so_ntt_inverse(c, K - 1, r);
        for (uint64_t j = 0; j < n; j++)
            r[j] = addmod(r[j], half, ql);
        for (int i = 0; i + 1 < K; i++)
        {
            uint64_t qi = c->q[i], inv = invmod(ql % qi, qi);
            for (uint64_t j = 0; j < n; j++)
                u[j] = submod(r[j] % qi, half % qi, qi);
            so_ntt_forward(c, i, u);
            uint64_t *op = out + ((size_t)p * (K - 1) + i) * n;
            for (uint64_t j = 0; j < n; j++)
                op[j] = mulmod(submod(poly[(size_t)i * n + j], u[j], qi), inv, qi);
        }
    }
    free(r);
    free(u);
}
void so_bfv_mod_switch(const so_ctx *c, int K, const uint64_t *in, int size, uint64_t *out)
{
    /* divide_and_round_q_last_inplace :789-828 (coefficient domain) */
    uint64_t n = c->n, ql = c->q[K - 1], half = ql >> 1;
    for (int p = 0; p < size; p++)
    {
        const uint64_t *poly = in + (size_t)p * K * n;
        for (int i = 0; i + 1 < K; i++)
        {
            uint64_t qi = c->q[i], inv = invmod(ql % qi, qi);
            uint64_t *op = out + ((size_t)p * (K - 1) + i) * n;
            for (uint64_t j = 0; j < n; j++)
            {
                uint64_t r = addmod(poly[(size_t)(K - 1) * n + j], half, ql);
                uint64_t u = submod(r % qi, half % qi, qi);
                op[j] = mulmod(submod(poly[(size_t)i * n + j], u, qi), inv, qi);
            }
        }
    }
}
void so_bgv_mod_switch(const so_ctx *c, int K, const uint64_t *in, int size, uint64_t *out)
{
    /* mod_t_and_divide_q_last_ntt_inplace (rns.cpp:1193-1236): c_l = INTT(c[l]); k = -(c_l mod t) q_l^-1 mod t;
       delta_i = ((k mod q_i) q_l + c_l) mod q_i; c'_i = (c_i - NTT_i(delta_i)) q_l^-1 mod q_i */
    uint64_t n = c->n, ql = c->q[K - 1], t = c->t, inv_t = invmod(ql % t, t);
    uint64_t *r = (uint64_t *)malloc(n * 8), *u = (uint64_t *)malloc(n * 8);
    for (int p = 0; p < size; p++)
    {
        const uint64_t *poly = in + (size_t)p * K * n;
        memcpy(r, poly + (size_t)(K - 1) * n, n * 8);
        so_ntt_inverse(c, K - 1, r);
        for (int i = 0; i + 1 < K; i++)
        {
            uint64_t qi = c->q[i], inv = invmod(ql % qi, qi);
            for (uint64_t j = 0; j < n; j++)
            {
                uint64_t k = mulmod(submod(0, r[j] % t, t), inv_t, t);
                u[j] = addmod(mulmod(k % qi, ql % qi, qi), r[j] % qi, qi);
            }
            so_ntt_forward(c, i, u);
            uint64_t *op = out + ((size_t)p * (K - 1) + i) * n;
            for (uint64_t j = 0; j < n; j++)
                op[j] = mulmod(submod(poly[(size_t)i * n + j], u[j], qi), inv, qi);
        }
    }
    free(r);
    free(u);
}
uint64_t so_bgv_mod_switch_correction(const so_ctx *c, int K, uint64_t correction_factor)
{
    /* evaluator.cpp:1286-1292 */
    return mulmod(correction_factor % c->t, invmod(c->q[K - 1] % c->t, c->t), c->t);
}

/* ---------------------------------------------------------------- plaintext operands (evaluator.cpp:1760-2287) */
void so_plain_lift(const so_ctx *c, int K, const uint64_t *m, uint64_t count, uint64_t scale_by, uint64_t *out)
{
    /* the centred lift of transform_to_ntt_inplace(Plaintext) / multiply_plain_normal (:2098-2125, :2243-2282):
       m >= (t+1)/2 represents m - t, i.e. m + (Q - t) modulo every q_i; coefficients beyond `count` are zero */
    uint64_t n = c->n, t = c->t, thr = (t + 1) >> 1;
    for (int i = 0; i < K; i++)
    {
        uint64_t qi = c->q[i], inc = (qi - t % qi) % qi;
        for (uint64_t j = 0; j < n; j++)
        {
            uint64_t v = 0;
            if (j < count)
            {
                uint64_t mv = scale_by == 1 ? m[j] : mulmod(m[j], scale_by % t, t);
                v = mv % qi;
                if (mv >= thr)
                    v = addmod(v, inc, qi);
            }
            out[(size_t)i * n + j] = v;
        }
    }
}
void so_bfv_addsub_plain(const so_ctx *c, int K, uint64_t *c0, const uint64_t *m, uint64_t count, int sub)
{
    /* multiply_add/sub_plain_with_scaling_variant (util/scalingvariant.cpp:70-175):
       fix = floor((m (Q mod t) + (t+1)/2) / t); c0_i +/-= (m floor(Q/t) + fix) mod q_i */
    uint64_t n = c->n, t = c->t, thr = (t + 1) >> 1;
    uint64_t q_mod_t = 1;
    for (int i = 0; i < K; i++)
        q_mod_t = mulmod(q_mod_t, c->q[i] % t, t);
    for (int i = 0; i < K; i++)
    {
        uint64_t qi = c->q[i];
        /* floor(Q/t) = (Q - Q mod t)/t and Q = 0 (mod q_i): floor(Q/t) = -(Q mod t) t^-1 (mod q_i) */
        uint64_t delta = mulmod(submod(0, q_mod_t % qi, qi), invmod(t % qi, qi), qi);
        for (uint64_t j = 0; j < count && j < n; j++)
        {
            unsigned __int128 num = (unsigned __int128)m[j] * q_mod_t + thr;
            uint64_t fix = (uint64_t)(num / t);
            uint64_t scaled = addmod(mulmod(m[j] % qi, delta, qi), fix % qi, qi);
            uint64_t *p = c0 + (size_t)i * n + j;
            *p = sub ? submod(*p, scaled, qi) : addmod(*p, scaled, qi);
        }
    }
}

void so_drop_last(const so_ctx *c, int K, const uint64_t *in, int size, uint64_t *out)
{
    /* mod_switch_drop_to_next (evaluator.cpp:1296-1367) */
    for (int p = 0; p < size; p++)
        memcpy(out + (size_t)p * (K - 1) * c->n, in + (size_t)p * K * c->n, (size_t)(K - 1) * c->n * 8);
}

/* ---------------------------------------------------------------- key switching (evaluator.cpp:2561-2867) */
void so_switch_key(const so_ctx *c, int K, uint64_t *ct, const uint64_t *target, const uint64_t *key)
{
    const uint64_t n = c->n;
    const int L = c->L;
    const uint64_t P = c->q[L - 1], half = P >> 1;
    uint64_t *t = (uint64_t *)malloc((size_t)K * n * 8);
    uint64_t *S = (uint64_t *)calloc((size_t)2 * (K + 1) * n, 8); /* S[k][I] */
    uint64_t *tmp = (uint64_t *)malloc(n * 8);
    memcpy(t, target, (size_t)K * n * 8);
    if (c->scheme != 1)
        for (int J = 0; J < K; J++)
            so_ntt_inverse(c, J, t + (size_t)J * n); /* CKKS and BGV targets are in NTT form :2651-2658 */
    for (int I = 0; I <= K; I++)
    {
        int pi = I == K ? L - 1 : I; /* key_index :2664 */
        uint64_t m = c->q[pi];
        for (int J = 0; J < K; J++)
        {
            for (uint64_t j = 0; j < n; j++)
                tmp[j] = t[(size_t)J * n + j] % m; /* :2690-2699 */
            so_ntt_forward(c, pi, tmp);
            for (int k = 0; k < 2; k++)
            {
                const uint64_t *kp = key + (((size_t)J * 2 + k) * L + pi) * n;
                uint64_t *sp = S + ((size_t)k * (K + 1) + I) * n;
                for (uint64_t j = 0; j < n; j++)
                    sp[j] = addmod(sp[j], mulmod(tmp[j], kp[j], m), m); /* :2705-2755 */
            }
        }
    }
    if (c->scheme == 3)
    {
        /* BGV mod-down (:2762-2805): k = -(t_last mod t) P^-1 mod t; delta = (k mod q_i) P + t_last (mod q_i);
           ct_k[i] += (S_k[q_i] - NTT_i(delta)) P^-1 */
        const uint64_t tt = c->t, pinv_t = invmod(P % tt, tt);
        for (int k = 0; k < 2; k++)
        {
            uint64_t *r = S + ((size_t)k * (K + 1) + K) * n;
            ntt_inv_generic(r, n, c->logn, P, c->psi[L - 1]);
            for (int i = 0; i < K; i++)
            {
                uint64_t qi = c->q[i], pinv = invmod(P % qi, qi);
                uint64_t *sp = S + ((size_t)k * (K + 1) + i) * n;
                uint64_t *cp = ct + ((size_t)k * K + i) * n;
                for (uint64_t j = 0; j < n; j++)
                {
                    uint64_t kk = mulmod(submod(0, r[j] % tt, tt), pinv_t, tt);
                    tmp[j] = addmod(mulmod(kk % qi, P % qi, qi), r[j] % qi, qi);
                }
                so_ntt_forward(c, i, tmp);
                for (uint64_t j = 0; j < n; j++)
                    cp[j] = addmod(cp[j], mulmod(submod(sp[j], tmp[j], qi), pinv, qi), qi);
            }
        }
        free(t);
        free(S);
        free(tmp);
        return;
    }
    for (int k = 0; k < 2; k++)
    {
        uint64_t *r = S + ((size_t)k * (K + 1) + K) * n;
        ntt_inv_generic(r, n, c->logn, P, c->psi[L - 1]); /* :2810 */
        for (uint64_t j = 0; j < n; j++)
            r[j] = addmod(r[j], half, P); /* :2813-2817 */
        for (int i = 0; i < K; i++)
        {
            uint64_t qi = c->q[i], pinv = invmod(P % qi, qi);
            uint64_t *sp = S + ((size_t)k * (K + 1) + i) * n;
            uint64_t *cp = ct + ((size_t)k * K + i) * n;
            for (uint64_t j = 0; j < n; j++)
                tmp[j] = submod(r[j] % qi, half % qi, qi); /* :2819-2832 */
            if (c->scheme == 2)
                so_ntt_forward(c, i, tmp); /* CKKS: everything in NTT form :2836 */
            else
                so_ntt_inverse(c, i, sp); /* BFV: bring S back to coefficient form :2848 */
            for (uint64_t j = 0; j < n; j++)
                cp[j] = addmod(cp[j], mulmod(submod(sp[j], tmp[j], qi), pinv, qi), qi); /* :2852-2863 */
        }
    }
    free(t);
    free(S);
    free(tmp);
}

/* ---------------------------------------------------------------- BEHZ (rns.cpp:418-463, 903-1131) */
/* FastBConv(x; base A -> p) = sum_a [x_a (A/a)^-1 mod a] ((A/a) mod p) mod p */
static uint64_t punct_mod(const uint64_t *base, int cnt, int skip, uint64_t m)
{
    uint64_t r = 1 % m;
    for (int i = 0; i < cnt; i++)
        if (i != skip)
            r = mulmod(r, base[i] % m, m);
    return r;
}
static uint64_t fastbconv(const uint64_t *x, const uint64_t *base, int cnt, uint64_t p)
{
    uint64_t s = 0;
    for (int i = 0; i < cnt; i++)
    {
        uint64_t y = mulmod(x[i] % base[i], invmod(punct_mod(base, cnt, i, base[i]), base[i]), base[i]);
        s = addmod(s, mulmod(y % p, punct_mod(base, cnt, i, p), p), p);
    }
    return s;
}
/* per-coefficient precomputation hoisted: build matrices once per call */
typedef struct
{
    int K, nB, nBsk;
    const uint64_t *q, *bsk;
    uint64_t mt, msk;
    uint64_t inv_punct_q[SO_MAX_PRIMES], inv_punct_b[SO_MAX_PRIMES + 1];
    uint64_t q_to_bsk[(SO_MAX_PRIMES + 2) * SO_MAX_PRIMES], q_to_mt[SO_MAX_PRIMES];
    uint64_t b_to_q[SO_MAX_PRIMES * (SO_MAX_PRIMES + 1)], b_to_msk[SO_MAX_PRIMES + 1];
    uint64_t prod_q_mod_bsk[SO_MAX_PRIMES + 2], inv_prod_q_mod_bsk[SO_MAX_PRIMES + 2], inv_mt_mod_bsk[SO_MAX_PRIMES + 2];
    uint64_t prod_b_mod_q[SO_MAX_PRIMES], inv_prod_b_mod_msk, neg_inv_prod_q_mod_mt;
} behz_tab;
static behz_tab *behz_build(const so_ctx *c, int K)
{
    behz_tab *b = (behz_tab *)calloc(1, sizeof(behz_tab));
    const so_behz *z = &c->behz[K];
    b->K = K;
    b->nB = z->nB;
    b->nBsk = z->nBsk;
    b->q = c->q;
    b->bsk = z->bsk;
    b->mt = (uint64_t)1 << 32; /* m_tilde, rns.cpp:635 */
    b->msk = z->bsk[z->nB];
    for (int i = 0; i < K; i++)
    {
        b->inv_punct_q[i] = invmod(punct_mod(c->q, K, i, c->q[i]), c->q[i]);
        b->q_to_mt[i] = punct_mod(c->q, K, i, b->mt);
        b->prod_b_mod_q[i] = punct_mod(z->bsk, z->nB, -1, c->q[i]);
        for (int j = 0; j < z->nB; j++)
            b->b_to_q[i * z->nB + j] = punct_mod(z->bsk, z->nB, j, c->q[i]);
    }
    for (int j = 0; j < z->nBsk; j++)
    {
        uint64_t p = z->bsk[j];
        for (int i = 0; i < K; i++)
            b->q_to_bsk[j * K + i] = punct_mod(c->q, K, i, p);
        b->prod_q_mod_bsk[j] = punct_mod(c->q, K, -1, p);
        b->inv_prod_q_mod_bsk[j] = invmod(b->prod_q_mod_bsk[j], p);
        b->inv_mt_mod_bsk[j] = invmod(b->mt % p, p);
    }
    for (int j = 0; j < z->nB; j++)
    {
        b->inv_punct_b[j] = invmod(punct_mod(z->bsk, z->nB, j, z->bsk[j]), z->bsk[j]);
        b->b_to_msk[j] = punct_mod(z->bsk, z->nB, j, b->msk);
    }
    b->inv_prod_b_mod_msk = invmod(punct_mod(z->bsk, z->nB, -1, b->msk), b->msk);
    b->neg_inv_prod_q_mod_mt = (b->mt - invmod(punct_mod(c->q, K, -1, b->mt), b->mt)) % b->mt;
    return b;
}
/* stage 0: fastbconv_m_tilde, x[K] -> out[nBsk+1] */
static void st_fastbconv_m_tilde(const behz_tab *b, const uint64_t *x, uint64_t *out)
{
    uint64_t y[SO_MAX_PRIMES];
    for (int i = 0; i < b->K; i++)
        y[i] = mulmod(mulmod(x[i], b->mt % b->q[i], b->q[i]), b->inv_punct_q[i], b->q[i]);
    for (int j = 0; j < b->nBsk; j++)
    {
        uint64_t s = 0, p = b->bsk[j];
        for (int i = 0; i < b->K; i++)
            s = addmod(s, mulmod(y[i] % p, b->q_to_bsk[j * b->K + i], p), p);
        out[j] = s;
    }
    uint64_t s = 0;
    for (int i = 0; i < b->K; i++)
        s = (s + mulmod(y[i] % b->mt, b->q_to_mt[i], b->mt)) % b->mt;
    out[b->nBsk] = s;
}
/* stage 1: sm_mrq, in[nBsk+1] -> out[nBsk] */
static void st_sm_mrq(const behz_tab *b, const uint64_t *in, uint64_t *out)
{
    uint64_t r = mulmod(in[b->nBsk], b->neg_inv_prod_q_mod_mt, b->mt);
    for (int j = 0; j < b->nBsk; j++)
    {
        uint64_t p = b->bsk[j], rp = r;
        if (rp >= (b->mt >> 1))
            rp += p - b->mt; /* centered, rns.cpp:1027-1031 */
        uint64_t v = addmod(mulmod(rp % p, b->prod_q_mod_bsk[j], p), in[j], p);
        out[j] = mulmod(v, b->inv_mt_mod_bsk[j], p);
    }
}
/* stage 2: fast_floor, in[K + nBsk] -> out[nBsk] */
static void st_fast_floor(const behz_tab *b, const uint64_t *in, uint64_t *out)
{
    uint64_t y[SO_MAX_PRIMES];
    for (int i = 0; i < b->K; i++)
        y[i] = mulmod(in[i] % b->q[i], b->inv_punct_q[i], b->q[i]);
    for (int j = 0; j < b->nBsk; j++)
    {
        uint64_t s = 0, p = b->bsk[j];
        for (int i = 0; i < b->K; i++)
            s = addmod(s, mulmod(y[i] % p, b->q_to_bsk[j * b->K + i], p), p);
        out[j] = mulmod(submod(in[b->K + j], s, p), b->inv_prod_q_mod_bsk[j], p);
    }
}
/* stage 3: fastbconv_sk, in[nBsk] -> out[K] */
static void st_fastbconv_sk(const behz_tab *b, const uint64_t *in, uint64_t *out)
{
    uint64_t w[SO_MAX_PRIMES + 1], msk = b->msk;
    for (int j = 0; j < b->nB; j++)
        w[j] = mulmod(in[j] % b->bsk[j], b->inv_punct_b[j], b->bsk[j]);
    uint64_t s = 0;
    for (int j = 0; j < b->nB; j++)
        s = addmod(s, mulmod(w[j] % msk, b->b_to_msk[j], msk), msk);
    uint64_t alpha = mulmod(submod(s, in[b->nB], msk), b->inv_prod_b_mod_msk, msk);
    for (int i = 0; i < b->K; i++)
    {
        uint64_t qi = b->q[i], g = 0;
        for (int j = 0; j < b->nB; j++)
            g = addmod(g, mulmod(w[j] % qi, b->b_to_q[i * b->nB + j], qi), qi);
        if (alpha > (msk >> 1)) /* rns.cpp:962-975 */
            out[i] = addmod(g, mulmod((msk - alpha) % qi, b->prod_b_mod_q[i], qi), qi);
        else
            out[i] = submod(g, mulmod(alpha % qi, b->prod_b_mod_q[i], qi), qi);
    }
}
int so_rns_stage(const so_ctx *c, int K, int which, const uint64_t *in, uint64_t *out)
{
    if (c->scheme != 1)
        return -1;
    behz_tab *b = behz_build(c, K);
    uint64_t n = c->n;
    int ic[] = { K, b->nBsk + 1, K + b->nBsk, b->nBsk }, oc[] = { b->nBsk + 1, b->nBsk, b->nBsk, K };
    uint64_t vi[2 * SO_MAX_PRIMES + 4], vo[SO_MAX_PRIMES + 4];
    for (uint64_t j = 0; j < n; j++)
    {
        for (int i = 0; i < ic[which]; i++)
            vi[i] = in[(size_t)i * n + j];
        if (which == 0)
            st_fastbconv_m_tilde(b, vi, vo);
        else if (which == 1)
            st_sm_mrq(b, vi, vo);
        else if (which == 2)
            st_fast_floor(b, vi, vo);
        else
            st_fastbconv_sk(b, vi, vo);
        for (int i = 0; i < oc[which]; i++)
            out[(size_t)i * n + j] = vo[i];
    }
    free(b);
    return 0;
}

/* bfv_multiply (evaluator.cpp:395-567), steps (1)-(8) */
int so_bfv_multiply(const so_ctx *c, int K, const uint64_t *x, int sx, const uint64_t *y, int sy, uint64_t *out)
{
    if (c->scheme != 1)
        return -1;
    behz_tab *b = behz_build(c, K);
    const so_behz *z = &c->behz[K];
    const uint64_t n = c->n;
    const int nBsk = b->nBsk, dest = sx + sy - 1;
    uint64_t *xq = (uint64_t *)malloc((size_t)sx * K * n * 8), *xb = (uint64_t *)malloc((size_t)sx * nBsk * n * 8);
    uint64_t *yq = (uint64_t *)malloc((size_t)sy * K * n * 8), *yb = (uint64_t *)malloc((size_t)sy * nBsk * n * 8);
    uint64_t *dq = (uint64_t *)malloc((size_t)dest * K * n * 8), *db = (uint64_t *)malloc((size_t)dest * nBsk * n * 8);
    uint64_t vi[2 * SO_MAX_PRIMES + 4], v1[SO_MAX_PRIMES + 4], v2[SO_MAX_PRIMES + 4];
    for (int side = 0; side < 2; side++)
    {
        const uint64_t *in = side ? y : x;
        int s = side ? sy : sx;
        uint64_t *oq = side ? yq : xq, *ob = side ? yb : xb;
        memcpy(oq, in, (size_t)s * K * n * 8);
        for (int p = 0; p < s; p++)
        {
            for (uint64_t j = 0; j < n; j++)
            {
                for (int i = 0; i < K; i++)
                    vi[i] = in[((size_t)p * K + i) * n + j];
                st_fastbconv_m_tilde(b, vi, v1); /* (1) */
                st_sm_mrq(b, v1, v2);            /* (2) */
                for (int i = 0; i < nBsk; i++)
                    ob[((size_t)p * nBsk + i) * n + j] = v2[i];
            }
            for (int i = 0; i < K; i++)
                so_ntt_forward(c, i, oq + ((size_t)p * K + i) * n); /* (3) */
            for (int i = 0; i < nBsk; i++)
                ntt_fwd_generic(ob + ((size_t)p * nBsk + i) * n, n, c->logn, z->bsk[i], z->bsk_psi[i]);
        }
    }
    tensor(c->q, K, n, xq, sx, yq, sy, dq);       /* (4) */
    tensor(z->bsk, nBsk, n, xb, sx, yb, sy, db);
    for (int p = 0; p < dest; p++)
    {
        for (int i = 0; i < K; i++)
            so_ntt_inverse(c, i, dq + ((size_t)p * K + i) * n); /* (5) */
        for (int i = 0; i < nBsk; i++)
            ntt_inv_generic(db + ((size_t)p * nBsk + i) * n, n, c->logn, z->bsk[i], z->bsk_psi[i]);
        for (uint64_t j = 0; j < n; j++)
        {
            for (int i = 0; i < K; i++)
                vi[i] = mulmod(dq[((size_t)p * K + i) * n + j], c->t % c->q[i], c->q[i]); /* (6) */
            for (int i = 0; i < nBsk; i++)
                vi[K + i] = mulmod(db[((size_t)p * nBsk + i) * n + j], c->t % z->bsk[i], z->bsk[i]);
            st_fast_floor(b, vi, v1);   /* (7) */
            st_fastbconv_sk(b, v1, v2); /* (8) */
            for (int i = 0; i < K; i++)
                out[((size_t)p * K + i) * n + j] = v2[i];
        }
    }
    free(xq);
    free(xb);
    free(yq);
    free(yb);
    free(dq);
    free(db);
    free(b);
    (void)fastbconv;
    return 0;
}

/* ---------------------------------------------------------------- CPU baseline ("port") */
double so_time_ckks_pipeline(const so_ctx *c, int K, const uint64_t *a, const uint64_t *b, const uint64_t *rlk, int reps,
                             uint64_t *out)
{
    const uint64_t n = c->n;
    uint64_t *prod = (uint64_t *)malloc((size_t)3 * K * n * 8);
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int r = 0; r < reps; r++)
    {
        so_ckks_multiply(c, K, a, 2, b, 2, prod);
        so_switch_key(c, K, prod, prod + (size_t)2 * K * n, rlk); /* relinearize: target = c2 */
        so_rescale(c, K, prod, 2, out);
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    free(prod);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
