"""ctypes wrapper over oracle/libsealoracle.so — the plain-C restatement of the path
(oracle/seal_oracle.c, the "port" oracle).  TEST INFRASTRUCTURE ONLY.  Built on demand with gcc."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(_HERE, "..", "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "libsealoracle.so")
_lib = None


def build():
    src = os.path.join(ORACLE_DIR, "seal_oracle.c")
    if not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-o", LIB_PATH, src], cwd=ORACLE_DIR)


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB_PATH)
        _lib.so_ctx_create.restype = C.c_void_p
        _lib.so_ntt_root.restype = C.c_uint64
        _lib.so_galois_elt_from_step.restype = C.c_uint32
        _lib.so_time_ckks_pipeline.restype = C.c_double
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def coeff_modulus_create(n, bit_sizes):
    bits = (C.c_int * len(bit_sizes))(*bit_sizes)
    out = np.zeros(len(bit_sizes), dtype=np.uint64)
    assert lib().so_coeff_modulus_create(C.c_uint64(n), bits, C.c_int(len(bit_sizes)), _p(out)) == 0
    return [int(x) for x in out]


def plain_modulus_batching(n, bits):
    return coeff_modulus_create(n, [bits])[0]


class PortContext:
    def __init__(self, scheme, n, primes, plain_modulus=0):
        self.scheme, self.n, self.primes, self.t = scheme, n, list(primes), plain_modulus
        self.L = len(primes)
        arr = np.array(primes, dtype=np.uint64)
        self.h = C.c_void_p(lib().so_ctx_create(C.c_int({"bfv": 1, "ckks": 2, "bgv": 3}[scheme]), C.c_uint64(n), _p(arr),
                                                C.c_int(len(primes)), C.c_uint64(plain_modulus)))
        assert self.h.value

    def __del__(self):
        if getattr(self, "h", None) and self.h.value:
            lib().so_ctx_destroy(self.h)
            self.h = None

    def ntt_root(self, i):
        return int(lib().so_ntt_root(self.h, C.c_int(i)))

    def base_bsk(self, K):
        out = np.zeros(80, dtype=np.uint64)
        cnt = lib().so_base_bsk(self.h, C.c_int(K), _p(out))
        return [int(x) for x in out[:cnt]]

    def ntt(self, first, data, mode):
        d = np.ascontiguousarray(data, dtype=np.uint64).copy()
        for i in range(d.shape[0]):
            if mode == "fwd":
                lib().so_ntt_forward(self.h, C.c_int(first + i), _p(d[i]))
            else:
                lib().so_ntt_inverse(self.h, C.c_int(first + i), _p(d[i]))
        return d

    def ntt_naive(self, prime_index, a):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        out = np.zeros_like(a)
        lib().so_ntt_forward_naive(self.h, C.c_int(prime_index), _p(a), _p(out))
        return out

    def dyadic(self, prime_index, a, b):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        b = np.ascontiguousarray(b, dtype=np.uint64)
        r = np.zeros_like(a)
        lib().so_dyadic(self.h, C.c_int(prime_index), _p(a), _p(b), _p(r))
        return r

    def multiply(self, x, y):
        """x: [sx][K][N], y: [sy][K][N] -> [sx+sy-1][K][N] (scheme-appropriate)."""
        x = np.ascontiguousarray(x, dtype=np.uint64)
        y = np.ascontiguousarray(y, dtype=np.uint64)
        K = x.shape[1]
        out = np.zeros((x.shape[0] + y.shape[0] - 1, K, self.n), dtype=np.uint64)
        if self.scheme != "bfv":   # CKKS and BGV share the NTT-domain tensor product (evaluator.cpp:569-841)
            lib().so_ckks_multiply(self.h, C.c_int(K), _p(x), C.c_int(x.shape[0]), _p(y), C.c_int(y.shape[0]), _p(out))
        else:
            assert lib().so_bfv_multiply(self.h, C.c_int(K), _p(x), C.c_int(x.shape[0]), _p(y), C.c_int(y.shape[0]), _p(out)) == 0
        return out

    def switch_key(self, ct, target, key):
        """ct [2][K][N] (copied), target [K][N], key [digits][2][L][N] -> new ct."""
        ct = np.ascontiguousarray(ct, dtype=np.uint64).copy()
        target = np.ascontiguousarray(target, dtype=np.uint64)
        key = np.ascontiguousarray(key, dtype=np.uint64)
        lib().so_switch_key(self.h, C.c_int(ct.shape[1]), _p(ct), _p(target), _p(key))
        return ct

    def relinearize(self, ct3, key):
        return self.switch_key(ct3[:2], ct3[2], key)

    def apply_galois_poly(self, poly, ntt_form, elt):
        poly = np.ascontiguousarray(poly, dtype=np.uint64)
        out = np.zeros_like(poly)
        lib().so_apply_galois(self.h, C.c_int(poly.shape[0]), C.c_int(int(ntt_form)), C.c_uint32(elt), _p(poly), _p(out))
        return out

    def apply_galois(self, ct2, elt, key):
        """Evaluator::apply_galois_inplace (evaluator.cpp:2384-2502)."""
        ntt_form = self.scheme != "bfv"
        c0 = self.apply_galois_poly(ct2[0], ntt_form, elt)
        c1 = self.apply_galois_poly(ct2[1], ntt_form, elt)
        return self.switch_key(np.stack([c0, np.zeros_like(c0)]), c1, key)

    def galois_elt_from_step(self, step):
        return int(lib().so_galois_elt_from_step(self.h, C.c_int(step)))

    def rescale(self, ct):
        ct = np.ascontiguousarray(ct, dtype=np.uint64)
        out = np.zeros((ct.shape[0], ct.shape[1] - 1, self.n), dtype=np.uint64)
        lib().so_rescale(self.h, C.c_int(ct.shape[1]), _p(ct), C.c_int(ct.shape[0]), _p(out))
        return out

    def bfv_mod_switch(self, ct):
        ct = np.ascontiguousarray(ct, dtype=np.uint64)
        out = np.zeros((ct.shape[0], ct.shape[1] - 1, self.n), dtype=np.uint64)
        lib().so_bfv_mod_switch(self.h, C.c_int(ct.shape[1]), _p(ct), C.c_int(ct.shape[0]), _p(out))
        return out

    def bgv_mod_switch(self, ct, correction_factor=1):
        """-> (ct at the next level, its correction factor)"""
        ct = np.ascontiguousarray(ct, dtype=np.uint64)
        out = np.zeros((ct.shape[0], ct.shape[1] - 1, self.n), dtype=np.uint64)
        lib().so_bgv_mod_switch(self.h, C.c_int(ct.shape[1]), _p(ct), C.c_int(ct.shape[0]), _p(out))
        lib().so_bgv_mod_switch_correction.restype = C.c_uint64
        cf = lib().so_bgv_mod_switch_correction(self.h, C.c_int(ct.shape[1]), C.c_uint64(correction_factor))
        return out, int(cf)

    def plain_lift(self, K, m, scale_by=1):
        m = np.ascontiguousarray(m, dtype=np.uint64)
        out = np.zeros((K, self.n), dtype=np.uint64)
        lib().so_plain_lift(self.h, C.c_int(K), _p(m), C.c_uint64(m.size), C.c_uint64(scale_by), _p(out))
        return out

    def plain_to_ntt(self, K, m, scale_by=1):
        """transform_to_ntt_inplace(Plaintext, parms_id of the level with K primes) (evaluator.cpp:2196-2287)"""
        return self.ntt(0, self.plain_lift(K, m, scale_by), "fwd")

    def addsub_plain(self, ct, m, sub=False, correction_factor=1):
        """add_plain_inplace / sub_plain_inplace (evaluator.cpp:1760-1985), coefficient-form plaintext (BFV, BGV)"""
        ct = np.ascontiguousarray(ct, dtype=np.uint64).copy()
        m = np.ascontiguousarray(m, dtype=np.uint64)
        K = ct.shape[1]
        if self.scheme == "bfv":
            lib().so_bfv_addsub_plain(self.h, C.c_int(K), _p(ct[0]), _p(m), C.c_uint64(m.size), C.c_int(int(sub)))
            return ct
        p = self.plain_to_ntt(K, m, correction_factor)
        q = np.array(self.primes[:K], dtype=np.uint64)[:, None]
        ct[0] = (ct[0] + (q - p) % q) % q if sub else (ct[0] + p) % q
        return ct

    def multiply_plain(self, ct, m):
        """multiply_plain_inplace with a coefficient-form plaintext, generic (non-monomial) path (evaluator.cpp:2096-2155)"""
        ct = np.ascontiguousarray(ct, dtype=np.uint64)
        K = ct.shape[1]
        p = self.plain_to_ntt(K, m)
        out = np.zeros_like(ct)
        for i in range(ct.shape[0]):
            x = ct[i] if self.scheme != "bfv" else self.ntt(0, ct[i], "fwd")
            y = np.stack([self.dyadic(k, x[k], p[k]) for k in range(K)])
            out[i] = y if self.scheme != "bfv" else self.ntt(0, y, "inv")
        return out

    def drop_last(self, ct):
        return np.ascontiguousarray(ct[:, :-1, :])

    def rns_stage(self, K, which, data, out_comps):
        w = {"fastbconv_m_tilde": 0, "sm_mrq": 1, "fast_floor": 2, "fastbconv_sk": 3}[which]
        d = np.ascontiguousarray(data, dtype=np.uint64)
        out = np.zeros((out_comps, self.n), dtype=np.uint64)
        assert lib().so_rns_stage(self.h, C.c_int(K), C.c_int(w), _p(d), _p(out)) == 0
        return out

    def time_ckks_pipeline(self, a, b, rlk, reps):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        b = np.ascontiguousarray(b, dtype=np.uint64)
        rlk = np.ascontiguousarray(rlk, dtype=np.uint64)
        K = a.shape[1]
        out = np.zeros((2, K - 1, self.n), dtype=np.uint64)
        s = lib().so_time_ckks_pipeline(self.h, C.c_int(K), _p(a), _p(b), _p(rlk), C.c_int(reps), _p(out))
        return s, out
