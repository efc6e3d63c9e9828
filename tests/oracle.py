"""One oracle facade for the parity tests.  TEST INFRASTRUCTURE ONLY.

kind == "reference": the REAL reference (oracle/_ref/libsealref.so, Microsoft SEAL 4.4.3 compiled from
/root/reference) — keys come from seal::KeyGenerator, every call runs seal::Evaluator.
kind == "port": the plain-C restatement (oracle/seal_oracle.c) — keys are uniform random slabs.
Either way the GPU path is fed exactly the same input and key words and must return the same words.

All operations take and return one ciphertext as a uint64 array [size][K][N].
"""
import os

import numpy as np

import sealoracle
import sealref


def kind_available():
    forced = os.environ.get("SEALHIP_ORACLE")
    if forced in ("port", "reference"):
        return forced
    return "reference" if sealref.available() else "port"


def coeff_modulus_create(n, bits):
    if kind_available() == "reference":
        return sealref.coeff_modulus_create(n, bits)
    return sealoracle.coeff_modulus_create(n, bits)


def plain_modulus_batching(n, bits):
    if kind_available() == "reference":
        return sealref.plain_modulus_batching(n, bits)
    return sealoracle.plain_modulus_batching(n, bits)


def rand_ct(rng, primes, K, n, size=2):
    """uniform residues per RNS component, as BMEnv::randomize_ct_* (native/bench/bench.h:195-270)"""
    return np.stack([np.stack([rng.integers(0, primes[i], n, dtype=np.uint64) for i in range(K)]) for _ in range(size)])


class Oracle:
    def __init__(self, scheme, n, primes, plain_modulus=0, galois_elts=(), kind=None, seed=0x5EA1):
        self.kind = kind or kind_available()
        self.scheme, self.n, self.primes, self.t = scheme, n, list(primes), plain_modulus
        self.L = len(primes)
        self.K_first = self.L - 1 if self.L > 1 else 1
        self.galois_elts = list(galois_elts)
        if self.kind == "reference":
            self.ref = sealref.RefContext(scheme, n, primes, plain_modulus, seed)
            if self.L > 1:
                self.ref.keygen_relin()
                if self.galois_elts:
                    self.ref.keygen_galois_elts(self.galois_elts)
        else:
            self.port = sealoracle.PortContext(scheme, n, primes, plain_modulus)
            self._rng = np.random.default_rng(seed)
            self._keys = {}

    # ---- constants
    def ntt_root(self, i):
        if self.kind == "reference":
            return self.ref.ntt_tables(self.ref.key_chain_index, i, False)[0]
        return self.port.ntt_root(i)

    def base_bsk(self, K):
        if self.kind == "reference":
            return self.ref.behz_bases(self._ci(K))[0]
        return self.port.base_bsk(K)

    def galois_elt_from_step(self, step):
        if self.kind == "reference":
            return self.ref.galois_elt_from_step(step)
        return self.port.galois_elt_from_step(step)

    # ---- keys: uint64 [digits][2][L][N]
    def relin_key(self):
        if self.kind == "reference":
            return self.ref.key("relin", 0)
        return self._rand_key(("relin", 0))

    def galois_key(self, elt):
        if self.kind == "reference":
            return self.ref.key("galois", (elt - 1) >> 1)
        return self._rand_key(("galois", elt))

    def _rand_key(self, name):
        if name not in self._keys:
            K = self.L - 1
            self._keys[name] = np.stack([np.stack([np.stack([
                self._rng.integers(0, self.primes[i], self.n, dtype=np.uint64) for i in range(self.L)])
                for _ in range(2)]) for _ in range(K)])
        return self._keys[name]

    # ---- helpers
    def _ci(self, K):
        # chain_index of the level with K data primes: key level (K == L) has the highest index
        return self.ref.key_chain_index - (self.L - K)

    def _ct(self, data, scale=None, cf=1):
        K = data.shape[1]
        is_ntt = self.scheme in ("ckks", "bgv")
        if scale is None:
            scale = 2.0 ** 20 if self.scheme == "ckks" else 1.0
        return self.ref.ct(self._ci(K), data, is_ntt, scale, cf)

    def run(self, op, operands, *args):
        """Generic reference call (reference kind only): operands = [(data, correction_factor), ...];
        op = name of a RefContext in-place method; returns (data, info dict) of the first operand."""
        assert self.kind == "reference"
        cts = [self._ct(d, None, cf) for d, cf in operands]
        getattr(self.ref, op)(*cts, *args)
        return cts[0].data(), cts[0].info()

    # ---- L1
    def ntt(self, first, data, mode):
        """data [count][N] components of primes first..first+count-1 (key-level indexing)."""
        if self.kind == "reference":
            return self.ref.ntt(self.ref.key_chain_index, first, data, mode)
        return self.port.ntt(first, data, mode)

    def dyadic(self, prime_index, a, b):
        if self.kind == "reference":
            return self.ref.dyadic_product(self.ref.key_chain_index, prime_index, a, b)
        return self.port.dyadic(prime_index, a, b)

    def apply_galois_poly(self, poly, ntt_form, elt):
        if self.kind == "reference":
            return self.ref.apply_galois_raw(self._ci(poly.shape[0]), ntt_form, elt, poly)
        return self.port.apply_galois_poly(poly, ntt_form, elt)

    def rns_stage(self, K, which, data, out_comps):
        if self.kind == "reference":
            return self.ref.rns_stage(self._ci(K), which, data, out_comps)
        return self.port.rns_stage(K, which, data, out_comps)

    # ---- Evaluator ops on one ciphertext [size][K][N]
    def multiply(self, x, y, scale=None):
        if self.kind == "reference":
            if scale is None and self.scheme == "ckks":
                scale = 2.0 ** 10  # keep the product below the level's modulus size
            a, b = self._ct(x, scale), self._ct(y, scale)
            self.ref.multiply_inplace(a, b)
            return a.data()
        return self.port.multiply(x, y)

    def relinearize(self, ct3):
        if self.kind == "reference":
            a = self._ct(ct3)
            self.ref.relinearize_inplace(a)
            return a.data()
        return self.port.relinearize(ct3, self.relin_key())

    def rescale(self, ct):
        if self.kind == "reference":
            K = ct.shape[1]
            scale = float(self.primes[K - 1]) * 2.0 ** 10
            a = self._ct(ct, scale)
            self.ref.rescale_to_next_inplace(a)
            return a.data()
        return self.port.rescale(ct)

    def mod_switch_to_next(self, ct):
        if self.kind == "reference":
            a = self._ct(ct)
            self.ref.mod_switch_to_next_inplace(a)
            return a.data()
        return self.port.drop_last(ct) if self.scheme == "ckks" else self.port.bfv_mod_switch(ct)

    def apply_galois(self, ct2, elt):
        if self.kind == "reference":
            a = self._ct(ct2)
            self.ref.apply_galois_inplace(a, elt)
            return a.data()
        return self.port.apply_galois(ct2, elt, self.galois_key(elt))

    def transform(self, ct, to_ntt):
        if self.kind == "reference":
            K = ct.shape[1]
            a = self.ref.ct(self._ci(K), ct, not to_ntt, 2.0 ** 20 if self.scheme == "ckks" else 1.0)
            (self.ref.transform_to_ntt_inplace if to_ntt else self.ref.transform_from_ntt_inplace)(a)
            return a.data()
        return np.stack([self.port.ntt(0, p, "fwd" if to_ntt else "inv") for p in ct])
