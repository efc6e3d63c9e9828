"""Shared helpers for the parity tests: build the device side (through the C ABI via seal_amd) for
the same parameters as an Oracle, upload the oracle's keys, move ciphertext slabs in and out."""
import numpy as np

import seal_amd as S


class DeviceSide:
    def __init__(self, scheme, n, primes, plain_modulus=0):
        self.scheme, self.n, self.primes, self.t = scheme, n, list(primes), plain_modulus
        p = S.EncryptionParameters(scheme)
        p.set_poly_modulus_degree(n)
        p.set_coeff_modulus(primes)
        if scheme != "ckks":
            p.set_plain_modulus(plain_modulus)
        self.parms = p
        self.ctx = S.SEALContext(p, True, 0)
        self.ev = S.Evaluator(self.ctx)
        self.L = len(primes)
        self.rlk = None
        self.glk = None

    def chain_index_for_K(self, K):
        key_ci = self.ctx.chain_index(self.ctx.key_parms_id())
        return key_ci - (self.L - K)

    def parms_id_for_K(self, K):
        return self.ctx.parms_id_at(self.chain_index_for_K(K))

    def upload_keys(self, oracle):
        self.rlk = S.RelinKeys(self.ctx)
        self.rlk.set_key(0, oracle.relin_key())
        self.glk = S.GaloisKeys(self.ctx)
        for elt in oracle.galois_elts:
            self.glk.set_key(S.GaloisKeys.get_index(elt), oracle.galois_key(elt))

    def ct(self, slabs, scale=None, is_ntt=None):
        """slabs: one [size][K][N] array or a list of them (a batch)."""
        if isinstance(slabs, np.ndarray) and slabs.ndim == 3:
            slabs = [slabs]
        arr = np.stack(slabs, axis=1)  # [size][batch][K][N]
        K = arr.shape[2]
        if is_ntt is None:
            is_ntt = self.scheme == "ckks"
        if scale is None:
            scale = 2.0 ** 20 if self.scheme == "ckks" else 1.0
        return S.Ciphertext.from_numpy(self.ctx, arr, self.parms_id_for_K(K), is_ntt, scale)

    @staticmethod
    def out(ct):
        """-> list of [size][K][N] arrays, one per batch item"""
        a = ct.to_numpy()
        return [np.ascontiguousarray(a[:, b]) for b in range(a.shape[1])]
