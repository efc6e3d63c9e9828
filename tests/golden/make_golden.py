"""Generates tests/golden/*.npz from the REAL reference (oracle/_ref/libsealref.so = Microsoft SEAL
4.4.3 compiled from /root/reference by oracle/Makefile).  Run in the build container:

    make -C oracle ref && python tests/golden/make_golden.py

The vectors are small (N = 64 / 32) so they travel with the repo; they pin both the plain-C oracle
(tests/test_oracle.py) and the HIP path (tests/test_gpu_parity.py::test_golden_*) on machines where
/root/reference and oracle/_ref do not exist.  Inputs are seeded; keys are the reference's own
(KeyGenerator with Blake2xb seed 0x5EA1), stored verbatim.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import sealref as R  # noqa: E402


def rand_ct(rng, primes, K, n, size=2):
    return np.stack([np.stack([rng.integers(0, primes[i], n, dtype=np.uint64) for i in range(K)]) for _ in range(size)])


def ckks():
    n, bits = 64, [40, 30, 30, 40]
    primes = R.coeff_modulus_create(n, bits)
    ref = R.RefContext("ckks", n, primes)
    ref.keygen_relin()
    elt = ref.galois_elt_from_step(1)
    ref.keygen_galois_elts([elt])
    K = len(primes) - 1
    fc = ref.first_chain_index
    rng = np.random.default_rng(0x5EA1)
    a, b = rand_ct(rng, primes, K, n), rand_ct(rng, primes, K, n)
    out = dict(n=n, primes=np.array(primes, dtype=np.uint64), bits=np.array(bits), a=a, b=b, galois_elt=elt,
               roots=np.array([ref.ntt_tables(ref.key_chain_index, i, False)[0] for i in range(len(primes))], dtype=np.uint64),
               relin_key=ref.key("relin", 0), galois_key=ref.key("galois", (elt - 1) >> 1))
    out["ntt_fwd_a0"] = ref.ntt(fc, 0, a[0], "fwd")
    out["ntt_inv_a0"] = ref.ntt(fc, 0, a[0], "inv")
    x = ref.ct(fc, a, True, 2.0 ** 10)
    y = ref.ct(fc, b, True, 2.0 ** 10)
    ref.multiply_inplace(x, y)
    out["multiply"] = x.data()
    ref.relinearize_inplace(x)
    out["relinearize"] = x.data()
    z = ref.ct(fc, x.data(), True, float(primes[K - 1]) * 2.0 ** 10)
    ref.rescale_to_next_inplace(z)
    out["rescale"] = z.data()
    out["rescale_scale"] = z.info()["scale"]
    ref.rotate_vector_inplace(z, 1)
    out["rotate1"] = z.data()
    ref.mod_switch_to_next_inplace(z)
    out["mod_switch"] = z.data()
    np.savez_compressed(os.path.join(HERE, "ckks_n64.npz"), **out)


def bfv():
    n = 32
    primes = R.coeff_modulus_create(n, [30, 30, 30, 30])
    t = R.plain_modulus_batching(n, 12)
    ref = R.RefContext("bfv", n, primes, t)
    ref.keygen_relin()
    elt = ref.galois_elt_from_step(1)
    ref.keygen_galois_elts([elt, 2 * n - 1])
    K = len(primes) - 1
    fc = ref.first_chain_index
    rng = np.random.default_rng(0x5EA1 + 1)
    a, b = rand_ct(rng, primes, K, n), rand_ct(rng, primes, K, n)
    bsk, m_tilde, gamma = ref.behz_bases(fc)
    out = dict(n=n, primes=np.array(primes, dtype=np.uint64), t=t, a=a, b=b, galois_elt=elt,
               bsk=np.array(bsk, dtype=np.uint64), relin_key=ref.key("relin", 0),
               galois_key=ref.key("galois", (elt - 1) >> 1), conj_key=ref.key("galois", (2 * n - 2) >> 1))
    nBsk = len(bsk)
    e0 = ref.rns_stage(fc, "fastbconv_m_tilde", a[0], nBsk + 1)
    e1 = ref.rns_stage(fc, "sm_mrq", e0, nBsk)
    e2 = ref.rns_stage(fc, "fast_floor", np.concatenate([a[0], e1]), nBsk)
    e3 = ref.rns_stage(fc, "fastbconv_sk", e2, K)
    out.update(fastbconv_m_tilde=e0, sm_mrq=e1, fast_floor=e2, fastbconv_sk=e3)
    x, y = ref.ct(fc, a, False), ref.ct(fc, b, False)
    ref.multiply_inplace(x, y)
    out["multiply"] = x.data()
    ref.relinearize_inplace(x)
    out["relinearize"] = x.data()
    ref.rotate_rows_inplace(x, 1)
    out["rotate_rows1"] = x.data()
    ref.rotate_columns_inplace(x)
    out["rotate_columns"] = x.data()
    ref.mod_switch_to_next_inplace(x)
    out["mod_switch"] = x.data()
    np.savez_compressed(os.path.join(HERE, "bfv_n32.npz"), **out)


if __name__ == "__main__":
    ckks()
    bfv()
    print("golden vectors written to", HERE)
