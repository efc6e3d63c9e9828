"""Generates tests/golden/engine_digests.json from the REAL reference (oracle/_ref/libsealref.so):
SHA-256 digests of the reference's outputs at the sizes the two-pass engine / fused key switch run at
(N >= 8192), for seeded synthetic inputs AND seeded synthetic keys (ref_key_set), so the fixture is a
few hundred bytes and everything else is regenerated from the seeds by the test
(tests/test_gpu_parity.py::test_golden_engine_digests, tests/test_emu_parity.py).  Run in the build
container:   make -C oracle ref && python tests/golden/make_golden_engine.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import sealref as R  # noqa: E402

CASES = [
    dict(name="ckks_n8192_fp_and_int", n=8192, bits=[50, 40, 60, 50], seed=101),
    dict(name="ckks_n16384_50bit", n=16384, bits=[60, 50, 50, 60], seed=102),
]


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.uint64).tobytes()).hexdigest()


def synth(case):
    """seeded inputs and keys, identical here and in the tests"""
    n, bits = case["n"], case["bits"]
    primes = R.coeff_modulus_create(n, bits)
    L, K = len(primes), len(primes) - 1
    rng = np.random.default_rng(case["seed"])
    ct = lambda: np.stack([np.stack([rng.integers(0, primes[i], n, dtype=np.uint64) for i in range(K)]) for _ in range(2)])
    a, b = ct(), ct()
    key = lambda: np.stack([np.stack([np.stack([rng.integers(0, primes[i], n, dtype=np.uint64) for i in range(L)])
                                      for _ in range(2)]) for _ in range(K)])
    return primes, a, b, key(), key()


def main():
    out = {}
    for case in CASES:
        primes, a, b, rlk, glk = synth(case)
        n, K = case["n"], len(primes) - 1
        ref = R.RefContext("ckks", n, primes)
        ref.keygen_relin()
        elt = ref.galois_elt_from_step(1)
        ref.keygen_galois_elts([elt])
        ref.set_key("relin", 0, rlk)
        ref.set_key("galois", (elt - 1) >> 1, glk)
        fc = ref.first_chain_index
        d = {}
        d["ntt_fwd_a0"] = digest(ref.ntt(fc, 0, a[0], "fwd"))
        d["ntt_inv_a0"] = digest(ref.ntt(fc, 0, a[0], "inv"))
        x, y = ref.ct(fc, a, True, 2.0 ** 10), ref.ct(fc, b, True, 2.0 ** 10)
        ref.multiply_inplace(x, y)
        d["multiply"] = digest(x.data())
        ref.relinearize_inplace(x)
        d["relinearize"] = digest(x.data())
        z = ref.ct(fc, x.data(), True, float(primes[K - 1]) * 2.0 ** 10)
        ref.rescale_to_next_inplace(z)
        d["rescale"] = digest(z.data())
        ref.rotate_vector_inplace(z, 1)
        d["rotate1"] = digest(z.data())
        out[case["name"]] = dict(n=n, bits=case["bits"], seed=case["seed"], galois_elt=int(elt), digests=d)
    with open(os.path.join(HERE, "engine_digests.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(json.dumps(out, indent=1)[:400])


if __name__ == "__main__":
    main()
