"""CPU tests that PIN THE ORACLE before it is trusted as a checker:
  * the reference's own known-answer tests for this path (cited per test),
  * the golden vectors generated from the real reference (tests/golden/make_golden.py),
  * live comparison with the real reference (oracle/_ref) when that library is present.
Nothing here touches the product."""
import os

import numpy as np
import pytest

import sealoracle as O
import sealref as R

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
Q60 = 0xFFFFFFFFFFC0001

needs_ref = pytest.mark.skipif(not R.available(), reason="oracle/_ref/libsealref.so not built on this machine")


def rand_ct(rng, primes, K, n, size=2):
    return np.stack([np.stack([rng.integers(0, primes[i], n, dtype=np.uint64) for i in range(K)]) for _ in range(size)])


# ---- native/tests/seal/util/ntt.cpp:53-72  NTTPrimitiveRootsTest
def test_kat_primitive_roots():
    c = O.PortContext("ckks", 2, [Q60])
    assert c.ntt_root(0) == 288794978602139552
    c = O.PortContext("ckks", 4, [Q60])
    psi = c.ntt_root(0)
    assert pow(psi, 2, Q60) == 288794978602139552  # root_powers[1] = psi^bitrev(1)
    assert psi == 178930308976060547                # root_powers[2] = psi^1
    assert pow(psi, 3, Q60) == 748001537669050592   # root_powers[3] = psi^3


# ---- native/tests/seal/util/ntt.cpp:75-101  NegacyclicNTTTest
def test_kat_negacyclic_ntt_n2():
    c = O.PortContext("ckks", 2, [Q60])
    assert list(c.ntt(0, np.array([[0, 0]], dtype=np.uint64), "fwd")[0]) == [0, 0]
    assert list(c.ntt(0, np.array([[1, 0]], dtype=np.uint64), "fwd")[0]) == [1, 1]
    assert list(c.ntt(0, np.array([[1, 1]], dtype=np.uint64), "fwd")[0]) == [288794978602139553, 864126526004445282]


# ---- native/tests/seal/util/ntt.cpp:103-133  InverseNegacyclicNTTTest (round trip)
@pytest.mark.parametrize("n", [2, 8, 64, 1024])
def test_ntt_round_trip_and_definition(n):
    primes = O.coeff_modulus_create(n, [40, 50])
    c = O.PortContext("ckks", n, primes)
    rng = np.random.default_rng(n)
    x = np.stack([rng.integers(0, p, n, dtype=np.uint64) for p in primes])
    f = c.ntt(0, x, "fwd")
    assert np.array_equal(c.ntt(0, f, "inv"), x)
    if n <= 64:  # O(N^2) definition, SURVEY §8(a')
        for i in range(len(primes)):
            assert np.array_equal(c.ntt_naive(i, x[i]), f[i])


# ---- native/tests/seal/util/galois.cpp:86-120
def test_kat_apply_galois():
    c = O.PortContext("ckks", 8, [17])
    x = np.arange(8, dtype=np.uint64)[None, :]
    assert list(c.apply_galois_poly(x, False, 3)[0]) == [0, 14, 6, 1, 13, 7, 2, 12]
    assert list(c.apply_galois_poly(x, True, 3)[0]) == [4, 5, 7, 6, 1, 0, 2, 3]


# ---- native/tests/seal/util/galois.cpp:28-84 (EltFromStep)
def test_galois_elt_from_step():
    c = O.PortContext("ckks", 8, [17])
    assert c.galois_elt_from_step(0) == 15
    assert c.galois_elt_from_step(1) == 3
    assert c.galois_elt_from_step(-3) == 3          # 3^(N/2 - 3) = 3^1
    assert c.galois_elt_from_step(2) == 9
    assert c.galois_elt_from_step(-1) == 11         # 3^3 mod 16


# ---- native/tests/seal/util/polyarithsmallmod.cpp:545-641 DyadicProductCoeffMod
def test_kat_dyadic():
    c = O.PortContext("ckks", 8, [17])
    a = np.array([1, 1, 1, 1, 2, 2, 2, 2], dtype=np.uint64)
    b = np.array([2, 3, 4, 5, 6, 7, 8, 9], dtype=np.uint64)
    assert list(c.dyadic(0, a, b)) == [2, 3, 4, 5, 12, 14, 16, 1]


# ---- golden vectors from the real reference
def test_golden_ckks():
    g = np.load(os.path.join(GOLDEN, "ckks_n64.npz"))
    n, primes = int(g["n"]), [int(x) for x in g["primes"]]
    assert O.coeff_modulus_create(n, [int(b) for b in g["bits"]]) == primes
    c = O.PortContext("ckks", n, primes)
    assert [c.ntt_root(i) for i in range(len(primes))] == [int(x) for x in g["roots"]]
    assert np.array_equal(c.ntt(0, g["a"][0], "fwd"), g["ntt_fwd_a0"])
    assert np.array_equal(c.ntt(0, g["a"][0], "inv"), g["ntt_inv_a0"])
    m = c.multiply(g["a"], g["b"])
    assert np.array_equal(m, g["multiply"])
    r = c.relinearize(m, g["relin_key"])
    assert np.array_equal(r, g["relinearize"])
    s = c.rescale(r)
    assert np.array_equal(s, g["rescale"])
    rot = c.apply_galois(s, int(g["galois_elt"]), g["galois_key"])
    assert np.array_equal(rot, g["rotate1"])
    assert np.array_equal(c.drop_last(rot), g["mod_switch"])


def test_golden_bfv():
    g = np.load(os.path.join(GOLDEN, "bfv_n32.npz"))
    n, primes, t = int(g["n"]), [int(x) for x in g["primes"]], int(g["t"])
    c = O.PortContext("bfv", n, primes, t)
    K = len(primes) - 1
    assert c.base_bsk(K) == [int(x) for x in g["bsk"]]
    nBsk = len(g["bsk"])
    a0 = g["a"][0]
    assert np.array_equal(c.rns_stage(K, "fastbconv_m_tilde", a0, nBsk + 1), g["fastbconv_m_tilde"])
    assert np.array_equal(c.rns_stage(K, "sm_mrq", g["fastbconv_m_tilde"], nBsk), g["sm_mrq"])
    assert np.array_equal(c.rns_stage(K, "fast_floor", np.concatenate([a0, g["sm_mrq"]]), nBsk), g["fast_floor"])
    assert np.array_equal(c.rns_stage(K, "fastbconv_sk", g["fast_floor"], K), g["fastbconv_sk"])
    m = c.multiply(g["a"], g["b"])
    assert np.array_equal(m, g["multiply"])
    r = c.relinearize(m, g["relin_key"])
    assert np.array_equal(r, g["relinearize"])
    rot = c.apply_galois(r, int(g["galois_elt"]), g["galois_key"])
    assert np.array_equal(rot, g["rotate_rows1"])
    col = c.apply_galois(rot, 2 * n - 1, g["conj_key"])
    assert np.array_equal(col, g["rotate_columns"])
    assert np.array_equal(c.bfv_mod_switch(col), g["mod_switch"])


# ---- live against the real reference, other sizes / seeds than the golden files
@needs_ref
@pytest.mark.parametrize("n,bits", [(16, [30, 30, 30]), (256, [50, 40, 40, 40, 50]), (2048, [60, 50, 60])])
def test_port_vs_reference_ckks(n, bits):
    primes = R.coeff_modulus_create(n, bits)
    assert primes == O.coeff_modulus_create(n, bits)
    ref = R.RefContext("ckks", n, primes)
    ref.keygen_relin()
    elt = ref.galois_elt_from_step(-1)
    ref.keygen_galois_elts([elt])
    c = O.PortContext("ckks", n, primes)
    assert elt == c.galois_elt_from_step(-1)
    K, fc = len(primes) - 1, ref.first_chain_index
    rng = np.random.default_rng(n + 7)
    a, b = rand_ct(rng, primes, K, n), rand_ct(rng, primes, K, n)
    x, y = ref.ct(fc, a, True, 2.0 ** 10), ref.ct(fc, b, True, 2.0 ** 10)
    ref.multiply_inplace(x, y)
    m = c.multiply(a, b)
    assert np.array_equal(m, x.data())
    ref.relinearize_inplace(x)
    r = c.relinearize(m, ref.key("relin", 0))
    assert np.array_equal(r, x.data())
    z = ref.ct(fc, r, True, float(primes[K - 1]) * 2.0 ** 10)
    ref.rescale_to_next_inplace(z)
    s = c.rescale(r)
    assert np.array_equal(s, z.data())
    ref.apply_galois_inplace(z, elt)
    assert np.array_equal(c.apply_galois(s, elt, ref.key("galois", (elt - 1) >> 1)), z.data())


@needs_ref
@pytest.mark.parametrize("n,bits,tb", [(16, [36, 36, 37], 10), (128, [40, 40, 40, 40, 40], 14)])
def test_port_vs_reference_bfv(n, bits, tb):
    primes = R.coeff_modulus_create(n, bits)
    t = R.plain_modulus_batching(n, tb)
    assert t == O.plain_modulus_batching(n, tb)
    ref = R.RefContext("bfv", n, primes, t)
    ref.keygen_relin()
    c = O.PortContext("bfv", n, primes, t)
    K, fc = len(primes) - 1, ref.first_chain_index
    assert ref.behz_bases(fc)[0] == c.base_bsk(K)
    rng = np.random.default_rng(n + 11)
    a, b = rand_ct(rng, primes, K, n), rand_ct(rng, primes, K, n)
    x, y = ref.ct(fc, a, False), ref.ct(fc, b, False)
    ref.multiply_inplace(x, y)
    m = c.multiply(a, b)
    assert np.array_equal(m, x.data())
    ref.relinearize_inplace(x)
    r = c.relinearize(m, ref.key("relin", 0))
    assert np.array_equal(r, x.data())
    ref.mod_switch_to_next_inplace(x)
    assert np.array_equal(c.bfv_mod_switch(r), x.data())


@needs_ref
@pytest.mark.parametrize("n,bits,tb", [(16, [36, 36, 37, 38], 10), (128, [40, 50, 40, 45], 14)])
def test_port_vs_reference_bgv(n, bits, tb):
    """BGV restatements of the port: tensor product, key switch with the BGV mod-down (evaluator.cpp:2762-2805),
    mod_t_and_divide_q_last_ntt (rns.cpp:1193-1236), rotate."""
    primes = R.coeff_modulus_create(n, bits)
    t = R.plain_modulus_batching(n, tb)
    ref = R.RefContext("bgv", n, primes, t)
    ref.keygen_relin()
    elt = ref.galois_elt_from_step(1)
    ref.keygen_galois_elts([elt])
    c = O.PortContext("bgv", n, primes, t)
    K, fc = len(primes) - 1, ref.first_chain_index
    rng = np.random.default_rng(n + 13)
    a, b = rand_ct(rng, primes, K, n), rand_ct(rng, primes, K, n)
    x, y = ref.ct(fc, a, True, 1.0, 3), ref.ct(fc, b, True, 1.0, 5)
    ref.multiply_inplace(x, y)
    m = c.multiply(a, b)
    assert np.array_equal(m, x.data())
    ref.relinearize_inplace(x)
    r = c.relinearize(m, ref.key("relin", 0))
    assert np.array_equal(r, x.data())
    ref.apply_galois_inplace(x, elt)
    g = c.apply_galois(r, elt, ref.key("galois", (elt - 1) >> 1))
    assert np.array_equal(g, x.data())
    cf = x.info()["correction_factor"]
    ref.mod_switch_to_next_inplace(x)
    s, cf2 = c.bgv_mod_switch(g, cf)
    assert np.array_equal(s, x.data()) and cf2 == x.info()["correction_factor"]


@needs_ref
@pytest.mark.parametrize("scheme,n,bits,tb", [("bfv", 64, [40, 40, 41], 13), ("bfv", 128, [30, 30, 30], 40), ("bgv", 64, [40, 40, 41], 13)])
def test_port_vs_reference_plain_operands(scheme, n, bits, tb):
    """plaintext lift / transform_to_ntt(Plaintext), add_plain, sub_plain, multiply_plain (generic path) of the port"""
    primes = R.coeff_modulus_create(n, bits)
    t = R.plain_modulus_batching(n, tb)
    ref = R.RefContext(scheme, n, primes, t)
    c = O.PortContext(scheme, n, primes, t)
    K, fc = len(primes) - 1, ref.first_chain_index
    ntt = scheme == "bgv"
    rng = np.random.default_rng(n + 17)
    a = rand_ct(rng, primes, K, n)
    for m in (rng.integers(0, t, n, dtype=np.uint64), rng.integers(0, t, n // 3, dtype=np.uint64)):
        p = ref.pt_transform_to_ntt_inplace(ref.pt(m), fc)
        assert np.array_equal(c.plain_to_ntt(K, m).reshape(-1), p.data())
        for sub in (False, True):
            x = ref.ct(fc, a, ntt, 1.0, 3 if scheme == "bgv" else 1)
            (ref.sub_plain_inplace if sub else ref.add_plain_inplace)(x, ref.pt(m))
            assert np.array_equal(c.addsub_plain(a, m, sub, 3 if scheme == "bgv" else 1), x.data())
        x = ref.ct(fc, a, ntt, 1.0, 1)
        ref.multiply_plain_inplace(x, ref.pt(m))
        assert np.array_equal(c.multiply_plain(a, m), x.data())
