"""The header-compatible drop-in of INTEGRATION.md section 2: seal::Evaluator implemented on the C ABI of libsealhip
(integration/seal_evaluator_hip.cpp) and linked with the REST of the real reference (every object of oracle/_ref except
evaluator.o).  The same flat test shim drives both libraries; every operation must give the same words, metadata and
exception class through the reference's own C++ API."""
import importlib.util
import os

import numpy as np
import pytest

import sealref as R

HERE = os.path.dirname(os.path.abspath(__file__))
BUILD = os.path.join(HERE, "..", "integration", "_build")


def _bind(path):
    """a second instance of the sealref ctypes wrapper bound to another library"""
    spec = importlib.util.spec_from_file_location("sealref_dropin_" + os.path.basename(path).replace(".", "_"), R.__file__)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.LIB_PATH = path
    return mod


def _run(D, scheme, n, bits, tb, seed):
    primes = R.coeff_modulus_create(n, bits)
    t = R.plain_modulus_batching(n, tb) if scheme != "ckks" else 0
    ntt = scheme != "bfv"
    scale = 2.0 ** 10 if scheme == "ckks" else 1.0
    rng = np.random.default_rng(seed)
    K = len(primes) - 1
    a = np.stack([np.stack([rng.integers(0, primes[i], n, dtype=np.uint64) for i in range(K)]) for _ in range(2)])
    b = np.stack([np.stack([rng.integers(0, primes[i], n, dtype=np.uint64) for i in range(K)]) for _ in range(2)])
    outs = []
    for lib in (R, D):
        ctx = lib.RefContext(scheme, n, primes, t)   # same seed: the same keys on both sides
        ctx.keygen_relin()
        ctx.keygen_galois_elts([ctx.galois_elt_from_step(1), 2 * n - 1])
        fc = ctx.first_chain_index
        res = []
        x, y = ctx.ct(fc, a, ntt, scale, 1), ctx.ct(fc, b, ntt, scale, 1)

        def snap(c):
            res.append((c.data().copy(), tuple(sorted(c.info().items()))))
        ctx.add_inplace(x, y); snap(x)
        ctx.multiply_inplace(x, y); snap(x)
        ctx.relinearize_inplace(x); snap(x)
        if scheme == "ckks":
            ctx.rescale_to_next_inplace(x); snap(x)
            ctx.rotate_vector_inplace(x, 1); snap(x)
            ctx.complex_conjugate_inplace(x); snap(x)
        else:
            ctx.mod_switch_to_next_inplace(x); snap(x)
            ctx.rotate_rows_inplace(x, 1); snap(x)
            ctx.rotate_columns_inplace(x); snap(x)
            m = rng.integers(0, t, n, dtype=np.uint64) if lib is R else m_saved
            m_saved = m
            ctx.add_plain_inplace(x, ctx.pt(m)); snap(x)
            ctx.multiply_plain_inplace(x, ctx.pt(m)); snap(x)
        ctx.square_inplace(x); snap(x)
        ctx.negate_inplace(x); snap(x)
        z = ctx.ct(fc, a, ntt, scale, 1)
        ctx.transform_from_ntt_inplace(z) if ntt else ctx.transform_to_ntt_inplace(z)
        snap(z)
        # error parity: the same exception class for the same invalid call
        w = ctx.ct(fc, a, not ntt if scheme != "bgv" else False, scale, 1)
        try:
            ctx.multiply_inplace(w, w.copy())
            res.append("no error")
        except lib.RefError as e:
            res.append(str(e))
        outs.append(res)
    assert len(outs[0]) == len(outs[1])
    for i, (r, d) in enumerate(zip(outs[0], outs[1])):
        if isinstance(r, str):
            assert r == d, "step %d: reference raised %r, drop-in %r" % (i, r, d)
            continue
        assert r[1] == d[1], "step %d metadata: %r vs %r" % (i, r[1], d[1])
        assert np.array_equal(r[0], d[0]), "step %d: words differ" % i


CASES = [("ckks", 64, [40, 30, 30, 40], 0), ("bfv", 64, [40, 40, 41], 13), ("bgv", 64, [40, 40, 41], 13)]


@pytest.mark.parametrize("scheme,n,bits,tb", CASES)
def test_dropin_emulated(emu, scheme, n, bits, tb):
    path = os.path.join(BUILD, "libsealdropin_emu.so")
    if not (R.available() and os.path.exists(path)):
        pytest.skip("needs oracle/_ref and integration/_build (make -C integration)")
    _run(_bind(path), scheme, n, bits, tb, seed=3)


@pytest.mark.gpu
@pytest.mark.parametrize("scheme,n,bits,tb", CASES + [("ckks", 8192, [60, 40, 40, 60], 0), ("bfv", 8192, [50, 55, 56], 20),
                                                      ("ckks", 32768, [60, 50, 50, 60], 0)])
def test_dropin_gpu(gpu, scheme, n, bits, tb):
    path = os.path.join(BUILD, "libsealdropin.so")
    if not (R.available() and os.path.exists(path)):
        pytest.skip("needs oracle/_ref and integration/_build (make -C integration)")
    _run(_bind(path), scheme, n, bits, tb, seed=4)


def _end_to_end(D, n_ckks, n_bfv):
    """A program against the reference's classes - KeyGenerator, CKKSEncoder / BatchEncoder, Encryptor, Evaluator,
    Decryptor - with the Evaluator being the drop-in: decrypted results must be the products of the inputs."""
    primes = R.coeff_modulus_create(n_ckks, [60, 40, 40, 60])
    ctx = D.RefContext("ckks", n_ckks, primes)
    ctx.keygen_relin()
    ctx.keygen_galois_steps([1])
    rng = np.random.default_rng(9)
    slots = n_ckks // 2
    u, v = rng.uniform(-1, 1, slots), rng.uniform(-1, 1, slots)
    x, y = ctx.ckks_encrypt(u, 2.0 ** 40), ctx.ckks_encrypt(v, 2.0 ** 40)
    ctx.multiply_inplace(x, y)
    ctx.relinearize_inplace(x)
    ctx.rescale_to_next_inplace(x)
    got = ctx.ckks_decrypt(x, slots)
    assert np.max(np.abs(got - u * v)) < 1e-6
    ctx.rotate_vector_inplace(x, 1)
    got = ctx.ckks_decrypt(x, slots)
    assert np.max(np.abs(got - np.roll(u * v, -1))) < 1e-6
    # BFV: batched integers modulo t
    primes = R.coeff_modulus_create(n_bfv, [40, 40, 41])
    t = R.plain_modulus_batching(n_bfv, 20)
    ctx = D.RefContext("bfv", n_bfv, primes, t)
    ctx.keygen_relin()
    a, b = rng.integers(0, t, n_bfv, dtype=np.uint64), rng.integers(0, t, n_bfv, dtype=np.uint64)
    x, y = ctx.batch_encrypt(a), ctx.batch_encrypt(b)
    ctx.multiply_inplace(x, y)
    ctx.relinearize_inplace(x)
    got = ctx.batch_decrypt(x, n_bfv)
    want = np.array([(int(p) * int(q)) % t for p, q in zip(a, b)], dtype=np.uint64)
    assert np.array_equal(got, want)


def test_dropin_end_to_end_emulated(emu):
    path = os.path.join(BUILD, "libsealdropin_emu.so")
    if not (R.available() and os.path.exists(path)):
        pytest.skip("needs oracle/_ref and integration/_build (make -C integration)")
    _end_to_end(_bind(path), 1024, 1024)


@pytest.mark.gpu
def test_dropin_end_to_end_gpu(gpu):
    path = os.path.join(BUILD, "libsealdropin.so")
    if not (R.available() and os.path.exists(path)):
        pytest.skip("needs oracle/_ref and integration/_build (make -C integration)")
    _end_to_end(_bind(path), 8192, 4096)
