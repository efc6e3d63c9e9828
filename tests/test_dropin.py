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


def _chain_program(lib, n, primes, a, b, b2, probe):
    ctx = lib.RefContext("ckks", n, primes)
    ctx.keygen_relin()
    ctx.keygen_galois_steps([1])
    fc = ctx.first_chain_index
    x, y = ctx.ct(fc, a, True, 2.0 ** 10, 1), ctx.ct(fc, b, True, 2.0 ** 10, 1)
    out = []
    s0 = probe()
    ctx.multiply_inplace(x, y)
    ctx.relinearize_inplace(x)
    ctx.rescale_to_next_inplace(x)
    ctx.rotate_vector_inplace(x, 1)
    s1 = probe()
    snap_copy = x.copy()                 # host-side copy construction of a device-resident object (read fault)
    out.append(snap_copy.data().copy())
    out.append(x.data().copy())
    s2 = probe()
    ctx.square_inplace(x)                # x was read on the host, not written
    ctx.relinearize_inplace(x)
    s3 = probe()
    ctx.multiply_inplace(snap_copy, snap_copy.copy())
    out.append(snap_copy.data().copy())
    z = ctx.ct(fc, a, True, 2.0 ** 10, 1)
    ctx.multiply_inplace(z, y)           # y: uploaded once, above
    s4 = probe()
    out.append(z.data().copy())
    # the second operand is overwritten on the host (operator=): its device copy must not be used again
    y2 = ctx.ct(fc, b2, True, 2.0 ** 10, 1)
    ctx.ct_assign(y, y2)
    w = ctx.ct(fc, a, True, 2.0 ** 10, 1)
    ctx.multiply_inplace(w, y)
    out.append(w.data().copy())
    out.append(x.data().copy())
    return out, (s0, s1, s2, s3, s4)


def _chain_inputs(n, bits):
    primes = R.coeff_modulus_create(n, bits)
    K = len(primes) - 1
    rng = np.random.default_rng(12)
    mk = lambda: np.stack([np.stack([rng.integers(0, primes[i], n, dtype=np.uint64) for i in range(K)]) for _ in range(2)])  # noqa: E731
    return primes, mk(), mk(), mk()


def _dropin_stats(D):
    import ctypes as C
    v = [C.c_uint64() for _ in range(4)]
    D.lib().sealhip_dropin_stats(*[C.byref(x) for x in v])
    return dict(zip(("uploads", "downloads", "reused", "faults"), [x.value for x in v]))


def _device_resident_chain(lib_name, n, bits, emulated):
    """SURVEY 8(f) N2 behind the headers: a chain of in-place operations on seal::Ciphertext objects runs without PCIe traffic
    (the host buffers are protected shadows of the device copies), host accesses of every kind see the right words, and a
    host write to an operand invalidates its device copy.  Words: in this process, next to the reference (both libraries
    loaded: they then share the reference's inline-static memory manager, the reference's pool serves both and the buffers are
    NOT page-aligned - the unaligned head / tail path).  Transfer counts: in a process that loads only the drop-in (its own
    pool through the SEAL_MALLOC hook: page-aligned buffers)."""
    import subprocess
    import sys
    import tempfile
    path = os.path.join(BUILD, lib_name)
    primes, a, b, b2 = _chain_inputs(n, bits)
    ref_out, _ = _chain_program(R, n, primes, a, b, b2, lambda: None)
    D = _bind(path)
    was = D.lib().sealhip_dropin_set_resident(1)   # opt-in since round 3 (the default copies every result down)
    try:
        got_out, _ = _chain_program(D, n, primes, a, b, b2, lambda: _dropin_stats(D))
    finally:
        D.lib().sealhip_dropin_set_resident(was)
    assert len(ref_out) == len(got_out)
    for i, (r, g) in enumerate(zip(ref_out, got_out)):
        assert r.shape == g.shape and np.array_equal(r, g), "snapshot %d differs from the reference" % i
    with tempfile.TemporaryDirectory() as tmp:
        out_path = os.path.join(tmp, "out.npz")
        code = ("import sys, os, json; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
                "import numpy as np, seal_amd as S\n"
                "S.load(%r) if %r else S.load()\n"
                "import test_dropin as T\n"
                "D = T._bind(%r)\n"
                "D.lib().sealhip_dropin_set_resident(1)\n"
                "primes, a, b, b2 = T._chain_inputs(%d, %r)\n"
                "out, st = T._chain_program(D, %d, primes, a, b, b2, lambda: T._dropin_stats(D))\n"
                "np.savez(%r, *out)\n"
                "print('STATS ' + json.dumps(st))\n"
                % (HERE, os.path.dirname(HERE), os.path.join(HERE, "hipemu", "libsealhip_emu.so"), emulated, path, n, bits, n, out_path))
        run = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900,
                             env=dict(os.environ, SEALHIP_COMM_NO_RCCL="1") if emulated else dict(os.environ))
        assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-2000:]
        import json
        st = json.loads([ln for ln in run.stdout.splitlines() if ln.startswith("STATS ")][0][6:])
        alone = np.load(out_path)
        for i, r in enumerate(ref_out):
            assert np.array_equal(r, alone["arr_%d" % i]), "snapshot %d (drop-in alone in its process) differs from the reference" % i
    s0, s1, s2, s3, s4 = st
    # the four chained operations moved x and y up once and nothing down
    assert s1["uploads"] - s0["uploads"] == 2 and s1["downloads"] == s0["downloads"] and s1["faults"] == s0["faults"], (s0, s1)
    assert s1["reused"] - s0["reused"] >= 3
    # the host copy and the snapshot cost exactly one download (x stays mirrored read-only afterwards)
    assert s2["downloads"] - s1["downloads"] == 1 and s2["faults"] > s1["faults"], (s1, s2)
    assert s3["uploads"] == s2["uploads"], "x was read on the host, not written: its device copy is still good"
    assert s4["reused"] > s3["reused"], "an operand uploaded once is found on the device the second time"


def test_dropin_resident_switch_is_symmetric_emulated(emu):
    """sealhip_dropin_set_resident(0) gives the process its SIGSEGV disposition back: installed by the opt-in, restored by the
    opt-out, installed again by a second opt-in.  In a process of its own: the disposition is process-wide state."""
    import subprocess
    import sys
    path = os.path.join(BUILD, "libsealdropin_emu.so")
    if not (R.available() and os.path.exists(path)):
        pytest.skip("needs oracle/_ref and integration/_build (make -C integration)")
    code = ("import ctypes as C, sys\n"
            "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import seal_amd as S; S.load(%r)\n"
            "class SA(C.Structure):\n"
            "    _fields_ = [('handler', C.c_void_p), ('mask', C.c_ubyte * 128), ('flags', C.c_int), ('restorer', C.c_void_p)]\n"
            "libc = C.CDLL(None)\n"
            "def handler():\n"
            "    sa = SA(); assert libc.sigaction(11, None, C.byref(sa)) == 0; return sa.handler or 0\n"
            "D = C.CDLL(%r)\n"
            "before = handler()\n"
            "assert D.sealhip_dropin_set_resident(1) == 0\n"
            "inside = handler()\n"
            "assert D.sealhip_dropin_set_resident(0) == 1\n"
            "after = handler()\n"
            "assert D.sealhip_dropin_set_resident(1) == 0\n"
            "again = handler()\n"
            "D.sealhip_dropin_set_resident(0)\n"
            "print('HANDLERS', before, inside, after, again, handler())\n"
            % (HERE, os.path.dirname(HERE), os.path.join(HERE, "hipemu", "libsealhip_emu.so"), path))
    run = subprocess.run([sys.executable, "-X", "faulthandler=0", "-c", code], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, SEALHIP_COMM_NO_RCCL="1", PYTHONFAULTHANDLER=""))
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-2000:]
    before, inside, after, again, last = [int(v) for v in [ln for ln in run.stdout.splitlines() if ln.startswith("HANDLERS")][0].split()[1:]]
    assert inside != before, "the opt-in installs the fault handler"
    assert after == before and last == before, "the opt-out restores what was there"
    assert again == inside


def test_dropin_device_resident_chain_emulated(emu):
    if not (R.available() and os.path.exists(os.path.join(BUILD, "libsealdropin_emu.so"))):
        pytest.skip("needs oracle/_ref and integration/_build (make -C integration)")
    _device_resident_chain("libsealdropin_emu.so", 4096, [60, 40, 40, 60], True)


@pytest.mark.gpu
@pytest.mark.parametrize("n,bits", [(8192, [60, 40, 40, 60]), (65536, [60] + [50] * 14 + [60])])
def test_dropin_device_resident_chain_gpu(gpu, n, bits):
    if not (R.available() and os.path.exists(os.path.join(BUILD, "libsealdropin.so"))):
        pytest.skip("needs oracle/_ref and integration/_build (make -C integration)")
    _device_resident_chain("libsealdropin.so", n, bits, False)


# ---- the reference's contract: concurrent calls on different ciphertexts are safe (evaluator.h:79-87, memorymanager.h:26-52)
def _concurrent(path, n, bits, threads=4, rounds=2):
    """`threads` host threads, each with its own ciphertexts, run multiply / relinearize / rescale / rotate chains through ONE
    seal::Evaluator of the drop-in at the same time (default mode: results copied down per call); every result equals the
    reference's.  Also with a settle + mode switch in between, which must not disturb the others' objects."""
    import threading
    primes = R.coeff_modulus_create(n, bits)
    K = len(primes) - 1
    D = _bind(path)
    rng = np.random.default_rng(5)
    mk = lambda: np.stack([np.stack([rng.integers(0, primes[i], n, dtype=np.uint64) for i in range(K)]) for _ in range(2)])  # noqa: E731
    inputs = [(mk(), mk()) for _ in range(threads)]
    sides = []
    for lib in (R, D):
        ctx = lib.RefContext("ckks", n, primes, 0)
        ctx.keygen_relin()
        ctx.keygen_galois_steps([1])
        sides.append(ctx)
    ref_ctx, dev_ctx = sides
    sc = float(primes[K - 1]) * 2.0 ** 10

    def program(ctx, a, b):
        fc = ctx.first_chain_index
        x, y = ctx.ct(fc, a, True, 2.0 ** 10, 1), ctx.ct(fc, b, True, 2.0 ** 10, 1)
        ctx.multiply_inplace(x, y)
        ctx.relinearize_inplace(x)
        z = ctx.ct(fc, x.data(), True, sc, 1)
        ctx.rescale_to_next_inplace(z)
        ctx.rotate_vector_inplace(z, 1)
        ctx.add_inplace(z, z.copy())
        return z.data().copy()
    expected = [program(ref_ctx, a, b) for a, b in inputs]
    for _ in range(rounds):
        got, errs = [None] * threads, []
        start = threading.Barrier(threads)

        def worker(i):
            try:
                start.wait()
                got[i] = program(dev_ctx, *inputs[i])
            except Exception as e:   # noqa: BLE001
                errs.append((i, e))
        ts = [threading.Thread(target=worker, args=(i,)) for i in range(threads)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        assert not errs, errs
        for i in range(threads):
            assert np.array_equal(got[i], expected[i]), "thread %d: words differ from the reference" % i
        D.lib().sealhip_dropin_settle(None, 0)   # a no-op in the default mode; must be callable at any time


def test_dropin_concurrent_emulated(emu):
    path = os.path.join(BUILD, "libsealdropin_emu.so")
    if not (R.available() and os.path.exists(path)):
        pytest.skip("needs oracle/_ref and integration/_build (make -C integration)")
    _concurrent(path, 1024, [40, 30, 30, 40], threads=3, rounds=1)


@pytest.mark.gpu
@pytest.mark.parametrize("n,bits", [(8192, [60, 40, 40, 60]), (32768, [60, 50, 50, 50, 60])])
def test_dropin_concurrent_gpu(gpu, n, bits):
    path = os.path.join(BUILD, "libsealdropin.so")
    if not (R.available() and os.path.exists(path)):
        pytest.skip("needs oracle/_ref and integration/_build (make -C integration)")
    _concurrent(path, n, bits, threads=4, rounds=2)
