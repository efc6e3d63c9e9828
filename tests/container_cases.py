"""The container surface of the C ABI (include/sealhip.h section 1d; seal_amd/csrc/capi_containers.cpp) against the real reference:
ContextData constants and qualifiers of every level, EncryptionParameters streams, Ciphertext reserve / resize / word access
bookkeeping, KSwitchKeys copies and key lists, SecretKey / PublicKey streams.  Used by the CPU suite (emulated build) and the GPU suite.
TEST INFRASTRUCTURE (imports the oracle)."""
import numpy as np

import seal_amd as S
import sealref
from harness import DeviceSide
from oracle import Oracle, coeff_modulus_create, plain_modulus_batching, rand_ct


def case_context_data(scheme, n, bits, tb=20, sec_level=0):
    primes = coeff_modulus_create(n, bits)
    t = plain_modulus_batching(n, tb) if scheme != "ckks" else 0
    ref = sealref.RefContext(scheme, n, primes, t)
    d = DeviceSide(scheme, n, primes, t)
    ctx = d.ctx
    assert ctx.parameters_set() and ctx.parameter_error() == ("success", "valid")
    key = ctx.key_context_data()
    assert key.chain_index() == ref.key_chain_index and key.prev_context_data() is None
    assert key.parms_id() == ctx.key_parms_id() == ref.parms_id(ref.key_chain_index)
    assert ctx.first_context_data().chain_index() == ref.first_chain_index
    assert ctx.last_context_data().chain_index() == 0 and ctx.last_context_data().next_context_data() is None
    assert ctx.get_context_data((1, 2, 3, 4)) is None
    seen = 0
    cd = key
    while cd is not None:
        ci = cd.chain_index()
        assert ctx.get_context_data(cd.parms_id()).chain_index() == ci
        level_primes = ref.level_primes(ci)
        p = cd.parms()
        assert p.coeff_modulus() == level_primes and p.poly_modulus_degree() == n and p.plain_modulus() == t and p.scheme == scheme
        assert p.parms_id() == cd.parms_id() == ref.parms_id(ci)
        q = ref.qualifiers(ci)
        mine = cd.qualifiers()
        for k in ("parameters_set", "using_fft", "using_ntt", "using_batching", "using_fast_plain_lift", "using_descending_modulus_chain", "sec_level"):
            assert mine[k] == q[k], (ci, k, mine, q)
        assert cd.total_coeff_modulus_bit_count() == q["total_coeff_modulus_bit_count"]
        assert cd.plain_upper_half_threshold() == q["plain_upper_half_threshold"]
        assert cd.total_coeff_modulus() == ref.data_words(ci, 0)
        assert cd.coeff_div_plain_modulus() == ref.data_words(ci, 1), (scheme, ci)
        assert cd.plain_upper_half_increment() == ref.data_words(ci, 2), (scheme, ci)
        assert cd.upper_half_threshold() == ref.data_words(ci, 3)
        assert cd.upper_half_increment() == ref.data_words(ci, 4)
        nxt = cd.next_context_data()
        if nxt is not None:
            assert nxt.chain_index() == ci - 1 and nxt.prev_context_data().chain_index() == ci
        # EncryptionParameters::save of this level: byte for byte, every compression mode the library has; loads back equal
        stream = p.save_bytes(0)
        assert stream == ref.parms_save(ci, 0), (scheme, ci)
        for mode in (0, 1, 2):
            try:
                mine_stream = p.save_bytes(mode)
            except S.InvalidArgument:
                assert mode == 2   # zstd absent on this host
                continue
            back = S.EncryptionParameters(scheme)
            assert back.load_bytes(mine_stream) == len(mine_stream)
            assert back.equals(p) and back.parms_id() == p.parms_id()
            if mode != 2:   # (the reference build of this image has no zstd: oracle/ref_config)
                assert sealref.parms_load(mine_stream) == (S.api.SCHEME[scheme], n, level_primes, t)
        # the array getters' capacity convention: a short buffer is refused, the length comes back
        import ctypes as C
        cnt = C.c_uint64(1)
        buf = (C.c_uint64 * 1)()
        hr = S._native.lib().ContextData_TotalCoeffModulus(cd._h, C.byref(cnt), buf) & 0xFFFFFFFF
        assert (hr == 0x80070057 and cnt.value == len(level_primes)) if len(level_primes) > 1 else hr == 0
        seen += 1
        cd = nxt
    assert seen == ref.key_chain_index + 1
    # copies and assignment of parameter objects
    p = key.parms()
    c = p.copy()
    assert c.equals(p)
    c.set_poly_modulus_degree(2 * n)
    assert not c.equals(p)
    c.assign(p)
    assert c.equals(p) and c.parms_id() == p.parms_id()
    # malformed parameter streams fail with the reference's class
    good = p.save_bytes(0)
    for name, bad in (("truncated", good[:-3]), ("bad magic", b"\x00\x00" + good[2:]), ("scheme 9", good[:16] + b"\x09" + good[17:]),
                      ("huge degree", good[:17] + (1 << 40).to_bytes(8, "little") + good[25:])):
        try:
            sealref.parms_load(bad)
            ref_exc = None
        except sealref.RefError as e:
            ref_exc = e.code
        back = S.EncryptionParameters(scheme)
        try:
            back.load_bytes(bad)
            mine_exc = None
        except S.InvalidArgument:
            mine_exc = 1
        except S.LogicError:
            mine_exc = 2
        except S.DeviceError:   # runtime_error("I/O error") -> COR_E_IO
            mine_exc = 4
        assert mine_exc == ref_exc, (name, mine_exc, ref_exc)


def case_security_level():
    """sec_level 128: parameters the standard allows build, larger ones are refused (the reference marks them invalid: context.cpp:219-231)"""
    n = 4096
    ok = coeff_modulus_create(n, [36, 36, 37])       # 109 bits
    p = S.EncryptionParameters("ckks")
    p.set_poly_modulus_degree(n)
    p.set_coeff_modulus(ok)
    ctx = S.SEALContext(p, True, 128)
    assert ctx.key_context_data().qualifiers()["sec_level"] == 128
    p.set_coeff_modulus(coeff_modulus_create(n, [40, 40, 40]))
    for level in (128, 192, 256):
        try:
            S.SEALContext(p, True, level)
            raise AssertionError("insecure parameters accepted at sec_level %d" % level)
        except S.InvalidArgument:
            pass
    try:
        S.SEALContext(p, True, 100)
        raise AssertionError("sec_level 100 accepted")
    except S.InvalidArgument:
        pass
    assert S.SEALContext(p, True, 0).key_context_data().qualifiers()["sec_level"] == 0
    # malformed AND too large: the reference checks the modulus before the security level (context.cpp:142-231), so the message names
    # the malformation (the early security verdict used to win)
    bad = coeff_modulus_create(n, [40, 40, 40])
    bad[1] = bad[1] + 2   # even + 2 = not a prime any more (an odd composite or an even number)
    while _is_probable_prime(bad[1]):
        bad[1] += 2
    p.set_coeff_modulus(bad)
    try:
        S.SEALContext(p, True, 128)
        raise AssertionError("a composite modulus was accepted")
    except S.InvalidArgument as e:
        assert "non_prime" in str(e) and "secure" not in str(e), str(e)


def _is_probable_prime(q):
    if q < 2 or q % 2 == 0:
        return q == 2
    d, r = q - 1, 0
    while d % 2 == 0:
        d //= 2
        r += 1
    for a in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        if a % q == 0:
            continue
        x = pow(a, d, q)
        if x in (1, q - 1):
            continue
        for _ in range(r - 1):
            x = x * x % q
            if x == q - 1:
                break
        else:
            return False
    return True


def case_ciphertext_container(scheme, n, bits, tb=20):
    rng = np.random.default_rng(5)
    primes = coeff_modulus_create(n, bits)
    t = plain_modulus_batching(n, tb) if scheme != "ckks" else 0
    o = Oracle(scheme, n, primes, t)
    d = DeviceSide(scheme, n, primes, t)
    K = len(primes) - 1
    ci = o._ci(K)
    ntt = scheme != "bfv"
    slab = rand_ct(rng, primes, K, n, size=3)
    c = d.ct(slab, is_ntt=ntt)
    r = o.ref.ct(ci, slab, ntt, c.scale(), 1)

    def same(what):
        info = r.info()
        assert (c.size(), c.coeff_modulus_size()) == (info["size"], info["coeff_modulus_size"]), what
        if c.size():
            assert np.array_equal(c.to_numpy()[:, 0], r.data()), what

    # word access: operator[] / data(poly)[coeff] (ciphertext.h:337-379)
    flat = slab.reshape(-1)
    for idx in (0, 1, n, K * n - 1, K * n, flat.size - 1):
        assert c.get_data_at(idx) == int(flat[idx])
    assert c.get_data_at(2, K * n - 1) == int(slab[2].reshape(-1)[-1])
    for fn, args in ((c.get_data_at, (flat.size,)), (c.get_data_at, (3, 0)), (c.get_data_at, (0, K * n)), (c.set_data_at, (flat.size, 1))):
        try:
            fn(*args)
            raise AssertionError("index past the end accepted: %r" % (args,))
        except S.OutOfRange:
            pass
    c.set_data_at(5, 12345)
    assert c.get_data_at(5) == 12345
    c.set_data_at(5, int(flat[5]))
    # reserve / resize bookkeeping follows the reference's: size, size_capacity, and the kept words
    steps = [(0, ci, 5), (3, None, 2), (1, None, 4), (3, None, 4), (2, ci, 3), (1, None, 2), (0, ci, 2), (3, None, 0), (3, None, 2)]
    for op, level, count in steps:
        ref_state = o.ref.ct_container_op(r, op, level if level is not None else 0, count)
        if op == 0:
            c.reserve(count, parms_id=d.ctx.parms_id_at(level))
        elif op == 1:
            c.reserve(count, with_context=False)
        elif op == 2:
            c.resize(d.ctx.parms_id_at(level), count)
        else:
            c.resize_same_level(count, with_context=False)
        assert (c.size(), c.size_capacity(), c.coeff_modulus_size(), c.poly_modulus_degree() or n) == (ref_state[0], ref_state[1], ref_state[2], ref_state[3] or n), (op, count, ref_state)
        same("after op %d count %d" % (op, count))
    for bad in (1, 17):
        for fn in (lambda: c.reserve(bad, with_context=False), lambda: c.reserve(bad), lambda: c.resize_same_level(bad)):
            try:
                fn()
                raise AssertionError("size %d accepted" % bad)
            except S.InvalidArgument:
                pass
    # Ciphertext(context, parms_id[, capacity]) and the loader's resize(size, N, K)
    e = S.Ciphertext.with_parms_id(d.ctx, d.ctx.parms_id_at(ci))
    assert (e.size(), e.size_capacity(), e.parms_id()) == (0, 2, d.ctx.parms_id_at(ci))
    e = S.Ciphertext.with_parms_id(d.ctx, d.ctx.parms_id_at(ci), 4)
    assert (e.size(), e.size_capacity()) == (0, 4)
    e.resize_geometry(2, n, K)
    assert (e.size(), e.coeff_modulus_size()) == (2, K)
    try:
        e.resize_geometry(2, 2 * n, K)
        raise AssertionError("foreign geometry accepted")
    except S.InvalidArgument:
        pass
    # set_parms_id: the ids of the chain and parms_id_zero only
    try:
        e.set_parms_id((9, 9, 9, 9))
        raise AssertionError("unknown parms_id accepted")
    except S.InvalidArgument:
        pass
    # ... and never a level whose polynomials would not fit the slab: a size-2 ciphertext at the LAST level (K = 1),
    # parms_id_zero first (no level: nothing to compare K with), then the first level's id - size x K x N would exceed the capacity
    if K >= 2:
        last = d.ctx.parms_id_at(0)
        f = S.Ciphertext(d.ctx)
        f.resize(last, 2)
        cap_words = f.size_capacity() * f.coeff_modulus_size() * n
        f.set_parms_id((0, 0, 0, 0))
        try:
            f.set_parms_id(d.ctx.parms_id_at(ci))
            raise AssertionError("a level that does not fit the slab was accepted: size %d K %d capacity %d words" % (f.size(), f.coeff_modulus_size(), cap_words))
        except S.InvalidArgument:
            pass
        f.set_parms_id(last)  # the level it was allocated for is still fine
        assert (f.size(), f.coeff_modulus_size(), f.size_capacity()) == (2, 1, 2)
    # release(): an empty object that can be used again
    c.release()
    o.ref.ct_container_op(r, 4, 0, 0)
    assert (c.size(), c.size_capacity(), c.parms_id()) == (0, 0, (0, 0, 0, 0)) and not c.is_ntt_form() and c.scale() == 1.0
    c.resize(d.ctx.parms_id_at(ci), 2)
    assert c.size() == 2 and not np.any(c.to_numpy())


def case_keys_container(scheme, n, bits, tb=20):
    primes = coeff_modulus_create(n, bits)
    t = plain_modulus_batching(n, tb) if scheme != "ckks" else 0
    probe = Oracle(scheme, n, primes, t)
    elt = probe.galois_elt_from_step(1)
    o = Oracle(scheme, n, primes, t, galois_elts=[elt])
    d = DeviceSide(scheme, n, primes, t)
    d.upload_keys(o)
    L, K = len(primes), len(primes) - 1
    key_id = d.ctx.key_parms_id()
    # SecretKey / PublicKey: save equals the reference's stream byte for byte; copies, parms_id
    sk = S.SecretKey(d.ctx)
    assert sk.parms_id() == (0, 0, 0, 0)
    sk.set(o.ref.secret_key())
    assert sk.parms_id() == key_id
    assert sk.save_bytes(0) == o.ref.secret_key_save()
    pk = S.PublicKey(d.ctx, o.ref.public_key())
    assert pk.save_bytes(0) == o.ref.public_key_save() and pk.parms_id() == key_id
    for obj, cls in ((sk, S.SecretKey), (pk, S.PublicKey)):
        twin = obj.copy()
        assert np.array_equal(twin.words(L, n), obj.words(L, n))
        other = cls(d.ctx)
        other.assign(obj)
        assert np.array_equal(other.words(L, n), obj.words(L, n)) and other.save_bytes(0) == obj.save_bytes(0)
        for mode in (1, 2):
            try:
                packed = obj.save_bytes(mode)
            except S.InvalidArgument:
                continue
            back = cls(d.ctx)
            assert back.load_bytes(packed) == len(packed) and np.array_equal(back.words(L, n), obj.words(L, n))
    # KSwitchKeys: deep copy, key lists, AddKeyList
    rlk, glk = d.rlk, d.glk
    assert rlk.parms_id() == key_id and rlk.raw_size() == 1 and rlk.size() == 1
    twin = rlk.copy()
    assert twin.save_bytes(0) == rlk.save_bytes(0)
    digits = rlk.key_list(0)
    assert len(digits) == K
    want = o.relin_key()
    for j, dg in enumerate(digits):
        assert np.array_equal(dg.words(L, n), want[j]), "digit %d of the relinearization key" % j
    built = S.RelinKeys(d.ctx)
    built.add_key_list(digits)
    assert built.raw_size() == 1 and built.save_bytes(0) == rlk.save_bytes(0)
    # the rebuilt and the copied key relinearize to the reference's words
    rng = np.random.default_rng(9)
    x3 = rand_ct(rng, primes, K, n, size=3)
    ntt = scheme != "bfv"
    expect = o.relinearize(x3)
    for keys in (twin, built):
        c = d.ct(x3, is_ntt=ntt)
        d.ev.relinearize_inplace(c, keys)
        assert np.array_equal(c.to_numpy()[:, 0], expect)
    gidx = S.GaloisKeys.get_index(elt)
    assert glk.raw_size() == gidx + 1 and glk.size() == 1
    gt = glk.copy()
    assert gt.has_index(gidx) and not gt.has_index(0) and gt.key_list(0) == [] and len(gt.key_list(gidx)) == K
    assert gt.save_bytes(0) == glk.save_bytes(0)
    gt.assign(rlk)
    assert gt.raw_size() == 1 and gt.save_bytes(0) == rlk.save_bytes(0)
    gt.clear_data_and_reserve(4)
    assert gt.raw_size() == 0 and gt.size() == 0
    gt.add_key_list([])
    assert gt.raw_size() == 1 and gt.size() == 0
    twin.set_parms_id((1, 2, 3, 4))
    assert twin.parms_id() == (1, 2, 3, 4)
