"""Decryption cases shared by the CPU (emulated kernels) and `-m gpu` suites: ciphertexts encrypted by the REAL
reference (oracle/_ref), evaluated on either side, are decrypted through the C ABI (Decryptor_Decrypt / Decryptor_DecryptBatch)
and must give the plaintext words, coefficient count, parms_id and scale of the reference's own Decryptor::decrypt.
TEST INFRASTRUCTURE: the reference is the checker."""
import numpy as np

import seal_amd as S
import sealref
from harness import DeviceSide
from oracle import coeff_modulus_create, plain_modulus_batching


def _setup(scheme, n, bits, tbits=20):
    primes = coeff_modulus_create(n, bits)
    t = plain_modulus_batching(n, tbits) if scheme != "ckks" else 0
    ref = sealref.RefContext(scheme, n, primes, t)
    d = DeviceSide(scheme, n, primes, t)
    sk = S.SecretKey(d.ctx)
    assert sk.load_bytes(ref.secret_key_save()) > 0            # the serialized SecretKey ...
    sk2 = S.SecretKey(d.ctx, ref.secret_key())                  # ... and its raw words give the same decryptor
    return primes, t, ref, d, S.Decryptor(d.ctx, sk), S.Decryptor(d.ctx, sk2)


def _to_device(d, rct):
    i = rct.info()
    return S.Ciphertext.from_numpy(d.ctx, rct.data(), d.ctx.parms_id_at(i["chain_index"]), i["is_ntt_form"], i["scale"], i["correction_factor"])


def _same_plain(pt, rpt, what):
    ri = rpt.info()
    assert pt.coeff_count() == ri["coeff_count"], (what, pt.coeff_count(), ri["coeff_count"])
    assert pt.is_ntt_form() == ri["is_ntt_form"], what
    if ri["is_ntt_form"]:
        assert pt.scale() == ri["scale"], what
    assert np.array_equal(pt.to_numpy(), rpt.data()), what


def case_decrypt(scheme, n, bits, batch=3):
    primes, t, ref, d, dec, dec2 = _setup(scheme, n, bits)
    ref.keygen_relin()
    rng = np.random.default_rng(17)

    def fresh():
        if scheme == "ckks":
            return ref.ckks_encrypt(rng.standard_normal(n // 2), 2.0 ** 30)
        return ref.batch_encrypt(rng.integers(0, t, n, dtype=np.uint64))

    # fresh ciphertext (size 2, first data level)
    a, b = fresh(), fresh()
    for dcr in (dec, dec2):
        _same_plain(dcr.decrypt(_to_device(d, a)), ref.decrypt(a), "fresh")

    def budget(c, what):
        if scheme == "ckks":
            try:
                dec.invariant_noise_budget(_to_device(d, c))
                raise AssertionError("expected LogicError: unsupported scheme")
            except S.LogicError:
                return
        assert dec.invariant_noise_budget(_to_device(d, c)) == ref.noise_budget(c), ("invariant_noise_budget", what)
    budget(a, "fresh")
    # product (size 3: needs s^2), relinearized, at the next level, squared again (size 3 at a lower level)
    prod = ref.multiply_inplace(a.copy(), b)
    _same_plain(dec.decrypt(_to_device(d, prod)), ref.decrypt(prod), "size 3")
    budget(prod, "size 3")
    ref.relinearize_inplace(prod)
    _same_plain(dec.decrypt(_to_device(d, prod)), ref.decrypt(prod), "relinearized")
    budget(prod, "relinearized")
    if len(primes) > 2:
        nxt = ref.rescale_to_next_inplace(prod.copy()) if scheme == "ckks" else ref.mod_switch_to_next_inplace(prod.copy())
        _same_plain(dec.decrypt(_to_device(d, nxt)), ref.decrypt(nxt), "next level")
        budget(nxt, "next level")
        if scheme == "bgv":
            assert nxt.info()["correction_factor"] != 1  # exercises the correction-factor fix of bgv_decrypt
    # a product of products without relinearization: size 5 (s^4)
    if scheme != "ckks":
        big = ref.multiply_inplace(ref.multiply_inplace(a.copy(), b), ref.multiply_inplace(a.copy(), a))
        assert big.info()["size"] == 5
        _same_plain(dec.decrypt(_to_device(d, big)), ref.decrypt(big), "size 5")
        budget(big, "size 5 (exhausted or nearly)")
    # the device's own evaluation decrypts to what the reference's evaluation decrypts to
    ca, cb = _to_device(d, a), _to_device(d, b)
    d.ev.multiply_inplace(ca, cb)
    _same_plain(dec.decrypt(ca), ref.decrypt(ref.multiply_inplace(a.copy(), b)), "device product")
    # batches: every item, untrimmed
    items = [fresh() for _ in range(batch)]
    arr = np.stack([c.data() for c in items], axis=1)
    i0 = items[0].info()
    cts = S.Ciphertext.from_numpy(d.ctx, arr, d.ctx.parms_id_at(i0["chain_index"]), i0["is_ntt_form"], i0["scale"], i0["correction_factor"])
    buf, words = dec.decrypt_batch(cts)
    out = buf.to_numpy((words,)).reshape(batch, -1)
    for k, c in enumerate(items):
        want = ref.decrypt(c).data()
        assert np.array_equal(out[k][: want.size], want) and not out[k][want.size:].any(), "batch item %d" % k
    # argument checks (decryptor.cpp:82-92, 119-122, 157-160)
    try:
        dec.decrypt(cts)
        raise AssertionError("expected InvalidArgument for a batch")
    except S.InvalidArgument:
        pass
    wrong = _to_device(d, a)
    wrong.set_is_ntt_form(not wrong.is_ntt_form())
    try:
        dec.decrypt(wrong)
        raise AssertionError("expected InvalidArgument for the wrong form")
    except S.InvalidArgument:
        pass
    empty = S.Ciphertext(d.ctx)
    try:
        dec.decrypt(empty)
        raise AssertionError("expected InvalidArgument for an empty ciphertext")
    except S.InvalidArgument:
        pass


def case_end_to_end_streams(scheme, n, bits):
    """the server flow on the device with nothing but the reference's byte streams on either side: keys + ciphertexts in,
    multiply / relinearize / (rescale | mod switch) / rotate, result stream out, and the client-side check: the REFERENCE
    decrypts our stream to the same plaintext as its own evaluation"""
    primes, t, ref, d, dec, _ = _setup(scheme, n, bits)
    rng = np.random.default_rng(23)
    if scheme == "ckks":
        a, b = (ref.ckks_encrypt(rng.standard_normal(n // 2), 2.0 ** 30) for _ in range(2))
    else:
        a, b = (ref.batch_encrypt(rng.integers(0, t, n, dtype=np.uint64)) for _ in range(2))
    rlk = S.RelinKeys(d.ctx)
    rlk.load_bytes(ref.keys_save("relin", True))
    elt = ref.galois_elt_from_step(1)
    glk = S.GaloisKeys(d.ctx)
    glk.load_bytes(ref.keys_save("galois", True, [elt]))
    ca, cb = S.Ciphertext(d.ctx), S.Ciphertext(d.ctx)
    ca.load_bytes(ref.ct_save(a))
    cb.load_bytes(ref.ct_save(b))
    d.ev.multiply_inplace(ca, cb)
    d.ev.relinearize_inplace(ca, rlk)
    if scheme == "ckks":
        d.ev.rescale_to_next_inplace(ca)
    else:
        d.ev.mod_switch_to_next_inplace(ca)
    d.ev.apply_galois_inplace(ca, elt, glk)
    stream = ca.save_bytes()
    # reference side
    r = ref.multiply_inplace(a.copy(), b)
    ref.relinearize_inplace(r)
    r = ref.rescale_to_next_inplace(r) if scheme == "ckks" else ref.mod_switch_to_next_inplace(r)
    ref.apply_galois_inplace(r, elt)
    assert stream == ref.ct_save(r), "result stream"
    back, _ = ref.ct_load(stream)
    assert np.array_equal(ref.decrypt(back).data(), ref.decrypt(r).data())
    _same_plain(dec.decrypt(ca), ref.decrypt(r), "device decrypt of the device result")


def case_encrypt_symmetric(scheme, n, bits, seed=0x5EA1):
    """secret-key encryption with the reference's randomness (same seeded Blake2xb factory on both sides): the zero encryption
    and the encryption of an encoded plaintext give the reference's bytes, as a full ciphertext and as the seeded stream, at every
    level; with operating-system entropy the ciphertext still decrypts to the plaintext on the reference side"""
    primes, t, ref, d, dec, _ = _setup(scheme, n, bits)   # RefContext's factory is Blake2xbPRNGFactory({seed, 0, ...})
    enc = S.Encryptor(d.ctx, S.SecretKey(d.ctx, ref.secret_key()), seed=seed)
    for ci in range(ref.first_chain_index, -1, -1):
        pid = d.ctx.parms_id_at(ci)
        assert enc.encrypt_zero_symmetric_save(pid) == ref.encrypt_zero_symmetric_save(ci, True), ("seeded zero", ci)
        ct = enc.encrypt_zero_symmetric(pid)
        assert ct.save_bytes() == ref.encrypt_zero_symmetric_save(ci, False), ("full zero", ci)
    # the form without a parms_id works at the first data level (Encryptor_EncryptZeroSymmetric2, c/encryptor.h:34)
    assert enc.encrypt_zero_symmetric().save_bytes() == ref.encrypt_zero_symmetric_save(ref.first_chain_index, False)
    rng = np.random.default_rng(29)
    if scheme == "ckks":
        rpt = ref.ckks_encode(rng.standard_normal(n // 2), max(ref.first_chain_index - 1, 0), 2.0 ** 25)
    else:
        rpt = ref.batch_encode(rng.integers(0, t, n, dtype=np.uint64))
    pt = S.Plaintext(d.ctx)
    pt.load_bytes(ref.pt_save(rpt))
    assert enc.encrypt_symmetric_save(pt) == ref.encrypt_symmetric_save(rpt, True), "seeded encryption of a plaintext"
    ct = enc.encrypt_symmetric(pt)
    assert ct.save_bytes() == ref.encrypt_symmetric_save(rpt, False), "encryption of a plaintext"
    _same_plain(dec.decrypt(ct), ref.decrypt(ref.ct_load(ct.save_bytes())[0]), "decrypt(encrypt(plain))")
    # fresh entropy: different bytes every time, same plaintext after the reference decrypts the stream
    enc.set_seed(None)
    s1, s2 = enc.encrypt_symmetric_save(pt), enc.encrypt_symmetric_save(pt)
    assert s1 != s2
    want = ref.decrypt(ref.ct_load(ref.encrypt_symmetric_save(rpt, True))[0]).data()
    if scheme != "ckks":
        for s in (s1, s2):
            assert np.array_equal(ref.decrypt(ref.ct_load(s)[0]).data(), want)
    # argument checks of Encryptor::encrypt_internal
    try:
        enc.encrypt_zero_symmetric((1, 2, 3, 4))
        raise AssertionError("expected InvalidArgument for an unknown parms_id")
    except S.InvalidArgument:
        pass


def case_batch_encoder(scheme, n, bits, batch=3):
    """BatchEncoder on the device: encode == the reference's plaintext words (unsigned and signed, short and full vectors),
    decode == its slot values; the whole client loop encode -> encrypt -> evaluate -> decrypt_batch -> decode_device stays on
    the device and gives (a * b + rot) modulo t slot by slot, as the reference's own pipeline does"""
    primes, t, ref, d, dec, _ = _setup(scheme, n, bits)
    be = S.BatchEncoder(d.ctx)
    assert be.slot_count() == n
    rng = np.random.default_rng(41)
    for count in (n, n // 2 + 3, 1, 0):
        vals = rng.integers(0, t, count, dtype=np.uint64)
        rpt = ref.batch_encode(vals) if count else ref.batch_encode(np.zeros(1, dtype=np.uint64))
        pt = be.encode(vals if count else np.zeros(0, dtype=np.uint64))
        assert pt.coeff_count() == n and not pt.is_ntt_form()
        assert np.array_equal(pt.to_numpy(), rpt.data()), ("encode", count)
        assert np.array_equal(be.decode(pt), ref.batch_decode(rpt)), ("decode", count)
    svals = rng.integers(-(t // 2), t // 2 + 1, n, dtype=np.int64)
    rpt = ref.batch_encode_signed(svals)
    pt = be.encode(svals, signed=True)
    assert np.array_equal(pt.to_numpy(), rpt.data()), "signed encode"
    assert np.array_equal(be.decode(pt, signed=True), ref.batch_decode(rpt, signed=True)), "signed decode"
    assert np.array_equal(be.decode(pt, signed=True), svals)
    # a trimmed plaintext (as Decryptor returns it) decodes like the reference's
    short = ref.pt(np.array([5, 0, 7], dtype=np.uint64))
    mine = S.Plaintext.from_numpy(d.ctx, np.array([5, 0, 7], dtype=np.uint64))
    assert np.array_equal(be.decode(mine), ref.batch_decode(short))
    # argument checks (batchencoder.cpp:131-145, 175-190, 363-371)
    for bad in (lambda: be.encode(np.full(3, t, dtype=np.uint64)), lambda: be.encode(np.zeros(n + 1, dtype=np.uint64)),
                lambda: be.encode(np.array([t // 2 + 1], dtype=np.int64), signed=True)):
        try:
            bad()
            raise AssertionError("expected InvalidArgument")
        except S.InvalidArgument:
            pass

    # the client loop on the device
    ref.keygen_relin()
    enc = S.Encryptor(d.ctx, S.SecretKey(d.ctx, ref.secret_key()))
    rlk = S.RelinKeys(d.ctx)
    rlk.load_bytes(ref.keys_save("relin", True))
    a = rng.integers(0, t, (batch, n), dtype=np.uint64)
    b = rng.integers(0, t, (batch, n), dtype=np.uint64)
    prods = S.Ciphertext(d.ctx, batch=batch)
    for k in range(batch):
        ca, cb = enc.encrypt_symmetric(be.encode(a[k])), enc.encrypt_symmetric(be.encode(b[k]))
        d.ev.multiply_inplace(ca, cb)
        d.ev.relinearize_inplace(ca, rlk)
        prods.load_bytes(ca.save_bytes(), item=k)
    coeffs, words = dec.decrypt_batch(prods)
    vals = be.decode_device(coeffs, batch).to_numpy((batch, n))
    want = (a.astype(object) * b.astype(object)) % t
    assert np.array_equal(vals, want.astype(np.uint64)), "slot-wise products modulo t"
    again = be.encode_device(S.DeviceBuffer.from_numpy(vals), batch).to_numpy((batch, n))
    assert np.array_equal(again, coeffs.to_numpy((batch, n))), "encode_device(decode_device(x)) == x"


def case_slot_semantics(scheme, n, bits):
    """an application-level check with a mathematical oracle (numpy on the slot vectors, no reference arithmetic involved):
    BatchEncoder -> secret-key encryption -> add / sub / multiply / relinearize / multiply_plain / add_plain / rotate_rows /
    rotate_columns / mod_switch_to_next on the device -> decrypt -> decode gives ((a*b + c) * p - a + q) rotated, modulo t,
    slot by slot.  Only the keys come from the reference's KeyGenerator (as serialized streams)."""
    primes, t, ref, d, dec, _ = _setup(scheme, n, bits)
    be = S.BatchEncoder(d.ctx)
    enc = S.Encryptor(d.ctx, S.SecretKey(d.ctx, ref.secret_key()))
    rlk = S.RelinKeys(d.ctx)
    rlk.load_bytes(ref.keys_save("relin", True))
    steps = (1, -2)
    elts = [ref.galois_elt_from_step(s) for s in steps] + [2 * n - 1]
    glk = S.GaloisKeys(d.ctx)
    glk.load_bytes(ref.keys_save("galois", True, elts))
    rng = np.random.default_rng(53)
    a, b, c, p, q = (rng.integers(0, t, n, dtype=np.uint64) for _ in range(5))
    ca, cb, cc = (enc.encrypt_symmetric(be.encode(v)) for v in (a, b, c))
    d.ev.multiply_inplace(ca, cb)                    # a*b
    d.ev.relinearize_inplace(ca, rlk)
    d.ev.add_inplace(ca, cc)                         # + c
    d.ev.multiply_plain_inplace(ca, be.encode(p))    # * p
    d.ev.sub_inplace(ca, enc.encrypt_symmetric(be.encode(a)))   # - a
    d.ev.add_plain_inplace(ca, be.encode(q))         # + q
    if len(primes) > 2:
        d.ev.mod_switch_to_next_inplace(ca)
    d.ev.rotate_rows_inplace(ca, 1, glk)
    d.ev.rotate_rows_inplace(ca, -2, glk)
    d.ev.rotate_columns_inplace(ca, glk)
    got = be.decode(dec.decrypt(ca))
    O = lambda v: v.astype(object)
    want = (((O(a) * O(b) + O(c)) * O(p) - O(a) + O(q)) % t).astype(np.uint64)
    m = want.reshape(2, n // 2)
    m = np.roll(m, -1, axis=1)       # rotate_rows(1): every row one slot to the left
    m = np.roll(m, 2, axis=1)        # rotate_rows(-2)
    m = m[::-1]                      # rotate_columns: swap the two rows
    assert np.array_equal(got, m.reshape(-1)), "slot semantics"


def case_encrypt_asymmetric(scheme, n, bits, seed=0x5EA1):
    """public-key encryption with the reference's randomness: zero encryptions at every data level (they go through the modulus
    switch from the level above) and the encryption of an encoded plaintext equal Encryptor(context, public_key)'s bytes; the
    PublicKey comes as raw words and as its serialized stream"""
    primes, t, ref, d, dec, _ = _setup(scheme, n, bits)
    pk_words = S.PublicKey(d.ctx, ref.public_key())
    pk_stream = S.PublicKey(d.ctx)
    assert pk_stream.load_bytes(ref.public_key_save()) > 0
    for pk in (pk_words, pk_stream):
        enc = S.Encryptor(d.ctx, public_key=pk, seed=seed)
        for ci in range(ref.first_chain_index, -1, -1):
            ct = enc.encrypt_zero(d.ctx.parms_id_at(ci))
            assert ct.save_bytes() == ref.encrypt_asymmetric_save(None, ci), ("encrypt_zero", ci)
        assert enc.encrypt_zero().save_bytes() == ref.encrypt_asymmetric_save(None, ref.first_chain_index)   # Encryptor_EncryptZero2
    rng = np.random.default_rng(37)
    if scheme == "ckks":
        rpt = ref.ckks_encode(rng.standard_normal(n // 2), ref.first_chain_index, 2.0 ** 25)
    else:
        rpt = ref.batch_encode(rng.integers(0, t, n, dtype=np.uint64))
    pt = S.Plaintext(d.ctx)
    pt.load_bytes(ref.pt_save(rpt))
    ct = enc.encrypt(pt)
    assert ct.save_bytes() == ref.encrypt_asymmetric_save(rpt), "encrypt(plain)"
    _same_plain(dec.decrypt(ct), ref.decrypt(ref.ct_load(ct.save_bytes())[0]), "decrypt(encrypt(plain))")
    # without a public key / secret key the respective calls are refused like the reference does (logic_error)
    only_sk = S.Encryptor(d.ctx, S.SecretKey(d.ctx, ref.secret_key()))
    for bad in (lambda: only_sk.encrypt(pt), lambda: enc.encrypt_symmetric(pt)):
        try:
            bad()
            raise AssertionError("expected LogicError")
        except S.LogicError:
            pass


def case_ckks_encoder(n, bits, check_bits=True):
    """CKKSEncoder on the device: the plaintext words of encode (real and complex vectors, short and full, small and > 64-bit
    scales, every level) and the doubles of decode equal the reference's BIT FOR BIT - the FFT, the rounding and the CRT
    composition repeat its IEEE operations in its order; the client loop encode -> encrypt -> multiply / relinearize / rescale
    -> decrypt -> decode returns a*b within the scheme's approximation"""
    primes, t, ref, d, dec, _ = _setup("ckks", n, bits)
    enc = S.CKKSEncoder(d.ctx)
    assert enc.slot_count() == n // 2
    rng = np.random.default_rng(71)
    total_bits = sum(bits[:-1])
    scales = [2.0 ** 30, 2.0 ** 20]
    if total_bits > 100:
        scales.append(2.0 ** 80)     # coefficients above 64 bits: the 128-bit decomposition path
    if total_bits > 180:
        scales.append(2.0 ** 150)    # above 128 bits: the multi-precision branch (ckks.h:624-672)
        scales.append(2.0 ** 131.5)
    for ci in range(ref.first_chain_index, -1, -1):
        pid = d.ctx.parms_id_at(ci)
        for scale in scales:
            if np.log2(scale) + 8 >= sum(bits[: ci + 1]):
                continue
            for vals in (rng.standard_normal(n // 2) * 10, rng.standard_normal(5), np.zeros(0),
                         rng.standard_normal(n // 2) + 1j * rng.standard_normal(n // 2), (rng.standard_normal(3) + 2j)):
                try:
                    if np.iscomplexobj(vals):
                        rpt = ref.ckks_encode_complex(vals, ci, scale)
                    else:
                        rpt = ref.ckks_encode(vals if vals.size else np.zeros(0), ci, scale)
                except sealref.RefError as err:
                    # e.g. "encoded values are too large" at a small level: the same exception class here
                    want_cls = {1: S.InvalidArgument, 2: S.LogicError}[err.code]
                    try:
                        enc.encode(vals, pid, scale)
                        raise AssertionError("the reference rejected this input (%s)" % err)
                    except want_cls:
                        continue
                pt = enc.encode(vals, pid, scale)
                assert pt.is_ntt_form() and pt.scale() == scale and pt.parms_id() == pid
                assert np.array_equal(pt.to_numpy(), rpt.data()), ("encode", ci, scale, vals.size)
                for cplx in (False, True):
                    got, want = enc.decode(pt, cplx), ref.ckks_decode(rpt, cplx)
                    if check_bits:
                        assert got.tobytes() == want.tobytes(), ("decode bits", ci, scale, vals.size, cplx)
                    assert np.array_equal(got, want)
    # the single-value overloads (ckks.cpp:72-250), incl. coefficients above 64 and above 128 bits
    pid = d.ctx.parms_id_at(ref.first_chain_index)
    for scale in scales:
        if np.log2(scale) + 8 >= total_bits:
            continue
        for v in (3.14159265, -0.4, 1234.5):
            try:
                want = ref.ckks_encode_value(v, ref.first_chain_index, scale).data()
            except sealref.RefError as err:
                want_cls = {1: S.InvalidArgument, 2: S.LogicError}[err.code]
                try:
                    enc.encode(v, pid, scale)
                    raise AssertionError("the reference rejected encode(%r, scale %r): %s" % (v, scale, err))
                except want_cls:
                    continue
            assert np.array_equal(enc.encode(v, pid, scale).to_numpy(), want), (v, scale)
        # one complex value in every slot (CKKSEncoder_Encode4; ckks.h:795-800 fills `slots` copies and encodes them)
        for v in (complex(0.5, -1.25), complex(-3.0, 2.0 ** -7)):
            try:
                want = ref.ckks_encode_complex(np.full(n // 2, v, dtype=np.complex128), ref.first_chain_index, scale).data()
            except sealref.RefError:
                continue
            assert np.array_equal(enc.encode(v, pid, scale).to_numpy(), want), (v, scale)
    # argument checks (ckks.h:463-509, 686-716)
    pid = d.ctx.parms_id_at(ref.first_chain_index)
    for bad in (lambda: enc.encode(np.zeros(n // 2 + 1), pid, 2.0 ** 20), lambda: enc.encode(np.ones(4), pid, 0.0),
                lambda: enc.encode(np.ones(4), pid, 2.0 ** 2000 if False else float("inf")), lambda: enc.encode(np.array([np.nan]), pid, 2.0 ** 20),
                lambda: enc.encode(np.ones(4), (1, 2, 3, 4), 2.0 ** 20), lambda: enc.encode(np.ones(4) * 1e300, pid, 2.0 ** 40)):
        try:
            bad()
            raise AssertionError("expected InvalidArgument")
        except S.InvalidArgument:
            pass
    # the client loop on the device
    ref.keygen_relin()
    e = S.Encryptor(d.ctx, S.SecretKey(d.ctx, ref.secret_key()))
    rlk = S.RelinKeys(d.ctx)
    rlk.load_bytes(ref.keys_save("relin", True))
    a, b = rng.standard_normal(n // 2), rng.standard_normal(n // 2)
    scale = 2.0 ** (bits[-2] if len(bits) > 2 else 12)   # about the prime the rescale divides by
    ca, cb = e.encrypt_symmetric(enc.encode(a, pid, scale)), e.encrypt_symmetric(enc.encode(b, pid, scale))
    d.ev.multiply_inplace(ca, cb)
    d.ev.relinearize_inplace(ca, rlk)
    if len(primes) > 2:
        d.ev.rescale_to_next_inplace(ca)
    got = enc.decode(dec.decrypt(ca))
    assert np.max(np.abs(got - a * b)) < 1e-2, np.max(np.abs(got - a * b))


def case_keygen(scheme, n, bits, seed=0x5EA1, elts=(3, 5)):
    """KeyGenerator on the device with the reference's seeded factory: the secret key, the public key, the relinearization key
    and Galois keys equal the reference KeyGenerator's word for word; keys installed by create_* act like the reference's
    (relinearize / apply_galois give the same ciphertext words); a generator built around an existing secret key continues
    from it; with operating-system entropy the keys differ per generator and still decrypt correctly."""
    primes = coeff_modulus_create(n, bits)
    t = plain_modulus_batching(n, 20) if scheme != "ckks" else 0
    ref = sealref.RefContext(scheme, n, primes, t, seed=seed)
    d = DeviceSide(scheme, n, primes, t)
    L, digits = len(primes), len(primes) - 1
    seed8 = np.array([seed, 0, 0, 0, 0, 0, 0, 0], dtype=np.uint64)
    kg = S.KeyGenerator(d.ctx, seed=seed8)
    assert np.array_equal(kg.secret_key().words(L, n), ref.secret_key()), "secret key"
    assert np.array_equal(kg.create_public_key().words(L, n), ref.public_key()), "public key"
    ref.keygen_relin()
    assert np.array_equal(kg.key_words(0, digits, L, n), ref.key("relin", 0)), "relin key"
    elts = [e for e in elts if e < 2 * n]
    elts.append(2 * n - 1)
    ref.keygen_galois_elts(elts)
    for e in elts:
        assert np.array_equal(kg.key_words(e, digits, L, n), ref.key("galois", (e - 1) >> 1)), ("galois key", e)
    # a generator around the existing secret key produces the same keys
    kg2 = S.KeyGenerator(d.ctx, secret_key=S.SecretKey(d.ctx, ref.secret_key()), seed=seed8)
    assert np.array_equal(kg2.key_words(0, digits, L, n), ref.key("relin", 0)), "relin key (existing secret key)"
    # installed keys in use: the same operations with reference-generated keys uploaded as words
    rlk, glk = kg.create_relin_keys(), kg.create_galois_keys(elts)
    rlk_ref, glk_ref = S.RelinKeys(d.ctx), S.GaloisKeys(d.ctx)
    rlk_ref.set_key(0, ref.key("relin", 0))
    for e in elts:
        assert glk.has_key(e)
        glk_ref.set_key((e - 1) >> 1, ref.key("galois", (e - 1) >> 1))
    enc = S.Encryptor(d.ctx, secret_key=kg.secret_key(), seed=seed8)
    a = enc.encrypt_zero_symmetric(d.ctx.first_parms_id())
    b = enc.encrypt_zero_symmetric(d.ctx.first_parms_id())
    if scheme == "ckks":
        a.scale = b.scale = 2.0 ** 20
    prod = a.copy()
    d.ev.multiply_inplace(prod, b)
    r1, r2 = prod.copy(), prod.copy()
    d.ev.relinearize_inplace(r1, rlk)
    d.ev.relinearize_inplace(r2, rlk_ref)
    assert np.array_equal(r1.to_numpy(), r2.to_numpy()), "relinearize"
    for e in elts:
        r1, r2 = a.copy(), a.copy()
        d.ev.apply_galois_inplace(r1, e, glk)
        d.ev.apply_galois_inplace(r2, e, glk_ref)
        assert np.array_equal(r1.to_numpy(), r2.to_numpy()), ("apply_galois", e)
    # the save_seed forms: the half-size streams equal Serializable<RelinKeys / GaloisKeys>::save byte for byte, and load back
    assert kg.save_seeded() == ref.keys_save("relin", seeded=True), "seeded RelinKeys stream"
    stream = kg.save_seeded(elts)
    assert stream == ref.keys_save("galois", seeded=True, elts=elts), "seeded GaloisKeys stream"
    back = S.GaloisKeys(d.ctx)
    assert back.load_bytes(stream) == len(stream) and all(back.has_key(e) for e in elts)
    # create_galois_keys(steps) and create_galois_keys(): the elements the reference's GaloisTool derives
    steps = [1, -2, 0]
    by_steps = kg.create_galois_keys(steps=steps)
    want = {ref.galois_elt_from_step(s) for s in steps}
    assert all(by_steps.has_key(e) for e in want) and by_steps.size() == len(want), "keys from steps"
    e1 = ref.galois_elt_from_step(1)
    ref.keygen_galois_elts([e1])
    glk_ref.set_key((e1 - 1) >> 1, ref.key("galois", (e1 - 1) >> 1))
    r1, r2 = a.copy(), a.copy()
    d.ev.apply_galois_inplace(r1, e1, by_steps)
    d.ev.apply_galois_inplace(r2, e1, glk_ref)
    assert np.array_equal(r1.to_numpy(), r2.to_numpy()), "rotation by one step"
    if n <= 4096:
        everything = kg.create_galois_keys()
        all_elts = ref.galois_elts_all()
        assert all(everything.has_key(e) for e in all_elts) and everything.size() == len(set(all_elts)), "all keys"
    # operating-system entropy: fresh keys each time, and a working set
    k1, k2 = S.KeyGenerator(d.ctx), S.KeyGenerator(d.ctx)
    assert not np.array_equal(k1.secret_key().words(L, n), k2.secret_key().words(L, n))
    if scheme != "ckks":
        be = S.BatchEncoder(d.ctx)
        vals = np.random.default_rng(5).integers(0, t, n, dtype=np.uint64)
        e1 = S.Encryptor(d.ctx, public_key=k1.create_public_key())
        ct = e1.encrypt(be.encode(vals))
        sq = ct.copy()
        d.ev.multiply_inplace(sq, ct)
        d.ev.relinearize_inplace(sq, k1.create_relin_keys())
        got = be.decode(S.Decryptor(d.ctx, k1.secret_key()).decrypt(sq))
        assert np.array_equal(np.asarray(got, dtype=np.uint64), vals * vals % np.uint64(t)), "fresh keys: square"
