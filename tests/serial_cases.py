"""Wire-format cases shared by the CPU (emulated kernels) and `-m gpu` suites: the reference's own byte
streams — Ciphertext::save / Serializable<Ciphertext> (seeded), RelinKeys / GaloisKeys (seeded or full) — produced by the
REAL reference (oracle/_ref) are loaded through the C ABI (Ciphertext_Load, KSwitchKeys_Load, ...) and must give the same
words, metadata, bytes back (Ciphertext_Save) and the same exception classes on malformed input.
TEST INFRASTRUCTURE: the reference is the checker here, never the thing tested."""
import struct

import numpy as np

import seal_amd as S
import sealref
from harness import DeviceSide
from oracle import coeff_modulus_create, plain_modulus_batching, rand_ct

# reference error code (oracle/sealref_shim.cpp) -> the exception class the C ABI's HRESULT maps to
CLASS_OF_CODE = {1: S.InvalidArgument, 2: S.LogicError, 3: S.OutOfRange, 4: S.DeviceError}


def setup(scheme, n, bits, tbits=20):
    primes = coeff_modulus_create(n, bits)
    t = plain_modulus_batching(n, tbits) if scheme != "ckks" else 0
    return primes, t, sealref.RefContext(scheme, n, primes, t), DeviceSide(scheme, n, primes, t)


def _same_ct(ct, rct, what):
    ri = rct.info()
    assert (ct.size(), ct.is_ntt_form(), ct.scale(), ct.correction_factor(), ct.coeff_modulus_size()) == \
        (ri["size"], ri["is_ntt_form"], ri["scale"], ri["correction_factor"], ri["coeff_modulus_size"]), what
    assert np.array_equal(ct.to_numpy()[:, 0], rct.data()), what


def case_parms_ids(scheme, n, bits):
    """every level's parms_id is the reference's BLAKE2b hash (encryptionparams.cpp:117-147)"""
    primes, t, ref, d = setup(scheme, n, bits)
    for ci in range(ref.key_chain_index + 1):
        assert d.ctx.parms_id_at(ci) == ref.parms_id(ci), (scheme, ci)


def case_ciphertext_streams(scheme, n, bits):
    """seeded and full streams at every data level: load == the reference's load, save == the reference's bytes"""
    primes, t, ref, d = setup(scheme, n, bits)
    for seeded in (True, False):
        for ci in range(ref.first_chain_index, -1, -1):
            data = ref.encrypt_zero_symmetric_save(ci, seeded)
            rct, nbytes = ref.ct_load(data)
            ct = S.Ciphertext(d.ctx)
            assert ct.load_bytes(data) == nbytes == len(data)
            _same_ct(ct, rct, (scheme, seeded, ci))
            assert ct.save_size() == len(ref.ct_save(rct))
            assert ct.save_bytes() == ref.ct_save(rct), "Ciphertext::save bytes"
            # trailing bytes after the object are left alone (concatenated objects)
            ct2 = S.Ciphertext(d.ctx)
            assert ct2.load_bytes(data + b"\x00" * 24, unsafe=True) == len(data)
            _same_ct(ct2, rct, "unsafe_load with trailing bytes")


def case_evaluated_ciphertext_roundtrip(scheme, n, bits):
    """a size-3 product computed on the device is saved, loaded by the REFERENCE, and equals the reference's own product;
    the reference's saved product loads back into the device slab"""
    primes, t, ref, d = setup(scheme, n, bits)
    ref.keygen_relin()
    K = len(primes) - 1
    rng = np.random.default_rng(11)
    x, y = rand_ct(rng, primes, K, n), rand_ct(rng, primes, K, n)
    is_ntt = scheme != "bfv"
    scale = 2.0 ** 10 if scheme == "ckks" else 1.0
    cx, cy = d.ct(x, scale=scale, is_ntt=is_ntt), d.ct(y, scale=scale, is_ntt=is_ntt)
    d.ev.multiply_inplace(cx, cy)
    stream = cx.save_bytes()
    rct, nbytes = ref.ct_load(stream)
    assert nbytes == len(stream)
    rx = ref.ct(ref.first_chain_index, x, is_ntt, scale)
    ry = ref.ct(ref.first_chain_index, y, is_ntt, scale)
    ref.multiply_inplace(rx, ry)
    assert np.array_equal(rct.data(), rx.data()) and rct.info() == rx.info()
    back = S.Ciphertext(d.ctx)
    back.load_bytes(ref.ct_save(rx))
    _same_ct(back, rx, "reference product loaded to the device")


def case_bgv_coefficient_form_stream(n, bits):
    """a BGV ciphertext serialized in coefficient form is transformed on load (ciphertext.cpp:384-403)"""
    primes, t, ref, d = setup("bgv", n, bits)
    data = ref.encrypt_zero_symmetric_save(ref.first_chain_index, False)
    rct, _ = ref.ct_load(data)
    ref.transform_from_ntt_inplace(rct)
    stream = ref.ct_save(rct)
    assert stream[16 + 32] == 0  # is_ntt_form byte
    loaded, _ = ref.ct_load(stream)
    assert loaded.info()["is_ntt_form"]
    ct = S.Ciphertext(d.ctx)
    ct.load_bytes(stream)
    _same_ct(ct, loaded, "bgv coefficient-form stream")
    # out-of-range coefficients are rejected before the transform, for unsafe_load as well
    bad = bytearray(stream)
    off = 16 + 32 + 1 + 40 + 16 + 8  # first data word
    bad[off:off + 8] = struct.pack("<Q", primes[0])
    for unsafe in (False, True):
        _same_failure(ref, d, bytes(bad), unsafe)


def case_batch_items(scheme, n, bits):
    """LoadItem / SaveItem: slots of a device-resident batch"""
    primes, t, ref, d = setup(scheme, n, bits)
    streams = [ref.encrypt_zero_symmetric_save(ref.first_chain_index, s) for s in (True, False, True)]
    refs = [ref.ct_load(s)[0] for s in streams]
    ct = S.Ciphertext(d.ctx, batch=3)
    for i, s in enumerate(streams):
        assert ct.load_bytes(s, item=i) == len(s)
    arr = ct.to_numpy()
    for i, r in enumerate(refs):
        assert np.array_equal(arr[:, i], r.data())
        assert ct.save_bytes(item=i) == ref.ct_save(r)
    # an item at another level does not fit the batch
    if ref.first_chain_index > 0:
        other = ref.encrypt_zero_symmetric_save(ref.first_chain_index - 1, False)
        try:
            ct.load_bytes(other, item=1)
            raise AssertionError("expected InvalidArgument")
        except S.InvalidArgument:
            pass
    try:
        ct.load_bytes(streams[0], item=3)
        raise AssertionError("expected OutOfRange")
    except S.OutOfRange:
        pass


def case_key_streams(scheme, n, bits, seeded, steps=(1,)):
    """serialized RelinKeys / GaloisKeys (seeded = the Serializable<> form) loaded straight into the device key slabs:
    relinearize / apply_galois with them give the reference's words"""
    primes, t, ref, d = setup(scheme, n, bits)
    K = len(primes) - 1
    rng = np.random.default_rng(5)
    is_ntt = scheme != "bfv"
    scale = 2.0 ** 10 if scheme == "ckks" else 1.0
    stream = ref.keys_save("relin", seeded)
    rlk = S.RelinKeys(d.ctx)
    assert rlk.load_bytes(stream) == len(stream)
    assert rlk.has_key(2) and rlk.size() == 1
    x3 = rand_ct(rng, primes, K, n, size=3)
    cx = d.ct(x3, scale=scale, is_ntt=is_ntt)
    d.ev.relinearize_inplace(cx, rlk)
    rx = ref.ct(ref.first_chain_index, x3, is_ntt, scale)
    ref.relinearize_inplace(rx)
    assert np.array_equal(cx.to_numpy()[:, 0], rx.data()), "relinearize with keys loaded from the stream"

    elts = [ref.galois_elt_from_step(s) for s in steps] + [2 * n - 1]
    stream = ref.keys_save("galois", seeded, elts)
    glk = S.GaloisKeys(d.ctx)
    assert glk.load_bytes(stream, unsafe=not seeded) == len(stream)
    assert glk.size() == len(set(elts))
    for e in elts:
        assert glk.has_key(e)
        x2 = rand_ct(rng, primes, K, n)
        cx = d.ct(x2, scale=scale, is_ntt=is_ntt)
        d.ev.apply_galois_inplace(cx, e, glk)
        rx = ref.ct(ref.first_chain_index, x2, is_ntt, scale)
        ref.apply_galois_inplace(rx, e)
        assert np.array_equal(cx.to_numpy()[:, 0], rx.data()), "apply_galois(%d) with keys loaded from the stream" % e
    # a load REPLACES the object (KSwitchKeys::load, kswitchkeys.cpp:92-180): indices absent from the new stream are gone
    only_first = ref.keys_save("galois", seeded, elts[:1])
    assert glk.load_bytes(only_first, unsafe=not seeded) == len(only_first)
    assert glk.size() == 1 and glk.has_key(elts[0])
    for e in set(elts[1:]) - {elts[0]}:
        assert not glk.has_key(e), "key %d survived a load that does not contain it" % e


def case_key_save(scheme, n, bits, steps=(1,)):
    """KSwitchKeys::save (c/kswitchkeys.h:41-43): keys loaded from the reference's streams (full and seeded) and keys made by
    the device KeyGenerator are written back as the reference's own bytes - the device slabs return from the key-switch
    kernels' register order (doubles / Shoup pairs) to canonical words - and the compressed forms load on both sides"""
    primes, t, ref, d = setup(scheme, n, bits)
    full = ref.keys_save("relin", False)
    rlk = S.RelinKeys(d.ctx)
    assert rlk.load_bytes(full) == len(full)
    assert rlk.save_bytes() == full, "full RelinKeys stream: load then save"
    seeded = ref.keys_save("relin", True)          # the reference's object now holds this stream's (expanded) keys
    assert rlk.load_bytes(seeded) == len(seeded)
    assert rlk.save_bytes() == ref.keys_save_mode("relin", 0), "seeded RelinKeys stream saved in full"
    elts = [ref.galois_elt_from_step(s) for s in steps] + [2 * n - 1]
    seeded = ref.keys_save("galois", True, elts)
    want = ref.keys_save_mode("galois", 0)
    glk = S.GaloisKeys(d.ctx)
    assert glk.load_bytes(seeded) == len(seeded)
    mine = glk.save_bytes()
    assert mine == want, "GaloisKeys (every slot, the empty ones included)"
    for mode in (1, 2):
        z = glk.save_bytes(compr_mode=mode)
        assert z[5] == mode and len(z) < len(want)
        if mode == 1:  # (the reference build here has zlib only: see case_compressed_streams)
            assert ref.keys_load(z) == len(z), "the reference reads our compressed key stream"
        g2 = S.GaloisKeys(d.ctx)
        assert g2.load_bytes(z) == len(z) and g2.save_bytes() == want
    # keys generated on the device: the reference installs our stream and key-switches to our words
    kg = S.KeyGenerator(d.ctx)
    own = kg.create_relin_keys()
    stream = own.save_bytes()
    assert ref.keys_load(stream) == len(stream)
    ref.keys_install("relin", stream)
    K = len(primes) - 1
    rng = np.random.default_rng(11)
    is_ntt, scale = scheme != "bfv", (2.0 ** 10 if scheme == "ckks" else 1.0)
    x3 = rand_ct(rng, primes, K, n, size=3)
    cx = d.ct(x3, scale=scale, is_ntt=is_ntt)
    d.ev.relinearize_inplace(cx, own)
    rx = ref.ct(ref.first_chain_index, x3, is_ntt, scale)
    ref.relinearize_inplace(rx)
    assert np.array_equal(cx.to_numpy()[:, 0], rx.data()), "the reference relinearizes with the keys we saved"
    own_g = kg.create_galois_keys(galois_elts=elts)
    stream = own_g.save_bytes()
    assert struct.unpack_from("<Q", stream, 48)[0] == n, "GaloisKeys::data() has a slot per odd element"
    ref.keys_install("galois", stream)
    x2 = rand_ct(rng, primes, K, n)
    cx = d.ct(x2, scale=scale, is_ntt=is_ntt)
    d.ev.apply_galois_inplace(cx, elts[0], own_g)
    rx = ref.ct(ref.first_chain_index, x2, is_ntt, scale)
    ref.apply_galois_inplace(rx, elts[0])
    assert np.array_equal(cx.to_numpy()[:, 0], rx.data()), "the reference rotates with the Galois keys we saved"
    # an empty object saves as the reference's empty object does
    empty = S.RelinKeys(d.ctx)
    e = empty.save_bytes()
    assert len(e) == 16 + 32 + 8 and struct.unpack_from("<Q", e, 48)[0] == 0
    # refusals: a buffer that is too small, an unknown compression mode, a digit-parallel slice (one rank's digits only)
    import ctypes as C
    from seal_amd import _native as N
    small, nb = (C.c_uint8 * 64)(), C.c_int64()
    assert _outcome(lambda: N.check(N.lib().KSwitchKeys_Save(own._h, small, C.c_uint64(64), C.c_uint8(0), C.byref(nb)))) == S.InvalidArgument
    assert _outcome(lambda: own.save_bytes(compr_mode=9)) == S.InvalidArgument
    if K >= 2:
        words = np.zeros((1, 2, len(primes), n), dtype=np.uint64)
        part = S.RelinKeys(d.ctx)
        part.set_key_digits(0, 1, words)          # digit 1 only
        assert _outcome(lambda: part.save_bytes()) == S.LogicError


def _walk_key_digits(stream):
    """offsets (start, size) of every digit (a framed seeded or full ciphertext) inside a KSwitchKeys stream"""
    pos = 16 + 32
    dim1 = struct.unpack_from("<Q", stream, pos)[0]
    pos += 8
    out = []
    for _ in range(dim1):
        dim2 = struct.unpack_from("<Q", stream, pos)[0]
        pos += 8
        for _ in range(dim2):
            size = struct.unpack_from("<Q", stream, pos + 8)[0]
            out.append((pos, size))
            pos += size
    assert pos == len(stream)
    return out


def case_shake256_seeded_streams(scheme, n, bits):
    """seeded objects whose seed names the reference's OTHER generator (prng_type 2 = Shake256PRNG, randomgen.cpp:216-227): the
    c_1 halves are expanded by the device SHAKE256 kernel when the polynomial is whole PRNG buffers - a ciphertext and every
    digit of a relinearization key; loaded words and the key-switching result equal the reference's"""
    primes, t, ref, d = setup(scheme, n, bits)
    K = len(primes) - 1
    rng = np.random.default_rng(15)
    seeded = bytearray(ref.encrypt_zero_symmetric_save(ref.first_chain_index, True))
    assert seeded[len(seeded) - 65] == 1
    seeded[len(seeded) - 65] = 2
    seeded = bytes(seeded)
    rct, _ = ref.ct_load(seeded)
    ct = S.Ciphertext(d.ctx)
    assert ct.load_bytes(seeded) == len(seeded)
    _same_ct(ct, rct, "shake256-seeded ciphertext")
    kstream = bytearray(ref.keys_save("relin", True))
    digits = _walk_key_digits(bytes(kstream))
    assert len(digits) == K
    for start, size in digits:
        assert kstream[start + size - 65] == 1
        kstream[start + size - 65] = 2
    kstream = bytes(kstream)
    assert ref.keys_install("relin", kstream) == len(kstream)   # the context's relinearization key is now the SHAKE256 expansion
    rlk = S.RelinKeys(d.ctx)
    assert rlk.load_bytes(kstream) == len(kstream)
    is_ntt, scale = scheme != "bfv", (2.0 ** 10 if scheme == "ckks" else 1.0)
    x3 = rand_ct(rng, primes, K, n, size=3)
    cx = d.ct(x3, scale=scale, is_ntt=is_ntt)
    d.ev.relinearize_inplace(cx, rlk)
    rx = ref.ct(ref.first_chain_index, x3, is_ntt, scale)
    ref.relinearize_inplace(rx)
    assert np.array_equal(cx.to_numpy()[:, 0], rx.data()), "relinearize with SHAKE256-seeded keys"
    # a seeded PublicKey stream (Serializable<PublicKey>): c_1 expanded on the device, both generators
    for prng_type in (1, 2):
        pstream = bytearray(ref.public_key_save_seeded())
        assert pstream[len(pstream) - 65] == 1 and len(pstream) < len(primes) * n * 8 + 4096
        pstream[len(pstream) - 65] = prng_type
        pstream = bytes(pstream)
        want = ref.public_key_load_words(pstream)
        pk = S.PublicKey(d.ctx)
        assert pk.load_bytes(pstream) == len(pstream)
        assert np.array_equal(pk.words(len(primes), n), want), "seeded PublicKey stream (prng_type %d)" % prng_type


def case_plaintext_streams(scheme, n, bits):
    """encoded plaintexts as a client serializes them (CKKSEncoder / BatchEncoder output): load == the reference's load,
    save == its bytes, multiply_plain with the loaded plaintext == the reference's, malformed streams fail alike"""
    primes, t, ref, d = setup(scheme, n, bits)
    K = len(primes) - 1
    rng = np.random.default_rng(9)
    if scheme == "ckks":
        rpt = ref.ckks_encode(rng.standard_normal(n // 2), ref.first_chain_index, 2.0 ** 30)
    else:
        rpt = ref.batch_encode(rng.integers(0, t, n, dtype=np.uint64))
    stream = ref.pt_save(rpt)
    back, nbytes = ref.pt_load(stream)
    pt = S.Plaintext(d.ctx)
    assert pt.load_bytes(stream) == nbytes == len(stream)
    bi = back.info()
    assert (pt.coeff_count(), pt.is_ntt_form(), pt.scale()) == (bi["coeff_count"], bi["is_ntt_form"], bi["scale"])
    if bi["is_ntt_form"]:
        assert pt.parms_id() == ref.parms_id(bi["chain_index"])
    assert np.array_equal(pt.to_numpy(), back.data())
    assert pt.save_bytes() == stream
    # used as an operand
    x = rand_ct(rng, primes, K, n)
    is_ntt = scheme != "bfv"
    scale = 2.0 ** 20 if scheme == "ckks" else 1.0
    cx = d.ct(x, scale=scale, is_ntt=is_ntt)
    d.ev.multiply_plain_inplace(cx, pt)
    rx = ref.ct(ref.first_chain_index, x, is_ntt, scale)
    ref.multiply_plain_inplace(rx, back)
    assert np.array_equal(cx.to_numpy()[:, 0], rx.data()) and cx.scale() == rx.info()["scale"]

    def mutate(base, off, fmt, value):
        b = bytearray(base)
        b[off:off + struct.calcsize(fmt)] = struct.pack(fmt, value)
        return bytes(b)

    data0 = 16 + 32 + 8 + 8 + 16 + 8
    cases = {
        "ok": stream, "truncated": stream[:-8], "bad_magic": mutate(stream, 0, "<H", 3), "zstd": mutate(stream, 5, "<B", 2),
        "unknown_parms_id": mutate(stream, 16, "<Q", 77), "zero_parms_id": stream[:16] + b"\x00" * 32 + stream[48:],
        "count_plus_1": mutate(stream, 48, "<Q", bi["coeff_count"] + 1), "count_huge": mutate(stream, 48, "<Q", 2 ** 40),
        "scale_nan": mutate(stream, 56, "<d", float("nan")), "scale_negative": mutate(stream, 56, "<d", -1.0),
        "dyn_count_small": mutate(stream, data0 - 8, "<Q", bi["coeff_count"] - 1),
        "coefficient_max": mutate(stream, data0 + 8 * 5, "<Q", 2 ** 64 - 1),
        "coefficient_eq_modulus": mutate(stream, data0, "<Q", primes[0] if bi["is_ntt_form"] else t),
    }
    for name, data in cases.items():
        for unsafe in (False, True):
            if name == "zstd" and _zstd() is not None:
                continue  # see case_malformed_streams: the reference build has no zstd
            want = _outcome(lambda: ref.pt_load(data, unsafe))
            got = _outcome(lambda: S.Plaintext(d.ctx).load_bytes(data, unsafe=unsafe))
            assert got == want, "%s (unsafe=%s): got %r, reference %r" % (name, unsafe, got, want)


def _outcome(fn):
    try:
        fn()
        return None
    except sealref.RefError as e:
        return CLASS_OF_CODE[e.code]
    except S.SealHipError as e:
        return type(e)


def _zstd():
    """libzstd through ctypes (test side): (compress, decompress) or None when the library is absent"""
    import ctypes as C
    try:
        z = C.CDLL("libzstd.so.1")
    except OSError:
        return None
    z.ZSTD_compressBound.restype = C.c_size_t
    z.ZSTD_compress.restype = C.c_size_t
    z.ZSTD_decompress.restype = C.c_size_t

    def compress(raw):
        cap = z.ZSTD_compressBound(C.c_size_t(len(raw)))
        buf = (C.c_uint8 * cap)()
        n = z.ZSTD_compress(buf, C.c_size_t(cap), raw, C.c_size_t(len(raw)), C.c_int(3))
        assert not z.ZSTD_isError(C.c_size_t(n))
        return bytes(buf[:n])

    def decompress(data, raw_len):
        buf = (C.c_uint8 * raw_len)()
        n = z.ZSTD_decompress(buf, C.c_size_t(raw_len), data, C.c_size_t(len(data)))
        assert n == raw_len, n
        return bytes(buf)
    return compress, decompress


def _recompress(stream, mode, compress):
    """an uncompressed SEAL stream -> the same object with its member bytes compressed (header in the clear)"""
    payload = compress(stream[16:])
    return stream[:5] + bytes([mode]) + stream[6:8] + struct.pack("<Q", 16 + len(payload)) + payload


def case_compressed_streams(scheme, n, bits):
    """compr_mode zlib (checked against the reference, which is built with the system zlib) and zstd (the reference's default
    in stock builds; checked against libzstd itself): compressed ciphertext / plaintext / key streams load to the same words,
    the compressed streams we save are loaded by the reference, truncated or corrupt payloads fail like the reference's"""
    import zlib
    primes, t, ref, d = setup(scheme, n, bits)
    K = len(primes) - 1
    rng = np.random.default_rng(61)
    data = ref.encrypt_zero_symmetric_save(ref.first_chain_index, False)
    rct, _ = ref.ct_load(data)
    seeded = ref.encrypt_zero_symmetric_save(ref.first_chain_index, True)
    # --- zlib: the reference writes, we read
    for stream in (ref.ct_save_mode(rct, 1), _recompress(seeded, 1, zlib.compress)):
        assert stream[5] == 1
        want, nb = ref.ct_load(stream)
        ct = S.Ciphertext(d.ctx)
        assert ct.load_bytes(stream) == nb == len(stream)
        _same_ct(ct, want, "zlib stream")
    # --- zlib: we write, the reference reads
    ct = S.Ciphertext(d.ctx)
    ct.load_bytes(data)
    mine = ct.save_bytes(compr_mode=1)
    assert mine[5] == 1 and len(mine) <= ct.save_size(1)
    back, nb = ref.ct_load(mine)
    assert nb == len(mine) and np.array_equal(back.data(), rct.data()) and back.info() == rct.info()
    assert zlib.decompress(mine[16:]) == data[16:], "the compressed payload is exactly the member bytes"
    # plaintexts and keys
    rpt = ref.ckks_encode(rng.standard_normal(n // 2), ref.first_chain_index, 2.0 ** 30) if scheme == "ckks" else ref.batch_encode(rng.integers(0, t, n, dtype=np.uint64))
    pz = ref.pt_save_mode(rpt, 1)
    pt = S.Plaintext(d.ctx)
    assert pt.load_bytes(pz) == len(pz) and np.array_equal(pt.to_numpy(), rpt.data())
    pback, _ = ref.pt_load(pt.save_bytes(compr_mode=1))
    assert np.array_equal(pback.data(), rpt.data())
    ref.keys_save("relin", True)                       # the context's key object now equals this stream
    kz = ref.keys_save_mode("relin", 1)
    assert kz[5] == 1
    rlk = S.RelinKeys(d.ctx)
    assert rlk.load_bytes(kz) == len(kz)
    x3 = rand_ct(rng, primes, K, n, size=3)
    is_ntt, scale = scheme != "bfv", (2.0 ** 10 if scheme == "ckks" else 1.0)
    cx = d.ct(x3, scale=scale, is_ntt=is_ntt)
    d.ev.relinearize_inplace(cx, rlk)
    rx = ref.ct(ref.first_chain_index, x3, is_ntt, scale)
    ref.relinearize_inplace(rx)
    assert np.array_equal(cx.to_numpy()[:, 0], rx.data()), "relinearize with keys from a zlib stream"
    # corrupt / truncated zlib payloads: the reference's classes
    good = ref.ct_save_mode(rct, 1)
    for name, bad in (("truncated", good[:-7]), ("size_field_short", good[:8] + struct.pack("<Q", len(good) - 9) + good[16:]),
                      ("flipped", good[:40] + bytes([good[40] ^ 0x55]) + good[41:]), ("not_deflate", data[:5] + b"\x01" + data[6:])):
        for unsafe in (False, True):
            try:
                _same_failure(ref, d, bad, unsafe)
            except AssertionError as e:
                raise AssertionError("zlib %s (unsafe=%s): %s" % (name, unsafe, e))
    # nested compressed objects inside a compressed outer object: the DynArray's own SEALHeader claims zlib and a
    # size far beyond the inflated buffer, followed by a valid inner zlib stream; and the legitimate form of the same nesting
    members_off, dyn_off = 16, 16 + 32 + 1 + 40
    inner_raw = data[dyn_off + 16:]
    inner_z = zlib.compress(inner_raw)
    def nested(claimed):
        nested_dyn = data[dyn_off:dyn_off + 5] + b"\x01" + data[dyn_off + 6:dyn_off + 8] + struct.pack("<Q", claimed) + inner_z
        return _recompress(data[:dyn_off] + nested_dyn, 1, zlib.compress)
    for unsafe in (False, True):
        # the legitimate nesting loads on both sides
        assert _same_failure(ref, d, nested(16 + len(inner_z)), unsafe) is None
        # a nested size beyond the inflated buffer: "I/O error" here.  (The reference build in this image does not survive
        # these two streams - an ios failure escapes its loader and std::terminate ends the process - so there is no class
        # to compare with; what matters is that nothing is read past the inflated buffer.)
        for claimed in (1 << 44, 16 + len(inner_z) + 4096):
            assert _outcome(lambda: S.Ciphertext(d.ctx).load_bytes(nested(claimed), unsafe=unsafe)) == S.DeviceError
    # --- zstd (libzstd on both sides of the check; the reference build has no zstd)
    zs = _zstd()
    if zs is None:
        return
    compress, decompress = zs
    for stream, want in ((_recompress(data, 2, compress), rct), (_recompress(seeded, 2, compress), ref.ct_load(seeded)[0])):
        ct = S.Ciphertext(d.ctx)
        assert ct.load_bytes(stream) == len(stream)
        _same_ct(ct, want, "zstd stream")
    ct = S.Ciphertext(d.ctx)
    ct.load_bytes(data)
    mine = ct.save_bytes(compr_mode=2)
    assert mine[5] == 2 and decompress(mine[16:], len(data) - 16) == data[16:]
    again = S.Ciphertext(d.ctx)
    assert again.load_bytes(mine) == len(mine)
    _same_ct(again, rct, "zstd round trip")
    kstream = ref.keys_save("relin", True)
    rlk2 = S.RelinKeys(d.ctx)
    assert rlk2.load_bytes(_recompress(kstream, 2, compress)) > 0
    cx = d.ct(x3, scale=scale, is_ntt=is_ntt)
    d.ev.relinearize_inplace(cx, rlk2)
    rx = ref.ct(ref.first_chain_index, x3, is_ntt, scale)
    ref.relinearize_inplace(rx)
    assert np.array_equal(cx.to_numpy()[:, 0], rx.data()), "relinearize with keys from a zstd stream"
    bad = _recompress(data, 2, compress)
    for broken in (bad[:-5], bad[:30] + bytes([bad[30] ^ 0xFF]) + bad[31:]):
        assert _outcome(lambda: S.Ciphertext(d.ctx).load_bytes(broken)) in (S.DeviceError, S.LogicError, S.InvalidArgument)


def _same_failure(ref, d, data, unsafe, keys=False):
    if keys:
        want = _outcome(lambda: ref.keys_load(data, unsafe))
        got = _outcome(lambda: S.KSwitchKeys(d.ctx).load_bytes(data, unsafe=unsafe))
    else:
        want = _outcome(lambda: ref.ct_load(data, unsafe))
        got = _outcome(lambda: S.Ciphertext(d.ctx).load_bytes(data, unsafe=unsafe))
    assert got == want, "exception class differs from the reference's: got %r, reference %r" % (got, want)
    return want


def case_malformed_streams(scheme, n, bits):
    """every mutation fails (or not) with the reference's exception class (Serialization::Load, serialization.cpp:341-553;
    Ciphertext::load_members; valcheck.cpp)"""
    primes, t, ref, d = setup(scheme, n, bits)
    full = ref.encrypt_zero_symmetric_save(ref.first_chain_index, False)
    seeded = ref.encrypt_zero_symmetric_save(ref.first_chain_index, True)
    pure_key = None
    members = 16  # offset of parms_id
    dyn = members + 32 + 1 + 40  # offset of the DynArray's SEALHeader
    K, nn = len(primes) - 1, n

    def mutate(base, off, fmt, value):
        b = bytearray(base)
        b[off:off + struct.calcsize(fmt)] = struct.pack(fmt, value)
        return bytes(b)

    cases = {
        "ok_full": full, "ok_seeded": seeded,
        "empty": b"", "short_header": full[:10], "truncated_members": full[:60], "truncated_data": full[:-8],
        "truncated_seed": seeded[:-4],
        "bad_magic": mutate(full, 0, "<H", 0x1234), "bad_header_size": mutate(full, 2, "<B", 0x20),
        "future_major": mutate(full, 3, "<B", 5), "future_minor": mutate(full, 4, "<B", 200), "old_3_3": mutate(mutate(full, 3, "<B", 3), 4, "<B", 3),
        "zlib_mode": mutate(full, 5, "<B", 1), "zstd_mode": mutate(full, 5, "<B", 2), "unknown_mode": mutate(full, 5, "<B", 9),
        "size_too_big": mutate(full, 8, "<Q", len(full) + 1), "size_too_small": mutate(full, 8, "<Q", len(full) - 8),
        "size_below_header": mutate(full, 8, "<Q", 8),
        "unknown_parms_id": mutate(full, members, "<Q", 12345),
        "size_1": mutate(full, members + 33, "<Q", 1), "size_7": mutate(full, members + 33, "<Q", 7), "size_3_short_data": mutate(full, members + 33, "<Q", 3),
        "wrong_degree": mutate(full, members + 41, "<Q", nn * 2), "wrong_K": mutate(full, members + 49, "<Q", K + 1),
        "scale_nan": mutate(full, members + 57, "<d", float("nan")), "scale_zero": mutate(full, members + 57, "<d", 0.0),
        "scale_two": mutate(full, members + 57, "<d", 2.0), "scale_negative": mutate(full, members + 57, "<d", -4.0),
        "correction_zero": mutate(full, members + 65, "<Q", 0), "correction_two": mutate(full, members + 65, "<Q", 2),
        "dyn_bad_magic": mutate(full, dyn, "<H", 0), "dyn_count_big": mutate(full, dyn + 16, "<Q", 2 * K * nn + 1),
        "dyn_count_small": mutate(full, dyn + 16, "<Q", 2 * K * nn - 1), "dyn_size_field": mutate(full, dyn + 8, "<Q", 24),
        "coefficient_out_of_range": mutate(full, dyn + 24, "<Q", primes[0]),
        "coefficient_max": mutate(full, dyn + 24 + 8 * (nn * K + 3), "<Q", 2 ** 64 - 1),
        "seed_prng_unknown": mutate(seeded, len(seeded) - 65, "<B", 0), "seed_prng_shake256": mutate(seeded, len(seeded) - 65, "<B", 2),
        "seed_prng_9": mutate(seeded, len(seeded) - 65, "<B", 9), "seed_header_bad": mutate(seeded, len(seeded) - 65 - 16, "<H", 1),
        "seeded_size_3": mutate(seeded, members + 33, "<Q", 3),
    }
    seen = set()
    for name, data in cases.items():
        for unsafe in (False, True):
            if name == "zstd_mode" and _zstd() is not None:
                # this reference build has no zstd and calls the header invalid; a library with zstd treats the payload as a
                # (broken) zstd stream, which is what the reference does with the zlib header above
                want = _outcome(lambda: ref.ct_load(cases["zlib_mode"], unsafe))
                assert _outcome(lambda: S.Ciphertext(d.ctx).load_bytes(data, unsafe=unsafe)) == want
                continue
            try:
                seen.add(_same_failure(ref, d, data, unsafe))
            except AssertionError as e:
                raise AssertionError("%s (unsafe=%s): %s" % (name, unsafe, e))
    # the SHAKE256 variant is not only accepted but expanded identically
    data = cases["seed_prng_shake256"]
    rct, _ = ref.ct_load(data)
    ct = S.Ciphertext(d.ctx)
    ct.load_bytes(data)
    _same_ct(ct, rct, "shake256-seeded stream")
    assert {None, S.InvalidArgument, S.LogicError} <= seen, seen
    # a key-level ciphertext (a PublicKey's members) is loadable only unsafely
    if pure_key is None and ref.key_chain_index > ref.first_chain_index:
        pk = ref.public_key_save()
        _same_failure(ref, d, pk, True)
        _same_failure(ref, d, pk, False)


def case_malformed_key_streams(scheme, n, bits):
    primes, t, ref, d = setup(scheme, n, bits)
    stream = ref.keys_save("relin", True)

    def mutate(base, off, fmt, value):
        b = bytearray(base)
        b[off:off + struct.calcsize(fmt)] = struct.pack(fmt, value)
        return bytes(b)

    first_key = 16 + 32 + 8 + 8  # SEALHeader of the first PublicKey
    cases = {
        "ok": stream, "truncated": stream[:-1], "bad_magic": mutate(stream, 0, "<H", 7),
        "wrong_parms_id": mutate(stream, 16, "<Q", 99),
        "dim1_huge": mutate(stream, 16 + 32, "<Q", n + 1), "dim2_huge": mutate(stream, 16 + 32 + 8, "<Q", len(primes)),
        "dim2_short": mutate(stream, 16 + 32 + 8, "<Q", len(primes) - 2),
        "key_not_ntt": mutate(stream, first_key + 16 + 32, "<B", 0),
        "key_size_3": mutate(stream, first_key + 16 + 33, "<Q", 3),
        "key_coefficient_out_of_range": mutate(stream, first_key + 16 + 73 + 24, "<Q", primes[0]),
    }
    for name, data in cases.items():
        for unsafe in (False, True):
            try:
                want = _same_failure(ref, d, data, unsafe, keys=True)
            except AssertionError as e:
                # documented difference: unsafe_load of structurally unusable keys fails at load time here, at first use in
                # the reference (the device layout needs key-level, NTT-form, size-2 digits)
                if unsafe and name in ("wrong_parms_id", "dim2_short", "key_not_ntt", "key_size_3"):
                    assert _outcome(lambda: S.KSwitchKeys(d.ctx).load_bytes(data, unsafe=True)) == S.LogicError
                    continue
                raise AssertionError("%s (unsafe=%s): %s" % (name, unsafe, e))
