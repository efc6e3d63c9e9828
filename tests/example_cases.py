"""The reference's own example programs restated on the device API ("real SEAL programs run on device-resident
data"): native/examples/5_ckks_basics.cpp (PI*x^3 + 0.4*x + 1 over 4096 points with rescaling and scale alignment) and the
batching part of native/examples/2_encoders.cpp / 6_rotation.cpp.  Only the keys come from the reference's KeyGenerator (as
serialized streams); encoding, encryption, evaluation, decryption and decoding all run through the C ABI.
TEST INFRASTRUCTURE: the reference is the checker (it runs the same program and must agree bit for bit where it can)."""
import numpy as np

import seal_amd as S
import sealref
from harness import DeviceSide


def example_ckks_basics(n=8192, bits=(60, 40, 40, 60)):
    primes = sealref.coeff_modulus_create(n, list(bits))
    ref = sealref.RefContext("ckks", n, primes, 0)
    d = DeviceSide("ckks", n, primes, 0)
    scale = 2.0 ** 40
    first = d.ctx.first_parms_id()
    sk = S.SecretKey(d.ctx)
    sk.load_bytes(ref.secret_key_save())
    pk = S.PublicKey(d.ctx)
    pk.load_bytes(ref.public_key_save())
    relin_keys = S.RelinKeys(d.ctx)
    relin_keys.load_bytes(ref.keys_save("relin", True))
    encryptor, evaluator, decryptor, encoder = S.Encryptor(d.ctx, public_key=pk), d.ev, S.Decryptor(d.ctx, sk), S.CKKSEncoder(d.ctx)
    slot_count = encoder.slot_count()
    x = np.arange(slot_count) * (1.0 / (slot_count - 1))
    plain_coeff3, plain_coeff1, plain_coeff0 = (encoder.encode(v, first, scale) for v in (3.14159265, 0.4, 1.0))
    # the single-value overload is the reference's word for word
    for v, mine in ((3.14159265, plain_coeff3), (0.4, plain_coeff1), (1.0, plain_coeff0)):
        assert np.array_equal(mine.to_numpy(), ref.ckks_encode_value(v, ref.first_chain_index, scale).data())
    assert np.array_equal(encoder.encode(-7, first, None).to_numpy(), ref.ckks_encode_value(-7, ref.first_chain_index).data())
    x1_encrypted = encryptor.encrypt(encoder.encode(x, first, scale))
    # x^2, relinearize, rescale
    x3_encrypted = x1_encrypted.copy()
    evaluator.square_inplace(x3_encrypted)
    evaluator.relinearize_inplace(x3_encrypted, relin_keys)
    assert x3_encrypted.scale() == 2.0 ** 80
    evaluator.rescale_to_next_inplace(x3_encrypted)
    # PI*x, rescale; then PI*x^3
    x1_encrypted_coeff3 = x1_encrypted.copy()
    evaluator.multiply_plain_inplace(x1_encrypted_coeff3, plain_coeff3)
    evaluator.rescale_to_next_inplace(x1_encrypted_coeff3)
    evaluator.multiply_inplace(x3_encrypted, x1_encrypted_coeff3)
    evaluator.relinearize_inplace(x3_encrypted, relin_keys)
    evaluator.rescale_to_next_inplace(x3_encrypted)
    # 0.4*x, rescale
    evaluator.multiply_plain_inplace(x1_encrypted, plain_coeff1)
    evaluator.rescale_to_next_inplace(x1_encrypted)
    # the three terms have different scales and levels (the example's point): adding them as they are is refused ...
    for bad in (lambda: evaluator.add_inplace(x3_encrypted.copy(), x1_encrypted),):
        try:
            bad()
            raise AssertionError("expected InvalidArgument (scale / parms mismatch)")
        except S.InvalidArgument:
            pass
    # ... so the scales are set to 2^40 and the levels aligned, as the example does
    x3_encrypted.set_scale(2.0 ** 40)
    x1_encrypted.set_scale(2.0 ** 40)
    last_parms_id = x3_encrypted.parms_id()
    evaluator.mod_switch_to_inplace(x1_encrypted, last_parms_id)
    evaluator.mod_switch_to_inplace(plain_coeff0, last_parms_id)
    evaluator.add_inplace(x3_encrypted, x1_encrypted)
    evaluator.add_plain_inplace(x3_encrypted, plain_coeff0)
    result = encoder.decode(decryptor.decrypt(x3_encrypted))
    want = (3.14159265 * x * x + 0.4) * x + 1
    assert np.max(np.abs(result - want)) < 1e-4, np.max(np.abs(result - want))
    return float(np.max(np.abs(result - want)))


def example_batching_rotation(n=8192, bits=(43, 43, 44, 44, 44)):
    """2_encoders.cpp (BatchEncoder part) + 6_rotation.cpp (example_rotation_bfv): the 2 x N/2 matrix, square it, rotate rows
    by 3, swap the rows, rotate rows by -4, decode"""
    primes = sealref.coeff_modulus_create(n, list(bits))
    t = sealref.plain_modulus_batching(n, 20)
    ref = sealref.RefContext("bfv", n, primes, t)
    d = DeviceSide("bfv", n, primes, t)
    sk = S.SecretKey(d.ctx, ref.secret_key())
    pk = S.PublicKey(d.ctx, ref.public_key())
    rlk = S.RelinKeys(d.ctx)
    rlk.load_bytes(ref.keys_save("relin", True))
    steps = (3, -4)
    glk = S.GaloisKeys(d.ctx)
    glk.load_bytes(ref.keys_save("galois", True, [ref.galois_elt_from_step(s) for s in steps] + [2 * n - 1]))
    be, enc, dec = S.BatchEncoder(d.ctx), S.Encryptor(d.ctx, public_key=pk), S.Decryptor(d.ctx, sk)
    row = n // 2
    m = np.zeros(n, dtype=np.uint64)
    m[0:4] = [0, 1, 2, 3]
    m[row:row + 4] = [4, 5, 6, 7]
    ct = enc.encrypt(be.encode(m))
    d.ev.square_inplace(ct)
    d.ev.relinearize_inplace(ct, rlk)
    d.ev.rotate_rows_inplace(ct, 3, glk)
    d.ev.rotate_columns_inplace(ct, glk)
    d.ev.rotate_rows_inplace(ct, -4, glk)
    got = be.decode(dec.decrypt(ct)).reshape(2, row)
    want = ((m.astype(object) ** 2) % t).astype(np.uint64).reshape(2, row)
    want = np.roll(np.roll(want, -3, axis=1)[::-1], 4, axis=1)
    assert np.array_equal(got, want)
