"""Client-side steps on the GPU box at the headline parameters (CKKS N = 65536, {60, 14x50, 60}) and BASELINE configs[3]'s
(BFV N = 32768, 14 x 55 bit): encode, encrypt (public key / secret key), decrypt, decode through the C ABI, one object at a time
(wall time per call incl. the host-side sampling and the host <-> device copies), next to the reference's own classes on one host
thread (oracle/_ref, baseline only)."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import seal_amd as S
import sealref
from harness import DeviceSide


def t(fn, reps=5):
    fn()
    S.device_synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    S.device_synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for scheme, n, bits in (("ckks", 65536, [60] + [50] * 14 + [60]), ("bfv", 32768, [55] * 14)):
    primes = sealref.coeff_modulus_create(n, bits)
    tt = sealref.plain_modulus_batching(n, 20) if scheme != "ckks" else 0
    ref = sealref.RefContext(scheme, n, primes, tt)
    d = DeviceSide(scheme, n, primes, tt)
    sk, pk = S.SecretKey(d.ctx, ref.secret_key()), S.PublicKey(d.ctx, ref.public_key())
    enc_pk, enc_sk, dec = S.Encryptor(d.ctx, public_key=pk), S.Encryptor(d.ctx, sk), S.Decryptor(d.ctx, sk)
    rng = np.random.default_rng(3)
    first = d.ctx.first_parms_id()
    if scheme == "ckks":
        coder = S.CKKSEncoder(d.ctx)
        vals = rng.standard_normal(n // 2)
        encode = lambda: coder.encode(vals, first, 2.0 ** 40)
        rencode = lambda: ref.ckks_encode(vals, ref.first_chain_index, 2.0 ** 40)
        rdecode = lambda p: ref.ckks_decode(p)
    else:
        coder = S.BatchEncoder(d.ctx)
        vals = rng.integers(0, tt, n, dtype=np.uint64)
        encode = lambda: coder.encode(vals)
        rencode = lambda: ref.batch_encode(vals)
        rdecode = lambda p: ref.batch_decode(p)
    pt = encode()
    ct = enc_pk.encrypt(pt)
    out = dec.decrypt(ct)
    rpt = rencode()
    rct, _ = ref.ct_load(ct.save_bytes())
    rout = ref.decrypt(rct)
    rows = [("encode", t(encode), t(rencode, 3)),
            ("encrypt (public key)", t(lambda: enc_pk.encrypt(pt)), t(lambda: ref.encrypt_asymmetric_save(rpt), 3)),
            ("encrypt (secret key)", t(lambda: enc_sk.encrypt_symmetric(pt)), t(lambda: ref.encrypt_symmetric_save(rpt, False), 3)),
            ("decrypt", t(lambda: dec.decrypt(ct)), t(lambda: ref.decrypt(rct), 3)),
            ("decode", t(lambda: coder.decode(out)), t(lambda: rdecode(rout), 3))]
    print("%s N=%d K=%d (ms per object: C ABI on MI355X incl. host sampling and copies | reference, one host thread incl. its save for the encrypt rows)" % (scheme, n, len(primes) - 1))
    for name, a, b in rows:
        print("  %-22s %8.2f | %8.2f" % (name, a, b), flush=True)
