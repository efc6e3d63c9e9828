"""Key generation on the GPU box at the headline parameters (CKKS N = 65536, {60, 14x50, 60}) and BASELINE configs[3]'s (BFV
N = 32768, 14 x 55 bit): KeyGenerator through the C ABI (secret key, public key, RelinKeys, GaloisKeys for 8 elements; wall time,
keys left in HBM in the key-switching layout) next to the reference's KeyGenerator on one host thread (oracle/_ref, baseline
only; its Galois keys timed on 2 elements and scaled)."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import seal_amd as S
import sealref
from harness import DeviceSide


def t(fn, reps=3):
    fn()
    S.device_synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    S.device_synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def once(fn):
    t0 = time.perf_counter()
    fn()
    return (time.perf_counter() - t0) * 1e3


for scheme, n, bits in (("ckks", 65536, [60] + [50] * 14 + [60]), ("bfv", 32768, [55] * 14)):
    primes = sealref.coeff_modulus_create(n, bits)
    tt = sealref.plain_modulus_batching(n, 20) if scheme != "ckks" else 0
    d = DeviceSide(scheme, n, primes, tt)
    elts = [3, 9, 27, 81, 243, 729, 2187, 2 * n - 1]
    kg = S.KeyGenerator(d.ctx)
    rows = [("KeyGenerator (secret key)", t(lambda: S.KeyGenerator(d.ctx))),
            ("create_public_key", t(kg.create_public_key)),
            ("create_relin_keys", t(kg.create_relin_keys)),
            ("create_galois_keys (8 elements)", t(lambda: kg.create_galois_keys(elts), 2)),
            ("create_galois_keys() (all: %d)" % (2 * (n.bit_length() - 2) + 1), t(lambda: kg.create_galois_keys(), 1))]
    rows.append(("RelinKeys seeded stream (%.0f MB)" % (len(kg.save_seeded()) / 1e6), t(kg.save_seeded, 2)))
    t_ctx = once(lambda: sealref.RefContext(scheme, n, primes, tt))   # includes the reference's KeyGenerator(context)
    ref = sealref.RefContext(scheme, n, primes, tt)
    refs = [t_ctx, once(ref.public_key), once(ref.keygen_relin), once(lambda: ref.keygen_galois_elts(elts[:2])) * 4]
    refs.append(refs[-1] / 8 * (2 * (n.bit_length() - 2) + 1))   # scaled from the two timed elements
    refs.append(once(lambda: ref.keys_save("relin", seeded=True)))   # create_relin_keys() + save + reload by the shim
    mb = (len(primes) - 1) * 2 * len(primes) * n * 8 / 1e6
    print("%s N=%d L=%d, one key = %.0f MB (ms: C ABI on MI355X | reference on one host thread%s)" %
          (scheme, n, len(primes), mb, "; its first row includes SEALContext creation"))
    for (name, a), b in zip(rows, refs):
        print("  %-34s %9.2f | %10.1f" % (name, a, b), flush=True)
