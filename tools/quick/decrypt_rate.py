"""Decryption rate on the GPU box: Decryptor_DecryptBatch over a device-resident batch of fresh ciphertexts (HIP-event time on the
null stream) next to the reference's own Decryptor::decrypt on one host thread (oracle/_ref, baseline only).
CKKS N=65536 {60,14x50,60} (the headline parameters) and BFV N=32768 14x55 bit (BASELINE configs[3])."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import seal_amd as S
import sealref
from harness import DeviceSide

for scheme, n, bits, batch in (("ckks", 65536, [60] + [50] * 14 + [60], 64), ("bfv", 32768, [55] * 14, 64)):
    primes = sealref.coeff_modulus_create(n, bits)
    t = sealref.plain_modulus_batching(n, 20) if scheme != "ckks" else 0
    ref = sealref.RefContext(scheme, n, primes, t)
    d = DeviceSide(scheme, n, primes, t)
    dec = S.Decryptor(d.ctx, S.SecretKey(d.ctx, ref.secret_key()))
    rng = np.random.default_rng(1)
    c = ref.ckks_encrypt(rng.standard_normal(n // 2), 2.0 ** 30) if scheme == "ckks" else ref.batch_encrypt(rng.integers(0, t, n, dtype=np.uint64))
    i = c.info()
    arr = np.repeat(c.data()[:, None], batch, axis=1)
    cts = S.Ciphertext.from_numpy(d.ctx, arr, d.ctx.parms_id_at(i["chain_index"]), i["is_ntt_form"], i["scale"], i["correction_factor"])
    buf, words = dec.decrypt_batch(cts)
    timer = S.HipTimer()
    reps = 10
    timer.start()
    for _ in range(reps):
        dec.decrypt_batch(cts, out=buf)
    ms = timer.stop() / reps
    t0 = time.perf_counter()
    for _ in range(3):
        ref.decrypt(c)
    cpu = (time.perf_counter() - t0) / 3
    alg = (2 if scheme == "ckks" else 2) * len(primes[:-1]) * n * 8 * batch  # the ciphertext words read once
    print("%s N=%d K=%d: Decryptor_DecryptBatch %d ciphertexts %.3f ms = %.0f ct/s (%.0f GB/s of ciphertext words); reference Decryptor::decrypt %.2f ms/ct on one host thread = %.0f ct/s" % (
        scheme, n, len(primes) - 1, batch, ms, batch / ms * 1e3, alg / ms / 1e6, cpu * 1e3, 1 / cpu), flush=True)
