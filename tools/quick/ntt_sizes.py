"""Forward / inverse NTT throughput of mixed chains (60-bit primes on the integer back end, the rest on the double-precision one)
at the single-launch sizes N = 2^13, 2^14 and at 2^15; working sets well above the 256 MiB Infinity Cache."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT)
import numpy as np
import seal_amd as S

for n, bits, polys in [(8192, [60, 40, 40, 60], 4096), (16384, [60, 50, 50, 50, 50, 50, 50, 60], 1024), (16384, [60, 59, 58, 60], 2048),
                       (16384, [50] * 8, 1024), (32768, [55] * 8, 512)]:
    pr = S.CoeffModulus.Create(n, bits)
    comps = len(bits)
    p = S.EncryptionParameters('ckks'); p.set_poly_modulus_degree(n); p.set_coeff_modulus(pr)
    ctx = S.SEALContext(p)
    rng = np.random.default_rng(1)
    d = rng.integers(0, min(pr) >> 1, size=(32, comps, n), dtype=np.uint64)
    d = np.tile(d, (polys // 32, 1, 1))
    buf = S.DeviceBuffer.from_numpy(d)
    t = S.HipTimer()
    out = []
    for name, fn in [('fwd', S.ntt_forward), ('inv', S.ntt_inverse)]:
        for _ in range(3):
            fn(ctx, buf, polys, comps)
        reps = 10
        t.start()
        for _ in range(reps):
            fn(ctx, buf, polys, comps)
        ms = t.stop() / reps
        out.append("%s %7.1f GB/s (%.3f)" % (name, 16.0 * n * comps * polys / ms / 1e6, 16.0 * n * comps * polys / ms / 1e6 / 8000.0))
    print("N=%6d %-28s x%5d: %s" % (n, bits, polys, "  ".join(out)), flush=True)
