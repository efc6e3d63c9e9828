"""Forward and inverse NTT at the bench shape (N=2^16, {60,14x50} primes, 32 polys) for rocprofv3 kernel stats."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT)
import numpy as np
import seal_amd as S

n, polys = 65536, 32
pr = S.CoeffModulus.Create(n, [60] + [50] * 14 + [60])
comps = 15
p = S.EncryptionParameters('ckks'); p.set_poly_modulus_degree(n); p.set_coeff_modulus(pr)
ctx = S.SEALContext(p)
rng = np.random.default_rng(1)
d = rng.integers(0, pr[1] >> 1, size=(polys, comps, n), dtype=np.uint64)
buf = S.DeviceBuffer.from_numpy(d)
t = S.HipTimer()
for name, fn in [('fwd', S.ntt_forward), ('inv', S.ntt_inverse)]:
    for _ in range(20):
        fn(ctx, buf, polys, comps)
    reps = 20
    t.start()
    for _ in range(reps):
        fn(ctx, buf, polys, comps)
    ms = t.stop() / reps
    print("%s %.4f ms  %.1f GB/s" % (name, ms, 16.0 * n * comps * polys / ms / 1e6), flush=True)
