"""Quick NTT throughput probe: algorithmic GB/s (16*N bytes per component transform)."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT)
import numpy as np
import seal_amd as S

print(S.device_info())
for n, comps, polys in [(4096, 3, 512), (8192, 3, 512), (16384, 7, 128), (32768, 13, 64), (65536, 15, 64), (65536, 15, 8)]:
    pr = S.CoeffModulus.Create(n, [50] * (comps + 1))
    p = S.EncryptionParameters('ckks'); p.set_poly_modulus_degree(n); p.set_coeff_modulus(pr)
    ctx = S.SEALContext(p)
    rng = np.random.default_rng(1)
    d = rng.integers(0, pr[0] >> 1, size=(polys, comps, n), dtype=np.uint64)
    buf = S.DeviceBuffer.from_numpy(d)
    t = S.HipTimer()
    for name, fn in [('fwd', S.ntt_forward), ('inv', S.ntt_inverse)]:
        for _ in range(3):
            fn(ctx, buf, polys, comps)
        reps = 10
        t.start()
        for _ in range(reps):
            fn(ctx, buf, polys, comps)
        ms = t.stop() / reps
        bytes_alg = 16.0 * n * comps * polys
        print("N=%6d comps=%2d polys=%4d %s: %8.3f ms  %8.1f GB/s algorithmic (%.1f%% of 8 TB/s)  %.2f us/transform" % (
            n, comps, polys, name, ms, bytes_alg / ms / 1e6, bytes_alg / ms / 1e6 / 80.0, ms * 1e3 / (comps * polys)), flush=True)
