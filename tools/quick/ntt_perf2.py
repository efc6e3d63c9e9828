"""NTT throughput at N=8192 / 16384 with a working set well above the 256 MiB Infinity Cache."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT)
import numpy as np
import seal_amd as S

for n, bits, polys in [(8192, [50, 50, 50, 50], 4096), (8192, [60, 40, 40, 60], 4096), (16384, [50] * 8, 1024)]:
    pr = S.CoeffModulus.Create(n, bits)
    comps = len(bits) - 1 if bits[0] == 50 else len(bits)
    p = S.EncryptionParameters('ckks'); p.set_poly_modulus_degree(n); p.set_coeff_modulus(pr)
    ctx = S.SEALContext(p)
    rng = np.random.default_rng(1)
    d = rng.integers(0, min(pr) >> 1, size=(64, comps, n), dtype=np.uint64)
    d = np.tile(d, (polys // 64, 1, 1))
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
        print("N=%6d bits=%s comps=%2d polys=%4d (%4.0f MB) %s: %8.3f ms  %8.1f GB/s algorithmic (%.1f%% of 8 TB/s)" % (
            n, bits, comps, polys, bytes_alg / 2e6, name, ms, bytes_alg / ms / 1e6, bytes_alg / ms / 1e6 / 80.0), flush=True)
