"""One NTT size, forward and inverse, for rocprofv3 counter passes:  ntt_one.py <log2 N> <comps> <polys> [reps]"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT)
import numpy as np
import seal_amd as S

logn, comps, polys = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
n = 1 << logn
bits = [int(b) for b in os.environ.get("NTT_BITS", "50").split(",")]
pr = S.CoeffModulus.Create(n, (bits * (comps + 1))[:comps + 1])
p = S.EncryptionParameters('ckks'); p.set_poly_modulus_degree(n); p.set_coeff_modulus(pr)
ctx = S.SEALContext(p)
rng = np.random.default_rng(1)
d = np.stack([rng.integers(0, q, size=(polys, n), dtype=np.uint64) for q in pr[:comps]], axis=1)
buf = S.DeviceBuffer.from_numpy(np.ascontiguousarray(d))
t = S.HipTimer()
for name, fn in [('fwd', S.ntt_forward), ('inv', S.ntt_inverse)]:
    fn(ctx, buf, polys, comps)
    t.start()
    for _ in range(reps):
        fn(ctx, buf, polys, comps)
    ms = t.stop() / reps
    print("N=%d comps=%d polys=%d %s: %.3f ms  %.1f GB/s algorithmic  %.3f us/transform" % (
        n, comps, polys, name, ms, 16.0 * n * comps * polys / ms / 1e6, ms * 1e3 / (comps * polys)), flush=True)
