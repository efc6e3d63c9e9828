"""The one-launch ring kernel of N = 2^16 (ntt2_ring.hip) against the two launches: algorithmic GB/s (16*N bytes per component transform).
Run once per setting (the switch is read once per process):  SEALHIP_NTT_RING=0|1 [SEALHIP_RING_DEBUG=1] python tools/quick/ring_probe.py [bits-list] [polys]"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT)
import numpy as np
import seal_amd as S

bits = [int(b) for b in (sys.argv[1] if len(sys.argv) > 1 else ",".join(["50"] * 15)).split(",")]
polys = int(sys.argv[2]) if len(sys.argv) > 2 else 512
n, comps = 65536, len(bits) - 1
pr = S.CoeffModulus.Create(n, bits)
p = S.EncryptionParameters('ckks'); p.set_poly_modulus_degree(n); p.set_coeff_modulus(pr)
ctx = S.SEALContext(p)
rng = np.random.default_rng(1)
d = rng.integers(0, min(pr) >> 1, size=(polys, comps, n), dtype=np.uint64)
buf = S.DeviceBuffer.from_numpy(d)
t = S.HipTimer()
for _ in range(3):
    S.ntt_forward(ctx, buf, polys, comps)
reps = 10
t.start()
for _ in range(reps):
    S.ntt_forward(ctx, buf, polys, comps)
ms = t.stop() / reps
alg = 16.0 * n * comps * polys
print("ring=%s bits=%s polys=%d: %8.3f ms  %8.1f GB/s algorithmic (%.3f of 8 TB/s)" % (os.environ.get("SEALHIP_NTT_RING", "default"), sys.argv[1] if len(sys.argv) > 1 else "15x50", polys, ms, alg / ms / 1e6, alg / ms / 1e6 / 8000.0), flush=True)
