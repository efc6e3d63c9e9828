"""single-launch NTT at N=8192: rate vs the number of workgroups per component (SEALHIP_NTT_FCHUNKS) and the number of components"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT)
import numpy as np
import seal_amd as S
import torch
dev = torch.device("cuda", 0)
n = 8192
for bits in ([50, 40, 40, 50], [50, 40, 50], [50, 40, 40, 40, 40, 50]):
    pr = S.CoeffModulus.Create(n, bits)
    p = S.EncryptionParameters('ckks'); p.set_poly_modulus_degree(n); p.set_coeff_modulus(pr)
    ctx = S.SEALContext(p, True, 0)
    comps = len(pr)
    polys = 16384 // comps
    data = torch.cat([torch.randint(0, int(q), (polys, 1, n), dtype=torch.int64, device=dev) for q in pr], dim=1).contiguous()
    class B: ptr = data.data_ptr()
    t = S.HipTimer()
    for ch in ("", "64", "128", "256", "512", "1024", "2048"):
        if ch: os.environ["SEALHIP_NTT_FCHUNKS"] = ch
        else: os.environ.pop("SEALHIP_NTT_FCHUNKS", None)
        res = []
        for fn in (S.ntt_forward, S.ntt_inverse):
            for _ in range(3): fn(ctx, B, polys, comps)
            t.start()
            for _ in range(10): fn(ctx, B, polys, comps)
            ms = t.stop() / 10
            res.append(16.0 * n * comps * polys / ms / 1e6)
        print("comps %d polys %d chunks/comp %-5s: fwd %7.1f GB/s  inv %7.1f GB/s" % (comps, polys, ch or "auto", res[0], res[1]), flush=True)
