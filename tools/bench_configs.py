#!/usr/bin/env python
"""Throughput of the OTHER BASELINE.json configurations on one MI355X (bench.py measures the headline one):
  C2  CKKS N=8192,  {60,40,40,60}:        forward + inverse NTT over all RNS components
  C3  CKKS N=16384, {60,6x50,60}:         multiply + relinearize + rescale_to_next
  C4  BFV  N=32768, 14x55-bit, t=Batching(32768,20): multiply + relinearize + mod_switch_to_next
  C5r CKKS N=65536, {60,14x50,60}:        rotate_vector(1) + rescale_to_next
Synthetic uniform ciphertexts and keys resident in HBM (native/bench/bench.h:195-270), HIP-event timing on the launch
stream; the reference's own Evaluator on the host cores next to each (oracle/_ref, bounded sample).  One line per config."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="C2,C3,C4,C5r")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()
    import torch
    import seal_amd as S
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)

    def uni(prs, prefix, n):
        return torch.cat([torch.randint(0, int(q), tuple(prefix) + (1, n), dtype=torch.int64, device=device) for q in prs],
                         dim=len(prefix)).contiguous()

    def context(scheme, n, bits, tb=0):
        primes = S.CoeffModulus.Create(n, bits)
        p = S.EncryptionParameters(scheme)
        p.set_poly_modulus_degree(n)
        p.set_coeff_modulus(primes)
        t = 0
        if scheme != "ckks":
            t = S.PlainModulus.Batching(n, tb)
            p.set_plain_modulus(t)
        return primes, t, S.SEALContext(p, True, 0)

    def timed(fn, reps):
        timer = S.HipTimer()
        fn()
        S.device_synchronize()
        timer.start()
        for _ in range(reps):
            fn()
        return timer.stop() / reps

    def cpu(scheme, n, primes, t, pipeline, threads_reps=2):
        if args.no_cpu:
            return None
        try:
            import sealref
            if not sealref.available():
                return None
            ref = sealref.RefContext(scheme, n, primes, t)
            ref.keygen_relin()
            if pipeline == "rotate":
                ref.keygen_galois_steps([1])
            threads = os.cpu_count() or 1
            secs = ref.time_pipeline(pipeline, threads, threads_reps)
            return dict(value=round(threads * threads_reps / secs, 2), cores=threads)
        except Exception as e:
            return dict(error=repr(e))

    for name in args.configs.split(","):
        if name == "C2":
            n, bits, polys = 8192, [60, 40, 40, 60], 8192
            primes, t, ctx = context("ckks", n, bits)
            K = len(primes)
            buf = uni(primes, (polys,), n)

            class B:
                ptr = buf.data_ptr()
            ms_f = timed(lambda: S.ntt_forward(ctx, B, polys, K), 10)
            ms_i = timed(lambda: S.ntt_inverse(ctx, B, polys, K), 10)
            alg = 16.0 * n * K * polys
            print(json.dumps(dict(config="C2 CKKS N=8192 L=4 NTT over all components (%d polys, %.0f MB)" % (polys, alg / 2e6),
                                  fwd_GBs=round(alg / ms_f / 1e6, 1), inv_GBs=round(alg / ms_i / 1e6, 1),
                                  fwd_frac=round(alg / ms_f / 1e6 / 8000, 3), inv_frac=round(alg / ms_i / 1e6 / 8000, 3))), flush=True)
            continue
        if name == "C3":
            scheme, n, bits, tb, B_, pipe = "ckks", 16384, [60] + [50] * 6 + [60], 0, 1024, "ckks_mul_relin_rescale"
        elif name == "C4":
            scheme, n, bits, tb, B_, pipe = "bfv", 32768, [55] * 14, 20, 64, "bfv_mul_relin_modswitch"
        else:
            scheme, n, bits, tb, B_, pipe = "ckks", 65536, [60] + [50] * 14 + [60], 0, 256, "rotate"
        primes, t, ctx = context(scheme, n, bits, tb)
        L, K = len(primes), len(primes) - 1
        ev = S.Evaluator(ctx)
        key = uni(primes, (K, 2), n)
        if pipe == "rotate":
            keys = S.GaloisKeys(ctx)
            keys.set_key_device(S.GaloisKeys.get_index(ctx.galois_elt_from_step(1)), K, key.data_ptr())
        else:
            keys = S.RelinKeys(ctx)
            keys.set_key_device(0, K, key.data_ptr())
        del key
        first = ctx.first_parms_id()
        ntt = scheme == "ckks"
        scale = 2.0 ** 24 if ntt else 1.0

        def make(tens):
            ct = S.Ciphertext(ctx, batch=B_)
            ct.resize(first, 2)
            ct.set_is_ntt_form(ntt)
            ct.set_scale(scale)
            ct.load_device(tens.data_ptr(), tens.numel())
            return ct
        xs, ys = uni(primes[:K], (2, B_), n), uni(primes[:K], (2, B_), n)
        x, y = make(xs), make(ys)
        work = S.Ciphertext(ctx, batch=B_)

        def step():
            if pipe == "rotate":
                work2 = x.copy()
                ev.rotate_vector_inplace(work2, 1, keys)
                work2.set_scale(float(primes[K - 1]) * 2.0 ** 10)
                ev.rescale_to_next_inplace(work2)
            elif scheme == "ckks":
                ev.multiply(x, y, work)
                ev.relinearize_inplace(work, keys)
                ev.rescale_to_next_inplace(work)
            else:
                ev.multiply(x, y, work)
                ev.relinearize_inplace(work, keys)
                ev.mod_switch_to_next_inplace(work)
        ms = timed(step, args.reps)
        line = dict(config="%s %s N=%d L=%d: %s, batch %d" % (name, scheme.upper(), n, L, pipe, B_),
                    value=round(B_ / ms * 1e3, 1), unit="ciphertexts/s", ms_per_batch=round(ms, 3),
                    cpu_reference=cpu(scheme, n, primes, t, pipe))
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
