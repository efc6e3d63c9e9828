#!/usr/bin/env python
"""bench.py — north-star benchmark of the hot path on MI355X.

metric (BASELINE.json): CKKS multiply + relinearize + rescale_to_next throughput, ciphertexts/s, at
N = 2^16, L = 16 (CoeffModulus::Create(65536, {60, 14x50, 60}), K = 15 data primes), batches of
synthetic uniform ciphertexts resident in HBM (the distribution sealbench uses, native/bench/bench.h:195-270).
A "step" = one pass of multiply_inplace + relinearize_inplace + rescale_to_next_inplace over a batch of
`--batch` ciphertexts per GPU.  N GPUs = N independent processes each owning its own batch, context
tables and keys (batch sharding, no data-path collective: SURVEY §8(e).1) -> "scaling": "weak".

One JSON line is printed by rank 0 with, besides the contract fields:
  roofline     the NTT (dominant kernel family: one batched ntt_forward = column-pass + row-pass kernel)
               measured live with HIP events on the stream it runs on; achieved = algorithmic bytes
               (16*N per RNS-component transform) / time; peak = 8 TB/s HBM (guide).
  roofline_configs1  the same measurement at BASELINE configs[1] (CKKS N=8192, L=4: forward and inverse NTT over all
               RNS components), where the transform fits the LDS and the single-launch kernels move algorithmic bytes only.
  cpu_baseline the reference's own Evaluator (oracle/_ref = Microsoft SEAL 4.4.3, HEXL off) timed on this
               host's cores on a bounded sample, rank 0, N=1 only ("port": the plain-C restatement, 1 core,
               when oracle/_ref did not travel).  Checker/baseline only — never the thing measured.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

N_POLY = 65536
BITS = [60] + [50] * 14 + [60]
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=256, help="ciphertexts per GPU per step (SURVEY 8(d): device-resident throughput batches of 64 / 256 / 1024)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = all host cores")
    ap.add_argument("--cpu-reps", type=int, default=4)
    ap.add_argument("--ntt-only", action="store_true", help="only the NTT roofline leg (for rocprofv3 runs)")
    ap.add_argument("--graph", action="store_true", help="replay the step as one captured hipGraph (Evaluator_BeginCapture/EndCapture): "
                    "for launch-bound small batches; the default (eager) path is what the headline number uses")
    return ap.parse_args()


def device_uniform(torch, primes, shape_prefix, n, device):
    """uniform residues per RNS component, generated on the device: [*prefix][len(primes)][n] int64"""
    comps = [torch.randint(0, int(q), tuple(shape_prefix) + (1, n), dtype=torch.int64, device=device) for q in primes]
    return torch.cat(comps, dim=len(shape_prefix)).contiguous()


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a MI355X: no HIP device visible (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend="nccl", device_id=device)

    import seal_amd as S
    from seal_amd import shard

    n, B = N_POLY, args.batch
    primes = S.CoeffModulus.Create(n, BITS)
    L, K = len(primes), len(primes) - 1
    parms = S.EncryptionParameters("ckks")
    parms.set_poly_modulus_degree(n)
    parms.set_coeff_modulus(primes)
    ctx = S.SEALContext(parms, True, 0)  # sec_level_type::none, as sealbench (native/bench/bench.h:35-36)
    ev = S.Evaluator(ctx)

    torch.manual_seed(0x5EA1 + rank)
    # synthetic relinearization key: K digits x 2 polys x L comps, uniform per component (240 MiB)
    key = device_uniform(torch, primes, (K, 2), n, device)
    rlk = S.RelinKeys(ctx)
    rlk.set_key_device(0, K, key.data_ptr())
    del key
    # synthetic size-2 ciphertext batches at the first data level, NTT form, scale = safe_scale
    scale = 2.0 ** (50 // 2 - 1)
    xs = device_uniform(torch, primes[:K], (2, B), n, device)
    ys = device_uniform(torch, primes[:K], (2, B), n, device)
    first = ctx.first_parms_id()

    def make_ct(t):
        ct = S.Ciphertext(ctx, batch=B)
        ct.resize(first, 2)
        ct.set_is_ntt_form(True)
        ct.set_scale(scale)
        ct.load_device(t.data_ptr(), t.numel())
        return ct

    x, y = make_ct(xs), make_ct(ys)
    work = S.Ciphertext(ctx, batch=B)
    torch.cuda.synchronize()

    def step():
        ev.multiply(x, y, work)          # work = x * y (size 3); x stays resident as the next step's input
        ev.relinearize_inplace(work, rlk)
        ev.rescale_to_next_inplace(work)

    if args.graph and not args.ntt_only:
        step()  # eager once: lazily built tables, pool warm-up
        torch.cuda.synchronize()
        graph = ev.capture(step)
        eager_step, step = step, graph.launch

    result = {}
    if not args.ntt_only:
        elapsed = shard.timed_steps(step, args.steps, args.warmup, dist if world > 1 else None, torch.cuda.synchronize, torch, device)
        assert work.size() == 2 and work.coeff_modulus_size() == K - 1
        rate = shard.whole_job_rate(B, args.steps, elapsed, dist if world > 1 else None, torch, device)
        result = dict(value=rate, ms_per_step=1e3 * elapsed / args.steps)

    # ---- roofline leg: the batched forward NTT over the resident batch (2*B polys x K comps)
    roofline = None
    if rank == 0:
        buf_words = xs.numel()
        timer = S.HipTimer()
        polys = 2 * B

        class _Buf:
            ptr = xs.data_ptr()
        for _ in range(3):
            S.ntt_forward(ctx, _Buf, polys, K)
        reps = 20
        timer.start()
        for _ in range(reps):
            S.ntt_forward(ctx, _Buf, polys, K)
        ms = timer.stop() / reps
        alg_bytes = 16.0 * n * K * polys
        achieved = alg_bytes / (ms * 1e-3) / 1e9
        # HBM bytes of one launch from the committed rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE, separate passes,
        # MI355X_MICROARCH.md).  The passes record their own launch shape; traffic per transform does not depend on
        # the batch, so a different shape is scaled by the number of transforms.
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "r01_ntt_pmc.json")
        if os.path.exists(pmc):
            try:
                pj = json.load(open(pmc))
                traffic = int(round(pj["hbm_bytes_per_launch"] * (K * polys) / float(pj.get("transforms_per_launch", 480))))
            except Exception:
                traffic = None
        roofline = dict(bound="hbm", kernel="ntt_forward = ntt2_fwd_p1 + ntt2_fwd_p2 (two-pass engine), %d transforms of 2^16 per launch" % (K * polys),
                        achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 4),
                        traffic=traffic, ms_per_launch=round(ms, 4), algorithmic_bytes_per_launch=alg_bytes)
        assert buf_words == 2 * B * K * n

    # ---- BASELINE configs[1]: CKKS N=8192, L=4, forward + inverse NTT over all RNS components (single-launch kernels,
    # the transform is 64 KiB and stays in LDS between the passes: HBM traffic = algorithmic bytes).  Two chains: the
    # config's own {60,40,40,60} (its two 60-bit primes run on the slower 64-bit integer back end, two launches) and a
    # chain whose primes are all below 2^50 (every component on the double-precision back end).
    ntt_c1 = None
    if rank == 0 and world == 1:
        ntt_c1 = ntt_configs1(S, torch, device)

    # ---- CPU baseline (rank 0, N=1 only): the reference's Evaluator on this host's cores
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.ntt_only:
        cpu = cpu_baseline(primes, args)

    if rank == 0:
        line = dict(
            metric="CKKS multiply+relinearize+rescale ciphertexts/sec @ N=2^16, L=16",
            value=round(result.get("value", 0.0), 2), unit="ciphertexts/s", n_gpus=world, steps=args.steps,
            warmup=args.warmup, ms_per_step=round(result.get("ms_per_step", 0.0), 3), higher_is_better=True,
            scaling="weak", vs_baseline=None, dtype="u64", data="synthetic",
            config=dict(workload="CKKS N=65536, coeff_modulus {60,14x50,60} (L=16, K=15): multiply_inplace + "
                                 "relinearize_inplace + rescale_to_next_inplace, device-resident batches",
                        batch_per_gpu=B, launch="hipGraph replay" if args.graph else "eager", parallelism="batch-sharded x%d, no data-path collective" % world,
                        arithmetic="64-bit residues: exact double-precision (error-free FMA) arithmetic for the 14 primes below "
                                   "2^50, 64-bit Shoup/Barrett integer arithmetic for the two 60-bit primes; results canonical u64",
                        key_bytes=2 * K * L * n * 8, algorithmic_bytes_per_ciphertext=(2 * K * K + 18 * K - 2) * 8 * n),
            roofline=roofline, roofline_configs1=ntt_c1, cpu_baseline=cpu)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


def ntt_configs1(S, torch, device, polys=4096, reps=10):
    """forward / inverse NTT rate at BASELINE configs[1] (N = 8192, L = 4), batch of `polys` polynomials resident in HBM
    (1 GiB: four times the Infinity Cache), HIP events on the launch stream, algorithmic bytes = 16*N per component."""
    n = 8192
    out = []
    for label, bits in (("configs[1] {60,40,40,60}", [60, 40, 40, 60]), ("all primes < 2^50 {50,40,40,50}", [50, 40, 40, 50])):
        pr = S.CoeffModulus.Create(n, bits)
        p = S.EncryptionParameters("ckks")
        p.set_poly_modulus_degree(n)
        p.set_coeff_modulus(pr)
        ctx = S.SEALContext(p, True, 0)
        comps = len(pr)
        data = device_uniform(torch, pr, (polys,), n, device)

        class _Buf:
            ptr = data.data_ptr()
        timer = S.HipTimer()
        rates = {}
        for name, fn in (("forward", S.ntt_forward), ("inverse", S.ntt_inverse)):
            for _ in range(3):
                fn(ctx, _Buf, polys, comps)
            timer.start()
            for _ in range(reps):
                fn(ctx, _Buf, polys, comps)
            ms = timer.stop() / reps
            alg = 16.0 * n * comps * polys
            rates[name] = dict(achieved=round(alg / (ms * 1e-3) / 1e9, 1), frac=round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                               ms_per_launch=round(ms, 4), algorithmic_bytes_per_launch=alg)
        # HBM bytes per launch of the single-launch kernels from the committed PMC passes (all-double-precision chain only:
        # the passes were taken on it), scaled by the number of transforms as for the main roofline leg
        if "2^50" in label:
            try:
                pj = json.load(open(os.path.join(ROOT, "profiles", "r01_ntt_pmc.json")))["single_launch_n8192"]
                per = float(pj["algorithmic_kib"]) * 1024 / (16.0 * n)  # transforms in the recorded launch
                for name in ("forward", "inverse"):
                    rates[name]["traffic"] = int(round(pj[name]["hbm_bytes_per_launch"] * (comps * polys) / per))
            except Exception:
                pass
        out.append(dict(chain=label, transforms_per_launch=comps * polys, **rates))
        del data
    return dict(bound="hbm", unit="GB/s", peak=HBM_PEAK_GBS, workload="CKKS N=8192, L=4: batched NTT / INTT over all RNS components", chains=out)


def cpu_baseline(primes, args):
    import numpy as np
    n = N_POLY
    try:
        import sealref
        if sealref.available():
            threads = args.cpu_threads or (os.cpu_count() or 1)
            ref = sealref.RefContext("ckks", n, primes)
            ref.keygen_relin()
            secs = ref.time_pipeline("ckks_mul_relin_rescale", threads, args.cpu_reps)
            cts = threads * args.cpu_reps
            one = ref.time_pipeline("ckks_mul_relin_rescale", 1, 2)
            return dict(value=round(cts / secs, 3), unit="ciphertexts/s", cores=threads, kind="reference",
                        single_thread_value=round(2 / one, 3),
                        sample="%d threads x %d ciphertexts each (after one untimed warm-up pass per thread), "
                               "seal::Evaluator multiply+relinearize+rescale, HEXL off, same parameters" % (threads, args.cpu_reps))
    except Exception as e:  # the baseline must never take the benchmark down
        sys.stderr.write("cpu_baseline(reference) unavailable: %r\n" % (e,))
    try:
        import sealoracle
        from oracle import rand_ct
        rng = np.random.default_rng(0x5EA1)
        K = len(primes) - 1
        po = sealoracle.PortContext("ckks", n, primes)
        a, b = rand_ct(rng, primes, K, n), rand_ct(rng, primes, K, n)
        rlk = np.stack([np.stack([np.stack([rng.integers(0, q, n, dtype=np.uint64) for q in primes]) for _ in range(2)])
                        for _ in range(K)])
        secs, _ = po.time_ckks_pipeline(a, b, rlk, 1)
        return dict(value=round(1 / secs, 4), unit="ciphertexts/s", cores=1, kind="port",
                    sample="1 ciphertext, plain-C restatement (oracle/seal_oracle.c), 1 thread")
    except Exception as e:
        sys.stderr.write("cpu_baseline(port) unavailable: %r\n" % (e,))
    return None


if __name__ == "__main__":
    main()
