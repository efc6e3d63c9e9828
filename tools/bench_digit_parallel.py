#!/usr/bin/env python
"""BASELINE configs[4] as written: CKKS N=65536, L=16 - rotate_vector (apply_galois key switch with the decomposition
digits spread over the GPUs of the node) + rescale_to_next.

  python -m torch.distributed.run --nnodes=1 --nproc-per-node G --master-addr 127.0.0.1 --master-port P \\
         tools/bench_digit_parallel.py [--batch B] [--steps K] [--warmup W]

Every rank holds the SAME batch of ciphertexts and only its slice of the Galois key's digits
(KSwitchKeys_SetKeyDigits); per key switch there is ONE all-reduce of 2 (K+1) N 64-bit words per ciphertext over
RCCL/xGMI (seal_amd.shard.DigitParallel); the rescale runs redundantly on every rank.  Reported: ciphertexts/s of the
node (strong scaling: the work of one batch is divided over the ranks), latency per batch, bytes all-reduced, and a
SHA-256 of rank 0's result that must not depend on G (bit-exact by construction).  With one process it runs the same
code path without the collective."""
import argparse
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    args = ap.parse_args()
    import numpy as np
    import torch
    import torch.distributed as dist
    import seal_amd as S
    from seal_amd import shard

    rank, world, local_rank = shard.env_world()
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend="nccl", device_id=device)
    n, bits = 65536, [60] + [50] * 14 + [60]
    primes = S.CoeffModulus.Create(n, bits)
    L, K = len(primes), len(primes) - 1
    p = S.EncryptionParameters("ckks")
    p.set_poly_modulus_degree(n)
    p.set_coeff_modulus(primes)
    ctx = S.SEALContext(p, True, 0)
    ev = S.Evaluator(ctx)
    dp = shard.DigitParallel(ev, torch, dist if world > 1 else None, device)
    first, count = dp.digit_range(K)

    torch.manual_seed(0x5EA1)  # the same key and ciphertexts on every rank

    def uni(prs, prefix):
        return torch.cat([torch.randint(0, int(q), tuple(prefix) + (1, n), dtype=torch.int64, device=device) for q in prs],
                         dim=len(prefix)).contiguous()
    elt = ctx.galois_elt_from_step(1)
    glk = S.GaloisKeys(ctx)
    full_key = uni(primes, (K, 2))                       # [digit][2][L][N]
    mine = full_key[first:first + count].contiguous()     # this rank's digits only
    if count:
        # device-to-device upload of the slice; set_key_digits takes host words, so stage through the host once
        glk.set_key_digits(S.GaloisKeys.get_index(elt), first, mine.cpu().numpy().astype(np.uint64))
    else:
        glk.set_key_device(S.GaloisKeys.get_index(elt), K, full_key.data_ptr())
    del full_key, mine
    B = args.batch
    xs = uni(primes[:K], (2, B))
    x = S.Ciphertext(ctx, batch=B)
    x.resize(ctx.first_parms_id(), 2)
    x.set_is_ntt_form(True)
    x.load_device(xs.data_ptr(), xs.numel())
    work = None

    def step():
        nonlocal work
        work = x.copy()
        work.set_scale(float(primes[K - 1]) * 2.0 ** 10)
        dp.rotate_vector_inplace(work, 1, glk)
        ev.rescale_to_next_inplace(work)

    elapsed = shard.timed_steps(step, args.steps, args.warmup, dist if world > 1 else None, torch.cuda.synchronize, torch, device)
    if rank == 0:
        digest = hashlib.sha256(work.to_numpy().tobytes()).hexdigest()
        acc_bytes = 2 * (K + 1) * n * 8 * B
        print(json.dumps(dict(
            config="CKKS N=65536 L=16: rotate_vector (digit-parallel key switch) + rescale_to_next", n_gpus=world,
            batch=B, steps=args.steps, ms_per_batch=round(1e3 * elapsed / args.steps, 3),
            value=round(B * args.steps / elapsed, 1), unit="ciphertexts/s", scaling="strong",
            digits_per_rank=[shard.split(K, world, r)[1] for r in range(world)],
            all_reduce_bytes_per_key_switch=acc_bytes if world > 1 else 0, result_sha256=digest)), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
