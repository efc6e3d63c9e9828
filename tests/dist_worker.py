"""Worker of tests/test_dist_gloo.py: one rank of a world_size-2 gloo job running the sharded
pipeline on the fiber-emulated library (CPU).  Every rank builds the same seeded global batch, processes
its shard (seal_amd.shard.split) and rank 0 checks the gathered results against the oracle."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def main():
    import torch
    import torch.distributed as dist
    import seal_amd as S
    from seal_amd import shard
    import parity_cases as P
    from harness import DeviceSide
    from oracle import Oracle, coeff_modulus_create, rand_ct

    S.load(os.path.join(HERE, "hipemu", "libsealhip_emu.so"))
    rank, world, _ = shard.env_world()
    dist.init_process_group(backend="gloo")
    n, bits, total = 64, [40, 30, 30, 40], 5
    primes = coeff_modulus_create(n, bits)
    K = len(primes) - 1
    o = Oracle("ckks", n, primes)
    d = DeviceSide("ckks", n, primes)
    d.upload_keys(o)
    rng = np.random.default_rng(0x5EA1)  # same global batch on every rank
    xs = [rand_ct(rng, primes, K, n) for _ in range(total)]
    ys = [rand_ct(rng, primes, K, n) for _ in range(total)]
    start, count = shard.split(total, world, rank)
    assert count >= 1
    cx, cy = d.ct(xs[start:start + count], scale=2.0 ** 10), d.ct(ys[start:start + count], scale=2.0 ** 10)
    work = S.Ciphertext(d.ctx, batch=count)

    def step():
        d.ev.multiply(cx, cy, work)
        d.ev.relinearize_inplace(work, d.rlk)
        work.set_scale(float(primes[K - 1]) * 2.0 ** 10)
        d.ev.rescale_to_next_inplace(work)

    elapsed = shard.timed_steps(step, 2, 1, dist, lambda: None, torch, torch.device("cpu"))
    rate = shard.whole_job_rate(count, 2, elapsed, dist, torch, torch.device("cpu"))
    mine = [(start + i, a) for i, a in enumerate(d.out(work))]
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    if rank == 0:
        got = dict(kv for part in gathered for kv in part)
        assert sorted(got) == list(range(total)), sorted(got)
        for i in range(total):
            exp = o.rescale(o.relinearize(o.multiply(xs[i], ys[i])))
            assert np.array_equal(got[i], exp), "item %d differs from the oracle" % i
        assert rate > 0 and elapsed > 0
        print("DIST_OK world=%d items=%d rate=%.1f" % (world, total, rate), flush=True)
    dist.barrier()

    # ---- digit-parallel key switching (seal_amd.shard.DigitParallel): the same ciphertexts on every rank, each rank
    #      holds only its slice of the key digits, one all-reduce per key switch; every rank checks against the oracle
    elt = o.galois_elt_from_step(1) if o.galois_elts else None
    dp = shard.DigitParallel(d.ev, torch, dist, torch.device("cpu"))
    first, cnt = dp.digit_range(K)
    rlk = S.RelinKeys(d.ctx)
    rlk.set_key_digits(0, first, o.relin_key()[first:first + cnt])
    x3 = [rand_ct(rng, primes, K, n, size=3) for _ in range(2)]
    c3 = d.ct(x3, scale=2.0 ** 10)
    dp.relinearize_inplace(c3, rlk)
    got = d.out(c3)
    for i in range(2):
        assert np.array_equal(got[i], o.relinearize(x3[i])), "digit-parallel relinearize item %d (rank %d)" % (i, rank)
    o2 = Oracle("ckks", n, primes, galois_elts=[o.galois_elt_from_step(1)])
    e1 = o2.galois_elt_from_step(1)
    glk = S.GaloisKeys(d.ctx)
    glk.set_key_digits(S.GaloisKeys.get_index(e1), first, o2.galois_key(e1)[first:first + cnt])
    x2 = [rand_ct(rng, primes, K, n) for _ in range(2)]
    c2 = d.ct(x2, scale=2.0 ** 10)
    dp.rotate_vector_inplace(c2, 1, glk)
    got = d.out(c2)
    for i in range(2):
        assert np.array_equal(got[i], o2.apply_galois(x2[i], e1)), "digit-parallel rotate item %d (rank %d)" % (i, rank)
    print("DIGIT_PARALLEL_OK rank=%d digits=[%d,%d)" % (rank, first, first + cnt), flush=True)
    dist.barrier()

    # ---- the same at a two-pass size (BASELINE configs[4]'s shape: rotate + rescale): `*Finish` leaves the mod-down pending and the
    #      rescale folds both divisions on every rank (sealhip.h section 1b); the all-reduce buffer is reused by the next key switch
    n2, bits2 = 8192, [50, 40, 40, 60]
    primes2 = coeff_modulus_create(n2, bits2)
    K2 = len(primes2) - 1
    probe = Oracle("ckks", n2, primes2)
    e2 = probe.galois_elt_from_step(1)
    o3 = Oracle("ckks", n2, primes2, galois_elts=[e2])
    d3 = DeviceSide("ckks", n2, primes2)
    dp3 = shard.DigitParallel(d3.ev, torch, dist, torch.device("cpu"))
    f3, c3n = dp3.digit_range(K2)
    glk3 = S.GaloisKeys(d3.ctx)
    glk3.set_key_digits(S.GaloisKeys.get_index(e2), f3, o3.galois_key(e2)[f3:f3 + c3n])
    rng3 = np.random.default_rng(0xC4)
    sc = float(primes2[K2 - 1]) * 2.0 ** 10
    folded0 = S.tail_stats()[0]
    for rnd in range(2):
        xr = [rand_ct(rng3, primes2, K2, n2)]
        cr = d3.ct(xr, scale=sc)
        dp3.rotate_vector_inplace(cr, 1, glk3)
        d3.ev.rescale_to_next_inplace(cr)
        assert np.array_equal(d3.out(cr)[0], o3.rescale(o3.apply_galois(xr[0], e2))), "digit-parallel rotate + rescale, round %d (rank %d)" % (rnd, rank)
    if not os.environ.get("SEALHIP_KS_EAGER_TAIL"):
        assert S.tail_stats()[0] - folded0 == 2, "the digit-parallel finish did not leave its tail to the rescale"
    print("DIGIT_PARALLEL_FOLDED_OK rank=%d" % rank, flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
