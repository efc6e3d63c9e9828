"""One rank of tests/test_gpu_multi.py: digit-parallel key switching over REAL ranks (one process per GPU, the library's RCCL
communicator, no torch).  argv: rank, nranks, file holding the 128-byte RCCL id (written by rank 0), exchange (0 | 1)."""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def main():
    rank, nranks, id_path, exchange = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
    import seal_amd as S
    S.load()
    S.set_device(rank)
    from harness import DeviceSide
    from oracle import Oracle, coeff_modulus_create, rand_ct
    if rank == 0:
        uid = S.Comm.unique_id()
        with open(id_path + ".tmp", "wb") as f:
            f.write(uid)
        os.replace(id_path + ".tmp", id_path)
    else:
        for _ in range(600):
            if os.path.exists(id_path):
                break
            time.sleep(0.1)
        uid = open(id_path, "rb").read()
    # ncclCommInitRank blocks until every rank has arrived: time-box it and say WHICH rank is stuck (VERDICT r4 next #7c)
    import threading
    box = {}

    def bring_up():
        try:
            box["comm"] = S.Comm(uid, nranks, rank)
        except Exception as e:  # reported by the main thread
            box["error"] = e
    th = threading.Thread(target=bring_up, daemon=True)
    th.start()
    th.join(float(os.environ.get("SEALHIP_COMM_INIT_TIMEOUT", "180")))
    if th.is_alive():
        print("MULTI_GPU_HUNG rank=%d/%d: ncclCommInitRank did not return within the time box (the ranks that print nothing "
              "never reached it)" % (rank, nranks), flush=True)
        os._exit(3)
    if "error" in box:
        raise box["error"]
    comm = box["comm"]
    assert not comm.loopback()
    for n, bits, batch in ((8192, [60, 40, 40, 50, 60], 2), (65536, [60] + [50] * 14 + [60], 1)):
        primes = coeff_modulus_create(n, bits)
        K = len(primes) - 1
        probe = Oracle("ckks", n, primes)
        elt = probe.galois_elt_from_step(1)
        o = Oracle("ckks", n, primes, galois_elts=[elt])      # same seed on every rank: the same keys
        d = DeviceSide("ckks", n, primes)
        first, count = comm.digit_range(K)
        # rank 0 holds the keys; every rank receives its digits through the library's broadcast
        rlk, glk = S.RelinKeys(d.ctx), S.GaloisKeys(d.ctx)
        L = len(primes)
        for keys, index, words in ((rlk, 0, o.relin_key()), (glk, S.GaloisKeys.get_index(elt), o.galois_key(elt))):
            stage = S.DeviceBuffer.from_numpy(words if rank == 0 else np.zeros((K, 2, L, n), dtype=np.uint64))
            d.ev.broadcast_key_digits(keys, index, stage.ptr, comm, 0)
        rng = np.random.default_rng(77)                        # the same ciphertexts on every rank
        x3 = [rand_ct(rng, primes, K, n, size=3) for _ in range(batch)]
        x2 = [rand_ct(rng, primes, K, n) for _ in range(batch)]
        c3, c2 = d.ct(x3), d.ct(x2)
        d.ev.relinearize_inplace_dp(c3, rlk, comm, exchange)
        d.ev.rotate_vector_inplace_dp(c2, 1, glk, comm, exchange)
        g3, g2 = d.out(c3), d.out(c2)
        for b in range(batch):
            assert np.array_equal(g3[b], o.relinearize(x3[b])), "relinearize n=%d item %d rank %d" % (n, b, rank)
            assert np.array_equal(g2[b], o.apply_galois(x2[b], elt)), "rotate n=%d item %d rank %d" % (n, b, rank)
    print("MULTI_GPU_OK rank=%d/%d exchange=%d digits=[%d,%d)" % (rank, nranks, exchange, first, first + count), flush=True)


if __name__ == "__main__":
    main()
