"""Sharding of the hot path over the GPUs of one node.

1. Batch sharding of independent ciphertexts (8(e).1, the throughput mode bench.py measures): no data-path
   collective at all.
2. Digit-parallel key switching (8(e).2, BASELINE configs[4]: "decomposition parallel across GPUs"): `DigitParallel`
   below - every rank holds the same ciphertext and a slice of the key-switching key's decomposition digits and computes
   its partial sums; the exchange and the mod-down run INSIDE the library (sealhip.h section 1c: RCCL calls on the
   evaluator's stream, no host synchronisation; all-reduce or reduce-scatter + all-gather by target modulus).  Where the
   library's communicator cannot be used (gloo in the CPU tests, a process group that is not RCCL) the partial sums go
   through torch.distributed.all_reduce instead and every rank finishes locally.

One process per GPU (torch.distributed; backend "nccl" = RCCL on ROCm, "gloo" in the CPU tests).
The hot path has no exchange step between independent ciphertexts, so the only collectives are the
barrier that brackets a timed region and the max-reduction of the elapsed time; tables and keys are
built per rank.  bench.py and tests/test_dist_gloo.py share these helpers.
"""
import os
import time


def env_world():
    """(rank, world_size, local_rank) from the torchrun environment (1 process = defaults)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def split(total, world, rank):
    """Contiguous shard [start, start+count) of `total` independent items for `rank`; sizes differ by at most 1."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("rank %r outside world %r" % (rank, world))
    base, extra = divmod(total, world)
    start = rank * base + min(rank, extra)
    return start, base + (1 if rank < extra else 0)


def sync_all(dist, device_sync):
    """device sync + barrier + device sync, as the bench contract asks on both sides of the timed region"""
    device_sync()
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
        device_sync()


def timed_steps(step, steps, warmup, dist, device_sync, torch=None, device=None, local=None):
    """Run `warmup` untimed and `steps` timed calls of step(); return the MAX elapsed seconds over ranks (`local`, a dict,
    receives this rank's own elapsed time)."""
    for _ in range(warmup):
        step()
    sync_all(dist, device_sync)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync_all(dist, device_sync)
    elapsed = time.perf_counter() - t0
    if local is not None:
        local["elapsed"] = elapsed
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def whole_job_rate(items_per_rank_per_step, steps, elapsed, dist, torch=None, device=None):
    """items/s summed over all ranks (ranks may own different shard sizes)"""
    n = float(items_per_rank_per_step * steps)
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        t = torch.tensor([n], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        n = float(t.item())
    return n / elapsed


class DigitParallel:
    """Key switching with the decomposition digits spread over the ranks of `dist` (sealhip.h section 1b).

    ev: seal_amd.Evaluator of this rank; torch/dist: the initialised process group; device: where the exchange
    buffer lives (cuda:<local_rank> on the GPU box, cpu under gloo).  Every rank must call the same methods in
    the same order with equal ciphertexts; results are bit-identical to Evaluator.relinearize_inplace /
    apply_galois_inplace on one GPU.  Keys: KSwitchKeys.set_key_digits(index, *digit_range(K), full_key[range])
    keeps only this rank's slice resident (a full key works too)."""

    def __init__(self, ev, torch, dist, device, exchange="all_reduce", native=None, comm=None):
        """exchange: "all_reduce" | "reduce_scatter" (the library's two shapes, sealhip.h section 1c).
        native: None = use the library's RCCL communicator when the tensors live on a GPU and RCCL loads, else
        torch.distributed; True / False forces either."""
        self.ev, self.torch, self.dist, self.device = ev, torch, dist, device
        self.rank = dist.get_rank() if dist is not None and dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
        if self.world > 8:
            raise ValueError("at most 8 partial sums fit a 64-bit word (residues are below 2^60)")
        self._acc = None
        self.exchange = {"all_reduce": 0, "reduce_scatter": 1}[exchange]
        self.comm = comm     # a communicator shared with other DigitParallel objects of this rank (one per stream of a pipeline)
        if comm is not None:
            return
        if native is None:
            from . import api
            native = self.world > 1 and getattr(device, "type", "cpu") == "cuda" and api.Comm.rccl_available()
        if native:
            from . import api
            ids = [api.Comm.unique_id() if self.rank == 0 else None]
            if self.world > 1:
                dist.broadcast_object_list(ids, src=0)  # the 128-byte RCCL id travels over the existing process group
            self.comm = api.Comm(ids[0], self.world, self.rank)

    def digit_range(self, K):
        """(first, count) of the decomposition digits this rank serves at a level with K data primes"""
        return split(K, self.world, self.rank)

    def _buffer(self, words):
        if self._acc is None or self._acc.numel() < words:
            self._acc = self.torch.empty(words, dtype=self.torch.int64, device=self.device)
        return self._acc[:words]

    def _exchange(self, acc):
        # the partial sums were written by kernels on the evaluator's stream: make them visible to the collective
        self.ev.synchronize()
        if self.world > 1:
            if acc.is_cuda and self.dist.get_backend() == "gloo":
                # device tensors behind a host-only process group (ranks sharing one GPU in tests/test_gpu_multi.py): stage through
                # the host - the sums are what matters there, not the rate
                host = acc.cpu()
                self.dist.all_reduce(host, op=self.dist.ReduceOp.SUM)
                acc.copy_(host)
                self.torch.cuda.synchronize()
                return
            self.dist.all_reduce(acc, op=self.dist.ReduceOp.SUM)  # < 8 * 2^60 < 2^63: no overflow as int64
            if acc.is_cuda:
                self.torch.cuda.current_stream().synchronize()

    def relinearize_inplace(self, ct, relin_keys):
        """size 3 -> 2 (Evaluator::relinearize_inplace, evaluator.cpp:1144-1199)"""
        if self.comm is not None:
            return self.ev.relinearize_inplace_dp(ct, relin_keys, self.comm, self.exchange)
        if self.world == 1:   # nothing to split: the single-GPU key switch (no exchange buffer, no host synchronisation)
            return self.ev.relinearize_inplace(ct, relin_keys)
        first, count = self.digit_range(ct.coeff_modulus_size())
        acc = self._buffer(self.ev.switch_key_acc_words(ct))
        self.ev.relinearize_partial(ct, relin_keys, first, count, acc.data_ptr())
        self._exchange(acc)
        self.ev.relinearize_finish(ct, acc.data_ptr(), self.world)
        return ct

    def apply_galois_inplace(self, ct, galois_elt, galois_keys):
        """Evaluator::apply_galois_inplace (evaluator.cpp:2384-2502) on a size-2 ciphertext"""
        if self.comm is not None:
            return self.ev.apply_galois_inplace_dp(ct, galois_elt, galois_keys, self.comm, self.exchange)
        if self.world == 1:
            return self.ev.apply_galois_inplace(ct, galois_elt, galois_keys)
        first, count = self.digit_range(ct.coeff_modulus_size())
        acc = self._buffer(self.ev.switch_key_acc_words(ct))
        self.ev.apply_galois_partial(ct, galois_elt, galois_keys, first, count, acc.data_ptr())
        self._exchange(acc)
        self.ev.apply_galois_finish(ct, acc.data_ptr(), self.world)
        return ct

    def rotate_vector_inplace(self, ct, steps, galois_keys):
        """CKKS rotate_vector with the exact Galois key present (evaluator.h:1209)"""
        if steps == 0:
            return ct
        return self.apply_galois_inplace(ct, ct.context.galois_elt_from_step(steps), galois_keys)
