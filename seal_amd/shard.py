"""Batch sharding of independent ciphertexts over the GPUs of one node.

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


def timed_steps(step, steps, warmup, dist, device_sync, torch=None, device=None):
    """Run `warmup` untimed and `steps` timed calls of step(); return the MAX elapsed seconds over ranks."""
    for _ in range(warmup):
        step()
    sync_all(dist, device_sync)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync_all(dist, device_sync)
    elapsed = time.perf_counter() - t0
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
