"""bench.py, launcher side: the command line, starting one process per GPU, bringing the process group up and PROVING that
every rank is connected before anything is timed."""
import argparse
import os
import datetime
import socket
import time
import subprocess
import sys

from .workloads import EMU, WORKLOADS


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="headline")
    ap.add_argument("--batch", type=int, default=0, help="ciphertexts per GPU per step (default: 256 for the headline workload; "
                    "SURVEY 8(d): device-resident throughput batches of 64 / 256 / 1024)")
    ap.add_argument("--total-batch", type=int, default=1024, help="bfv_c4: ciphertexts per step over ALL ranks (BASELINE configs[3])")
    ap.add_argument("--exchange", choices=["all_reduce", "reduce_scatter"], default="all_reduce",
                    help="rotate_c5: shape of the key-switch exchange (sealhip.h section 1c), RCCL calls inside the library")
    ap.add_argument("--native-comm", action="store_true", help="rotate_c5: use the library's RCCL communicator even with one rank "
                    "(exercises pack / reduce-scatter / all-gather on a single GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true", help="skip the reference check of sampled output items")
    ap.add_argument("--no-pmc", action="store_true", help="do not run the rocprofv3 PMC passes (roofline.traffic, roofline_step.kernels)")
    ap.add_argument("--no-children", action="store_true", help="headline on one GPU: do not append the BASELINE configs[3] / configs[4] "
                    "workloads (each a short child run of this script) to the line")
    ap.add_argument("--child-steps", type=int, default=5, help="timed steps of each appended workload")
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = all logical host cores")
    ap.add_argument("--cpu-reps", type=int, default=2)
    ap.add_argument("--ntt-only", action="store_true", help="only the NTT roofline leg (for rocprofv3 runs)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--step-child", action="store_true", help=argparse.SUPPRESS)  # the timed step alone, under rocprofv3 --pmc (roofline_step)
    ap.add_argument("--sweep", default="1,8,64,1024", help="headline on one GPU: batch sizes of the appended batch sweep (SURVEY 8(d): "
                    "1 = latency, 64 / 256 / 1024 = throughput); batch 1 is timed eager and as a hipGraph replay")
    ap.add_argument("--no-sweep", action="store_true", help="leave the batch sweep out (config.batch_sweep)")
    ap.add_argument("--sweep-child", action="store_true", help=argparse.SUPPRESS)  # the sweep itself: one process, one context per batch size
    ap.add_argument("--streams", type=int, default=1, help="divide the GPU's batch over this many evaluators, each on its own HIP stream, so "
                    "that one sub-batch's memory-bound phases overlap another's key switching; rotate_c5: the sub-batches share the "
                    "communicator, so that the digit-parallel exchange of one overlaps the key-switch kernels of the next")
    ap.add_argument("--graph", action="store_true", help="replay the step as one captured hipGraph (Evaluator_BeginCapture/EndCapture): "
                    "for launch-bound small batches; the default (eager) path is what the headline number uses")
    return ap.parse_args(argv)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(args, script):
    """--gpus N > 1 outside torchrun: start the N ranks (one process per GPU) and wait; rank 0 prints the JSON line."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), script] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def pin_rank_to_its_share_of_the_host(local_rank, local_world):
    """N ranks on one host: give each its own contiguous slice of the CPUs this process may run on, so that eight Python launch loops
    and eight reference-check thread pools do not fight over the same cores (a rank's kernel launches stall when its thread is
    descheduled: a one-GPU run never sees that).  Returns the CPU list, or None when nothing was changed (one rank, no
    sched_setaffinity, SEALHIP_BENCH_NO_AFFINITY=1, fewer CPUs than ranks)."""
    if local_world <= 1 or os.environ.get("SEALHIP_BENCH_NO_AFFINITY") or not hasattr(os, "sched_setaffinity"):
        return None
    try:
        cpus = sorted(os.sched_getaffinity(0))
        per = len(cpus) // local_world
        if per < 1:
            return None
        mine = cpus[local_rank * per:(local_rank + 1) * per]
        os.sched_setaffinity(0, mine)
        return mine
    except OSError:
        return None


class Ranks:
    """this process's place in the job: torch, the process group (None for one rank), device, and what the probe collective saw"""


def init_ranks(args):
    import torch
    import torch.distributed as dist
    r = Ranks()
    r.torch, r.dist = torch, dist
    r.world = int(os.environ.get("WORLD_SIZE", "1"))
    r.rank = int(os.environ.get("RANK", "0"))
    r.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if r.world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, r.world))
    r.cpu_affinity = pin_rank_to_its_share_of_the_host(r.local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", r.world)))
    r.backend = None
    # SEALHIP_BENCH_SHARE_GPU=1 (tests/test_gpu_multi.py on the one-GPU boxes): the ranks are real processes with real kernels but
    # share device 0, so the process group is gloo (RCCL refuses two ranks on one device) and the helper collectives use host
    # tensors.  Never a measurement - the line says so.
    r.shared_gpu = bool(os.environ.get("SEALHIP_BENCH_SHARE_GPU")) and not EMU
    init_timeout = datetime.timedelta(seconds=int(os.environ.get("SEALHIP_BENCH_INIT_TIMEOUT", "300")))
    if EMU:
        r.device = torch.device("cpu")
        r.dev_sync = lambda: None  # noqa: E731
        if r.world > 1:
            r.backend = "gloo"
            dist.init_process_group(backend="gloo", timeout=init_timeout)
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a MI355X: no HIP device visible (there is no CPU fallback)")
        ndev = torch.cuda.device_count()
        if ndev <= r.local_rank and not r.shared_gpu:
            raise SystemExit("bench.py: rank %d has no GPU (%d visible)" % (r.local_rank, ndev))
        dev_index = r.local_rank % ndev if r.shared_gpu else r.local_rank
        torch.cuda.set_device(dev_index)
        r.device = torch.device("cuda", dev_index)
        r.dev_sync = torch.cuda.synchronize
        if r.world > 1:
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            t0 = time.time()
            try:
                if r.shared_gpu:
                    r.backend = "gloo"
                    dist.init_process_group(backend="gloo", timeout=init_timeout)
                else:
                    r.backend = "nccl"  # = RCCL on ROCm
                    dist.init_process_group(backend="nccl", device_id=r.device, timeout=init_timeout)
            except Exception as e:
                raise SystemExit("bench.py: rank %d/%d: the %s process group did not come up within %.0f s (%r) - which rank is "
                                 "missing shows in the other ranks' messages" % (r.rank, r.world, r.backend, time.time() - t0, e))
    # helper collectives (probe, max of the elapsed time, sums, per-rank table) run on this device: the GPU with RCCL, the host with gloo
    r.coll_device = torch.device("cpu") if (EMU or r.shared_gpu) else r.device
    # one collective before anything is timed: RCCL (gloo under emulation) really connects all ranks.  Every rank contributes 1 and
    # 2^rank: the sums say how many ranks the collective reached and which; a rank that sees anything else stops the job with
    # the count in its message instead of timing a job that is not the one that was asked for.
    r.collective_ranks = 1
    if r.world > 1:
        if dist.get_world_size() != args.gpus:
            raise SystemExit("bench.py: the process group has %d ranks, --gpus %d" % (dist.get_world_size(), args.gpus))
        probe = torch.tensor([1, 1 << r.rank], dtype=torch.int64, device=r.coll_device)
        dist.all_reduce(probe)
        seen, mask = int(probe[0].item()), int(probe[1].item())
        r.collective_ranks = seen
        if seen != args.gpus or mask != (1 << args.gpus) - 1:
            raise SystemExit("bench.py: rank %d: the %s probe all-reduce reached %d rank(s) (rank mask %s), expected %d - refusing to time it"
                             % (r.rank, r.backend, seen, bin(mask), args.gpus))
    r.group = dist if r.world > 1 else None
    return r


def check_device_memory(r, need_bytes, what):
    """before anything large is allocated: this rank's free HBM against what the workload will hold (VERDICT r4 next #7a).  Prints
    one line per rank to stderr and stops the job with a clear message instead of an out-of-memory abort half-way in."""
    if EMU:
        return None
    free, total = r.torch.cuda.mem_get_info(r.device)
    share = r.world if r.shared_gpu else 1
    print("[bench] rank %d/%d on %s: %.1f GiB free of %.1f GiB; %s needs about %.1f GiB per rank%s" % (
        r.rank, r.world, r.device, free / 2 ** 30, total / 2 ** 30, what, need_bytes / 2 ** 30,
        " (x%d ranks sharing this device)" % share if share > 1 else ""), file=sys.stderr, flush=True)
    if need_bytes * share > 0.95 * free:
        raise SystemExit("bench.py: rank %d: %s needs about %.1f GiB of HBM on %s but only %.1f GiB are free - lower --batch / "
                         "--total-batch (or SEALHIP_KS_SCRATCH_CAP_MIB)" % (r.rank, what, need_bytes * share / 2 ** 30, r.device, free / 2 ** 30))
    return dict(free_bytes_before=int(free), total_bytes=int(total), estimated_need_bytes=int(need_bytes))


def gather_per_rank(r, value, ms_per_step):
    """[(rank, value, ms_per_step)] of every rank, on every rank (one all-gather outside the timed region)"""
    if r.world == 1:
        return [dict(rank=0, value=round(value, 2), ms_per_step=round(ms_per_step, 3))]
    torch, dist = r.torch, r.dist
    mine = torch.tensor([value, ms_per_step], dtype=torch.float64, device=r.coll_device)
    out = [torch.zeros_like(mine) for _ in range(r.world)]
    dist.all_gather(out, mine)
    return [dict(rank=i, value=round(float(t[0].item()), 2), ms_per_step=round(float(t[1].item()), 3)) for i, t in enumerate(out)]
