#!/usr/bin/env python
"""bench.py — north-star benchmark of the hot path on MI355X.

metric (BASELINE.json): CKKS multiply + relinearize + rescale_to_next throughput, ciphertexts/s, at
N = 2^16, L = 16 (CoeffModulus::Create(65536, {60, 14x50, 60}), K = 15 data primes), batches of
synthetic uniform ciphertexts resident in HBM (the distribution sealbench uses, native/bench/bench.h:195-270).
A "step" = one pass of multiply_inplace + relinearize_inplace + rescale_to_next_inplace over a batch of
`--batch` ciphertexts per GPU.  N GPUs = N processes (one per GPU, torch.distributed over RCCL) each owning its
own batch, context tables and keys (batch sharding, no data-path collective: SURVEY §8(e).1) -> "scaling": "weak".

  python bench.py --gpus N          N > 1 without a torchrun environment: bench.py starts the N ranks itself
                                    (python -m torch.distributed.run, rendezvous on 127.0.0.1) and rank 0 prints the line;
                                    under torchrun (WORLD_SIZE set) it is one rank and checks WORLD_SIZE == --gpus.
  --workload headline               (default) the metric above
  --workload bfv_c4                 BASELINE configs[3]: BFV N=32768, 14x55-bit, multiply + relinearize + mod_switch_to_next,
                                    a TOTAL batch (--total-batch, default 1024) sharded over the ranks ("scaling": "strong")
  --workload rotate_c5              BASELINE configs[4]: CKKS N=65536 L=16 rotate_vector with the key-switch decomposition
                                    digits spread over the ranks (one exchange per key switch) + rescale ("strong")

One JSON line is printed by rank 0 with, besides the contract fields:
  verified_items  ciphertexts of the timed batch (16 spread evenly over every rank's share) compared word for word, outside the timed
               region, with the reference Evaluator (oracle/_ref) run on the same input and key words
  roofline     the NTT (dominant kernel family) measured live with HIP events on the stream it runs on; achieved = algorithmic
               bytes (16*N per RNS-component transform) / time; peak = 8 TB/s HBM (guide); traffic = HBM bytes
               per launch from two rocprofv3 PMC passes (FETCH_SIZE x2, WRITE_SIZE) taken by this run (traffic_source says so)
  roofline_step  the timed step itself: SURVEY 8(d)'s algorithmic bytes per ciphertext x the measured rate against the HBM peak, with
               the key counted per ciphertext and per batch; and the kernels that bound the step (key switch, BEHZ: vector ALU)
               from one live rocprofv3 --pmc pass - share of GPU time, VALU wave instructions, issue utilisation
  roofline_configs1  the same measurement at BASELINE configs[1] (CKKS N=8192, L=4: forward and inverse NTT over all RNS components)
  cpu_baseline the reference's own Evaluator (oracle/_ref = Microsoft SEAL 4.4.3, HEXL off) timed on this host's cores on a
               bounded sample, rank 0, N=1 only.  Checker/baseline only — never the thing measured.
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
EMU = bool(os.environ.get("SEALHIP_BENCH_EMU"))  # CPU tests only: fiber-emulated kernels + gloo, tiny parameters

WORKLOADS = {
    # name: (scheme, N, coeff-modulus bit sizes, plain-modulus bits, default batch per GPU)
    "headline": ("ckks", 65536, [60] + [50] * 14 + [60], 0, 256),
    "bfv_c4": ("bfv", 32768, [55] * 14, 20, 0),
    "rotate_c5": ("ckks", 65536, [60] + [50] * 14 + [60], 0, 32),
}
if EMU:
    WORKLOADS = {"headline": ("ckks", 1024, [40, 30, 30, 40], 0, 2), "bfv_c4": ("bfv", 1024, [36, 36, 37], 20, 0),
                 "rotate_c5": ("ckks", 1024, [40, 30, 30, 40], 0, 2)}


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
    ap.add_argument("--no-pmc", action="store_true", help="do not run the two rocprofv3 PMC passes for roofline.traffic")
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = all logical host cores")
    ap.add_argument("--cpu-reps", type=int, default=2)
    ap.add_argument("--ntt-only", action="store_true", help="only the NTT roofline leg (for rocprofv3 runs)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--step-child", action="store_true", help=argparse.SUPPRESS)  # the timed step alone, under rocprofv3 --pmc (roofline_step)
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


def launch_ranks(args):
    """--gpus N > 1 outside torchrun: start the N ranks (one process per GPU) and wait; rank 0 prints the JSON line."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def device_uniform(torch, primes, shape_prefix, n, device):
    """uniform residues per RNS component, generated on the device: [*prefix][len(primes)][n] int64"""
    comps = [torch.randint(0, int(q), tuple(shape_prefix) + (1, n), dtype=torch.int64, device=device) for q in primes]
    return torch.cat(comps, dim=len(shape_prefix)).contiguous()


def main():
    args = parse()
    if args.pmc_child:
        return pmc_child(args)
    if args.step_child:
        args.no_cpu_baseline = args.no_pmc = args.no_verify = True
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    import seal_amd as S
    from seal_amd import shard
    if EMU:
        S.load(os.path.join(ROOT, "tests", "hipemu", "libsealhip_emu.so"))
        device = torch.device("cpu")
        dev_sync = lambda: None  # noqa: E731
        if world > 1:
            dist.init_process_group(backend="gloo")
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a MI355X: no HIP device visible (there is no CPU fallback)")
        if torch.cuda.device_count() <= local_rank:
            raise SystemExit("bench.py: rank %d has no GPU (%d visible)" % (local_rank, torch.cuda.device_count()))
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
        dev_sync = torch.cuda.synchronize
        if world > 1:
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            dist.init_process_group(backend="nccl", device_id=device)
    if world > 1:
        assert dist.get_world_size() == args.gpus, "process group has %d ranks, --gpus %d" % (dist.get_world_size(), args.gpus)
        # one collective before anything is timed: RCCL (gloo under emulation) really connects all ranks
        probe = torch.ones(1, dtype=torch.int64, device=device)
        dist.all_reduce(probe)
        assert int(probe.item()) == args.gpus, "all-reduce saw %d ranks, expected %d" % (int(probe.item()), args.gpus)
    group = dist if world > 1 else None

    scheme, n, bits, tbits, default_batch = WORKLOADS[args.workload]
    primes = S.CoeffModulus.Create(n, bits)
    L, K = len(primes), len(primes) - 1
    parms = S.EncryptionParameters(scheme)
    parms.set_poly_modulus_degree(n)
    parms.set_coeff_modulus(primes)
    t_plain = 0
    if scheme != "ckks":
        t_plain = S.PlainModulus.Batching(n, tbits)
        parms.set_plain_modulus(t_plain)
    ctx = S.SEALContext(parms, True, 0)  # sec_level_type::none, as sealbench (native/bench/bench.h:35-36)
    ev = S.Evaluator(ctx)
    first = ctx.first_parms_id()

    scaling = "weak"
    if args.workload == "bfv_c4":
        # BASELINE configs[3]: a fixed total batch sharded over the ranks, no data-path collective
        start, B = shard.split(args.total_batch if not EMU else 4, world, rank)
        scaling = "strong"
    else:
        B = args.batch or default_batch
        start = rank * B
    if args.workload == "rotate_c5":
        scaling = "strong"  # every rank holds the same batch; the key-switch digits are divided over the ranks

    # ---- synthetic keys: K digits x 2 polys x L comps, uniform per component (C5: 240 MiB)
    same_on_all_ranks = args.workload == "rotate_c5"
    torch.manual_seed(0x5EA1 + (0 if same_on_all_ranks else rank))
    key = device_uniform(torch, primes, (K, 2), n, device)
    key_host = None
    want_verify = not args.no_verify and not args.ntt_only and reference_available()
    if want_verify:
        key_host = key.cpu().numpy().view("uint64")
    dp = None
    if args.workload == "rotate_c5":
        elt = ctx.galois_elt_from_step(1)
        keys = S.GaloisKeys(ctx)
        dp = shard.DigitParallel(ev, torch, group, device, exchange=args.exchange, native=True if args.native_comm else None)
        d0, dc = dp.digit_range(K)
        if dp.comm is not None:
            # one-time key distribution inside the library: rank 0's key is broadcast over RCCL and every rank keeps its own
            # digits resident (Evaluator_BroadcastKeyDigits); the other ranks' tensors are only the receive buffers
            if rank != 0:
                key.zero_()
            dev_sync()
            ev.broadcast_key_digits(keys, S.GaloisKeys.get_index(elt), key.data_ptr(), dp.comm, 0)
        elif world > 1 and dc:
            keys.set_key_digits(S.GaloisKeys.get_index(elt), d0, key[d0:d0 + dc].cpu().numpy().view("uint64"))
        else:
            keys.set_key_device(S.GaloisKeys.get_index(elt), K, key.data_ptr())
    else:
        keys = S.RelinKeys(ctx)
        keys.set_key_device(0, K, key.data_ptr())
    del key

    # ---- synthetic size-2 ciphertext batches at the first data level (CKKS: NTT form, scale 2^24)
    ntt_form = scheme != "bfv"
    scale = 2.0 ** (50 // 2 - 1) if scheme == "ckks" else 1.0
    xs = device_uniform(torch, primes[:K], (2, B), n, device)
    ys = device_uniform(torch, primes[:K], (2, B), n, device) if args.workload != "rotate_c5" else None

    def make_ct(t):
        ct = S.Ciphertext(ctx, batch=B)
        ct.resize(first, 2)
        ct.set_is_ntt_form(ntt_form)
        ct.set_scale(scale)
        ct.load_device(t.data_ptr(), t.numel())
        return ct

    x = make_ct(xs)
    y = make_ct(ys) if ys is not None else None
    work = S.Ciphertext(ctx, batch=B)
    dev_sync()

    lanes = None
    if args.streams > 1 and B >= args.streams:
        # sub-batches [lo, hi) of the resident inputs, one evaluator + stream + output batch each
        lanes = []
        for si in range(args.streams):
            lo, cnt = shard.split(B, args.streams, si)
            st = S.Stream()
            e = S.Evaluator(ctx)
            e.set_stream(st.handle)

            def sub(t, lo=lo, cnt=cnt):
                ct = S.Ciphertext(ctx, batch=cnt)
                ct.resize(first, 2)
                ct.set_is_ntt_form(ntt_form)
                ct.set_scale(scale)
                src = t[:, lo:lo + cnt].contiguous()   # [2][cnt][K][n]
                ct.load_device(src.data_ptr(), src.numel())
                dev_sync()
                return ct
            lane = dict(ev=e, stream=st, lo=lo, cnt=cnt, x=sub(xs), y=sub(ys) if ys is not None else None, work=S.Ciphertext(ctx, batch=cnt))
            if dp is not None:
                # rotate_c5: every sub-batch has its own evaluator / stream and shares the communicator, so the exchange of
                # sub-batch i (a collective queued on stream i) runs while sub-batch i + 1 is still in its key-switch kernels
                lane["dp"] = shard.DigitParallel(e, torch, group, device, exchange=args.exchange, comm=dp.comm, native=dp.comm is not None)
            lanes.append(lane)
        dev_sync()

    last_op = {"headline": "rescale_to_next_inplace", "bfv_c4": "mod_switch_to_next_inplace"}.get(args.workload)
    if lanes and args.workload == "rotate_c5":
        rot_scale = float(primes[K - 1]) * 2.0 ** 10

        def step():
            for ln in lanes:
                w = ln["ev"].copy_to(ln["x"], ln["work"])   # the rotation works in place: stage the resident input on the lane's stream
                w.set_scale(rot_scale)
                ln["dp"].rotate_vector_inplace(w, 1, keys)
                ln["ev"].rescale_to_next_inplace(w)
    elif lanes:
        def step():
            for ln in lanes:
                ln["ev"].multiply(ln["x"], ln["y"], ln["work"])
            for ln in lanes:
                ln["ev"].relinearize_inplace(ln["work"], keys)
            for ln in lanes:
                getattr(ln["ev"], last_op)(ln["work"])
    elif args.workload == "headline":
        def step():
            ev.multiply(x, y, work)          # work = x * y (size 3); x stays resident as the next step's input
            ev.relinearize_inplace(work, keys)
            ev.rescale_to_next_inplace(work)
    elif args.workload == "bfv_c4":
        def step():
            ev.multiply(x, y, work)
            ev.relinearize_inplace(work, keys)
            ev.mod_switch_to_next_inplace(work)
    else:
        rot_scale = float(primes[K - 1]) * 2.0 ** 10
        holder = {}

        def step():
            w = x.copy()                    # device-to-device copy of the resident batch (the rotation works in place)
            w.set_scale(rot_scale)
            dp.rotate_vector_inplace(w, 1, keys)
            ev.rescale_to_next_inplace(w)
            holder["work"] = w

    if args.graph and lanes:
        raise SystemExit("bench.py: --graph captures one evaluator's stream; not combined with --streams")
    if args.graph and not args.ntt_only and args.workload != "rotate_c5":
        step()  # eager once: lazily built tables, pool warm-up
        dev_sync()
        graph = ev.capture(step)
        step = graph.launch

    result = {}
    verified = None
    if not args.ntt_only:
        elapsed = shard.timed_steps(step, args.steps, args.warmup, group, dev_sync, torch, device)
        if args.workload == "rotate_c5" and not lanes:
            work = holder["work"]
        if lanes:
            work = LaneView(lanes)
        assert work.size() == 2 and work.coeff_modulus_size() == K - 1 and work.batch() == B
        per_step = B if args.workload != "rotate_c5" else float(B) / world  # rotate_c5: all ranks worked on the same B items
        rate = shard.whole_job_rate(per_step, args.steps, elapsed, group, torch, device)
        result = dict(value=rate, ms_per_step=1e3 * elapsed / args.steps)
        if want_verify and B > 0:
            verified = verify_items(args.workload, scheme, n, primes, t_plain, key_host, xs, ys, work, B, scale)
            if group is not None:
                v = torch.tensor([verified], dtype=torch.int64, device=device)
                dist.all_reduce(v)
                verified = int(v.item())
    del key_host

    # ---- roofline leg: the batched forward NTT over the resident batch (2*B polys x K comps)
    roofline = None
    if rank == 0 and not EMU and not args.step_child:
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
        roofline = dict(bound="hbm", kernel="ntt_forward over %d transforms of 2^%d per launch" % (K * polys, n.bit_length() - 1),
                        achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 4),
                        traffic=None, ms_per_launch=round(ms, 4), algorithmic_bytes_per_launch=alg_bytes)
        assert buf_words == 2 * B * K * n

    ntt_c1 = None
    if rank == 0 and world == 1 and args.workload == "headline" and not EMU and not args.step_child:
        ntt_c1 = ntt_configs1(S, torch, device)

    # free the device before the PMC child processes and the CPU baseline start
    if world > 1:
        dist.barrier()
    if rank == 0 and world == 1 and roofline is not None and not args.no_pmc:
        del x, y, work, xs, ys
        lanes = None
        torch.cuda.empty_cache()
        S.release_pool()
        roofline.update(pmc_traffic(args, B, K, n))

    roofline_step = None
    if rank == 0 and result and not EMU and not args.step_child:
        # the whole step against the memory roofline in both accountings, and the kernels that actually bound it (vector ALU)
        sb = step_bytes(args.workload, K, L, n)
        cts_per_s_per_gpu = result["value"] / world
        roofline_step = dict(
            bound="valu", note="the step is bound by vector-ALU issue in the key-switch (and BEHZ) kernels, not by HBM: the memory "
            "fractions below are what the algorithmic bytes amount to, the kernel table says how busy the vector ALU is",
            algorithmic_bytes_per_ciphertext=sb, peak=HBM_PEAK_GBS, unit="GB/s",
            achieved={k: round(v * cts_per_s_per_gpu / 1e9, 1) for k, v in sb.items()},
            frac={k: round(v * cts_per_s_per_gpu / 1e9 / HBM_PEAK_GBS, 4) for k, v in sb.items()})
        if world == 1 and not args.no_pmc:
            roofline_step.update(step_counters(args, B))

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.ntt_only and not EMU:
        cpu = cpu_baseline(args.workload, scheme, n, primes, t_plain, args)

    tail_note = None
    if args.workload != "bfv_c4":
        folded, plain, dropped = S.tail_stats()
        tail_note = ("deferred: relinearize leaves the mod-down by the special prime to the rescale that follows, which does both "
                     "rounding divisions with one transform per component (same words; %d folded / %d separate / %d discarded in "
                     "this process)" % (folded, plain, dropped)) if folded else "separate pass (%d tails completed on their own)" % plain
    if rank == 0:
        names = {
            "headline": ("CKKS multiply+relinearize+rescale ciphertexts/sec @ N=2^16, L=16",
                         "CKKS N=65536, coeff_modulus {60,14x50,60} (L=16, K=15): multiply_inplace + relinearize_inplace + "
                         "rescale_to_next_inplace, device-resident batches"),
            "bfv_c4": ("BFV multiply+relinearize+mod_switch ciphertexts/sec @ N=32768, 14 primes",
                       "BASELINE configs[3]: BFV N=32768, 14x55-bit chain, t=Batching(32768,20): multiply + relinearize + "
                       "mod_switch_to_next, total batch %d sharded over the ranks" % args.total_batch),
            "rotate_c5": ("CKKS rotate_vector+rescale ciphertexts/sec @ N=2^16, L=16, digit-parallel key switch",
                          "BASELINE configs[4]: CKKS N=65536 L=16 rotate_vector (decomposition digits spread over the ranks, one "
                          "exchange of 2(K+1)N words per ciphertext) + rescale_to_next"),
        }[args.workload]
        par = {"headline": "batch-sharded x%d, no data-path collective" % world,
               "bfv_c4": "total batch sharded x%d, no data-path collective" % world,
               "rotate_c5": "key-switch digits split x%d, exchange %s per key switch (%s)" % (
                   world, args.exchange, "RCCL inside libsealhip" if dp is not None and dp.comm is not None else "torch.distributed")}[args.workload]
        line = dict(
            metric=names[0], value=round(result.get("value", 0.0), 2), unit="ciphertexts/s", n_gpus=world, steps=args.steps,
            warmup=args.warmup, ms_per_step=round(result.get("ms_per_step", 0.0), 3), higher_is_better=True,
            scaling=scaling, vs_baseline=None, dtype="u64", data="synthetic", verified_items=verified,
            config=dict(workload=names[1] + (" [EMULATED KERNELS, CPU test]" if EMU else ""),
                        batch_per_gpu=B, key_switch_tail=tail_note, launch=("hipGraph replay" if args.graph else "eager") + (
                            ", %d evaluators x %d-item sub-batches on %d HIP streams" % (len(lanes), lanes[0]["cnt"], len(lanes)) if lanes else ""),
                        parallelism=par,
                        arithmetic="64-bit residues: exact double-precision (error-free FMA) arithmetic for primes below "
                                   "2^50, 64-bit Shoup/Barrett integer arithmetic for larger primes; results canonical u64",
                        key_bytes=2 * K * L * n * 8, algorithmic_bytes_per_ciphertext=(2 * K * K + 18 * K - 2) * 8 * n,
                        **(dict(exchange_bytes_per_ciphertext=2 * (K + 1) * n * 8,
                                exchange_overlap=("%d sub-batches on %d streams sharing the communicator: the exchange of one runs while the "
                                                  "next is in its key-switch kernels" % (len(lanes), len(lanes))) if lanes else
                                "none: one batch, kernels and exchange in stream order (--streams S pipelines S sub-batches)")
                           if args.workload == "rotate_c5" else {})),
            roofline=roofline, roofline_step=roofline_step, roofline_configs1=ntt_c1, cpu_baseline=cpu)
    # RCCL prints a version banner through C stdio when a communicator comes up; it sits in the C buffer until exit.  Tear
    # the process group down and flush the C streams first, so that the JSON line is the LAST thing on stdout.
    del dp
    if world > 1:
        dist.destroy_process_group()
    try:
        C.CDLL(None).fflush(None)
    except Exception:
        pass
    if rank == 0:
        print(json.dumps(line), flush=True)


# --------------------------------------------------------------------------------------------------------------------
def reference_available():
    try:
        import sealref
        return sealref.available()
    except Exception:
        return False


class LaneView:
    """the per-stream output batches of --streams seen as one batch (metadata of lane 0, items by global index)"""

    def __init__(self, lanes):
        self.lanes = lanes

    def size(self):
        sizes = {ln["work"].size() for ln in self.lanes}
        assert len(sizes) == 1
        return sizes.pop()

    def coeff_modulus_size(self):
        return self.lanes[0]["work"].coeff_modulus_size()

    def batch(self):
        return sum(ln["work"].batch() for ln in self.lanes)

    def scale(self):
        return self.lanes[0]["work"].scale()

    def item_to_numpy(self, b):
        for ln in self.lanes:
            if ln["lo"] <= b < ln["lo"] + ln["cnt"]:
                return ln["work"].item_to_numpy(b - ln["lo"])
        raise IndexError(b)


def verify_items(workload, scheme, n, primes, t_plain, key_host, xs, ys, work, B, scale, count=16):
    """`count` items spread evenly over this rank's timed batch (first and last included) against seal::Evaluator (oracle/_ref)
    on the same words, the reference running on host threads (ctypes releases the GIL).  Raises on the first differing word;
    returns the number of items compared.  Outside the timed region."""
    import numpy as np
    import sealref
    from concurrent.futures import ThreadPoolExecutor
    K = len(primes) - 1
    ref = sealref.RefContext(scheme, n, primes, t_plain)
    take = min(count, B)
    items = sorted({int(round(i * (B - 1) / max(1, take - 1))) for i in range(take)})
    if workload == "rotate_c5":
        ref.keygen_galois_steps([1])
        elt = ref.galois_elt_from_step(1)
        ref.set_key("galois", (elt - 1) >> 1, key_host)
    else:
        ref.keygen_relin()
        ref.set_key("relin", 0, key_host)
    ci = ref.first_chain_index

    def expected(b):
        xw = xs[:, b].cpu().numpy().view("uint64")
        if workload == "rotate_c5":
            a = ref.ct(ci, xw, True, float(primes[K - 1]) * 2.0 ** 10)
            ref.rotate_vector_inplace(a, 1)
            ref.rescale_to_next_inplace(a)
        else:
            yw = ys[:, b].cpu().numpy().view("uint64")
            a, c = ref.ct(ci, xw, scheme != "bfv", scale), ref.ct(ci, yw, scheme != "bfv", scale)
            ref.multiply_inplace(a, c)
            ref.relinearize_inplace(a)
            if workload == "headline":
                ref.rescale_to_next_inplace(a)
            else:
                ref.mod_switch_to_next_inplace(a)
        return a.data(), a.info()["scale"]

    inputs_ready = [(b, work.item_to_numpy(b)) for b in items]      # device reads on this thread
    with ThreadPoolExecutor(max_workers=min(len(items), os.cpu_count() or 1)) as pool:
        results = list(pool.map(expected, items))
    for (b, got), (exp, ref_scale) in zip(inputs_ready, results):
        if got.shape != exp.shape or not np.array_equal(got, exp):
            raise SystemExit("bench.py: item %d of the timed batch differs from the reference Evaluator" % b)
        if scheme == "ckks" and work.scale() != ref_scale:
            raise SystemExit("bench.py: scale metadata differs from the reference (%r vs %r)" % (work.scale(), ref_scale))
    return len(items)


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
        out.append(dict(chain=label, transforms_per_launch=comps * polys, **rates))
        del data
    return dict(bound="hbm", unit="GB/s", peak=HBM_PEAK_GBS, workload="CKKS N=8192, L=4: batched NTT / INTT over all RNS components", chains=out)


# ---- roofline.traffic: HBM bytes of one ntt_forward launch, measured by this run -----------------------------------
PMC_CALLS = 3


def pmc_child(args):
    """Run under `rocprofv3 --kernel-trace --pmc <counter>`: the roofline leg's launch (same shape), no torch, PMC_CALLS calls."""
    import seal_amd as S
    scheme, n, bits, tbits, default_batch = WORKLOADS[args.workload]
    primes = S.CoeffModulus.Create(n, bits)
    K = len(primes) - 1
    parms = S.EncryptionParameters(scheme)
    parms.set_poly_modulus_degree(n)
    parms.set_coeff_modulus(primes)
    if scheme != "ckks":
        parms.set_plain_modulus(S.PlainModulus.Batching(n, tbits))
    ctx = S.SEALContext(parms, True, 0)
    polys = 2 * (args.batch or default_batch)
    buf = S.DeviceBuffer(polys * K * n)  # contents do not matter for the byte counters
    for _ in range(PMC_CALLS):
        S.ntt_forward(ctx, buf, polys, K)
    S.device_synchronize()
    return 0


def pmc_traffic(args, B, K, n):
    """Two separate rocprofv3 passes (FETCH_SIZE, WRITE_SIZE: they do not fit one pass, MI355X_MICROARCH.md §PMC slots) over a
    child process that issues the roofline leg's launch; FETCH_SIZE is doubled (gfx950 tallies 128-B requests at 64 B, same
    guide §HBM).  Returns {'traffic': bytes per launch or None, 'traffic_source': how it was obtained}."""
    import glob
    import shutil
    import sqlite3
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return dict(traffic=None, traffic_source="rocprofv3 not found on this host")
    totals = {}
    per_kernel = {}
    tmp = tempfile.mkdtemp(prefix="sealhip_pmc_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = [exe, "--kernel-trace", "--pmc", counter, "-d", out, "-o", "r", "--", sys.executable, os.path.abspath(__file__),
                   "--pmc-child", "--batch", str(B), "--workload", args.workload]
            env = dict(os.environ, TMPDIR="/tmp")
            p = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=420)
            dbs = glob.glob(os.path.join(out, "**", "*.db"), recursive=True)
            if p.returncode != 0 or not dbs:
                return dict(traffic=None, traffic_source="rocprofv3 --pmc %s failed (rc %d): %s" % (counter, p.returncode, (p.stderr or p.stdout)[-300:]))
            cur = sqlite3.connect(dbs[0]).cursor()
            rows = cur.execute("select kernel_name, count(*), sum(value) from counters_collection where counter_name = ? "
                               "group by kernel_name", (counter,)).fetchall()
            tot = 0.0
            for name, calls, val in rows:
                if "ntt" not in name:
                    continue
                kib = float(val) / PMC_CALLS
                short = name.replace("sealhip::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
                per_kernel.setdefault(short, {})[counter] = round(kib * (2 if counter == "FETCH_SIZE" else 1), 1)
                tot += kib
            totals[counter] = tot * 1024.0
        traffic = int(round(2.0 * totals["FETCH_SIZE"] + totals["WRITE_SIZE"]))
        return dict(traffic=traffic, traffic_source="live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this run "
                    "(KiB per launch per kernel, FETCH doubled): %s" % json.dumps(per_kernel, sort_keys=True),
                    traffic_over_algorithmic=round(traffic / (16.0 * n * K * 2 * B), 3))
    except Exception as e:  # the counters must never take the benchmark down
        return dict(traffic=None, traffic_source="PMC passes failed: %r" % (e,))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


# ---- roofline_step: the kernels that dominate the timed step, by this run's own counters --------------------------------
VALU_CYCLES_PER_WAVE_INST = 4      # a wave64 VALU instruction occupies a SIMD16 for four cycles (MI355X_MICROARCH.md; measured 4.2-5)
SIMDS, ENGINE_HZ = 256 * 4, 2.4e9


def step_counters(args, B):
    """One rocprofv3 --kernel-trace --pmc pass over a child that runs the timed step alone (a smaller batch: the per-dispatch
    figures scale with it, the utilisation does not once the chip is full).  Per kernel: share of the step's GPU time, wave
    instructions on the vector ALU per dispatch, and issue utilisation = those x 4 cycles / (duration x 1024 SIMDs x 2.4 GHz)."""
    import glob
    import shutil
    import sqlite3
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return dict(source="rocprofv3 not found on this host")
    child_batch = max(1, min(B, 64))
    tmp = tempfile.mkdtemp(prefix="sealhip_step_", dir="/tmp")
    try:
        cmd = [exe, "--kernel-trace", "--pmc", "SQ_INSTS_VALU", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "-d", tmp, "-o", "r", "--",
               sys.executable, os.path.abspath(__file__), "--step-child", "--workload", args.workload, "--batch", str(child_batch),
               "--total-batch", str(child_batch), "--steps", "2", "--warmup", "1"]
        p = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=600)
        dbs = glob.glob(os.path.join(tmp, "**", "*.db"), recursive=True)
        if p.returncode != 0 or not dbs:
            return dict(source="rocprofv3 --pmc pass failed (rc %d): %s" % (p.returncode, (p.stderr or p.stdout)[-300:]))
        cur = sqlite3.connect(dbs[0]).cursor()
        rows = cur.execute("select kernel_name, counter_name, count(*), sum(value), sum(duration) from counters_collection "
                           "group by kernel_name, counter_name").fetchall()
        table = {}
        for name, ctr, cnt, val, dur in rows:
            if "sealhip" not in name:
                continue  # torch's input generation, copies
            short = name.replace("sealhip::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            e = table.setdefault(short, dict(dispatches=cnt, ns=float(dur)))
            e[ctr] = float(val)
        total_ns = sum(e["ns"] for e in table.values()) or 1.0
        out = []
        for k, e in sorted(table.items(), key=lambda kv: -kv[1]["ns"])[:8]:
            insts = e.get("SQ_INSTS_VALU", 0.0)
            util = insts * VALU_CYCLES_PER_WAVE_INST / (e["ns"] * 1e-9 * SIMDS * ENGINE_HZ) if e["ns"] else 0.0
            row = dict(kernel=k, share_of_gpu_time=round(e["ns"] / total_ns, 3), dispatches=e["dispatches"],
                       avg_ms=round(e["ns"] / e["dispatches"] / 1e6, 4), valu_wave_insts_per_dispatch=int(insts / e["dispatches"]),
                       valu_issue_utilisation=round(util, 3))
            if e.get("SQ_WAVE_CYCLES"):
                row["waiting_to_issue_frac_of_wave_cycles"] = round(e.get("SQ_WAIT_INST_ANY", 0.0) / e["SQ_WAVE_CYCLES"], 3)
            out.append(row)
        return dict(source="live: one rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY pass over "
                           "`bench.py --step-child --batch %d` (3 steps); utilisation = VALU wave instructions x %d cycles / (kernel time "
                           "x %d SIMDs x %.1f GHz)" % (child_batch, VALU_CYCLES_PER_WAVE_INST, SIMDS, ENGINE_HZ / 1e9),
                    kernels=out)
    except Exception as e:  # the counters must never take the benchmark down
        return dict(source="PMC pass failed: %r" % (e,))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def step_bytes(workload, K, L, n):
    """algorithmic bytes of one ciphertext through the timed step (words of 8 bytes; twiddles and scratch excluded),
    with the switching key read once per ciphertext / once per batch (what the kernels do: it stays in L2 for the whole batch)."""
    key = 2 * K * L * n * 8
    if workload == "rotate_c5":
        total = (2 * K * K + 10 * K - 2) * 8 * n          # apply_galois + key switch + rescale
    else:
        total = (2 * K * K + 18 * K - 2) * 8 * n          # multiply + relinearize + rescale / mod_switch
    out = dict(with_key=total, key_amortised=total - key)
    if workload == "bfv_c4":
        # BEHZ multiply is transform-heavy: (8K+4) forward + (6K+3) inverse transforms of 16 N bytes each, K(K+1) more in the key switch
        out["ntt_equivalent"] = ((8 * K + 4) + (6 * K + 3) + K * (K + 1) + 2 * K) * 16 * n
    return out


# ---- CPU baseline --------------------------------------------------------------------------------------------------
def physical_cores():
    try:
        import psutil
        return psutil.cpu_count(logical=False) or (os.cpu_count() or 1)
    except Exception:
        return os.cpu_count() or 1


def cpu_baseline(workload, scheme, n, primes, t_plain, args):
    pipeline = {"headline": "ckks_mul_relin_rescale", "bfv_c4": "bfv_mul_relin_modswitch", "rotate_c5": "rotate"}[workload]
    try:
        import sealref
        if sealref.available():
            logical = args.cpu_threads or (os.cpu_count() or 1)
            phys = min(physical_cores(), logical)
            ref = sealref.RefContext(scheme, n, primes, t_plain)
            ref.keygen_relin()
            if pipeline == "rotate":
                ref.keygen_galois_steps([1])
            reps = args.cpu_reps
            secs = ref.time_pipeline(pipeline, logical, reps)
            out = dict(value=round(logical * reps / secs, 3), unit="ciphertexts/s", cores=logical, kind="reference",
                       sample="%d threads x %d ciphertexts each; every thread builds its inputs, runs one untimed pass, waits at a "
                              "start barrier; wall time from the barrier to the last thread's finish; per-thread "
                              "MemoryPoolHandle::New(); seal::Evaluator, HEXL off, same parameters" % (logical, reps))
            if phys != logical:
                secs_p = ref.time_pipeline(pipeline, phys, reps)
                out["physical_cores_run"] = dict(value=round(phys * reps / secs_p, 3), cores=phys)
            one = ref.time_pipeline(pipeline, 1, 2)
            out["single_thread_value"] = round(2 / one, 3)
            return out
    except Exception as e:  # the baseline must never take the benchmark down
        sys.stderr.write("cpu_baseline(reference) unavailable: %r\n" % (e,))
    if workload != "headline":
        return None
    try:
        import numpy as np
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
