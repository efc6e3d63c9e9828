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
  workloads    headline on one GPU only: BASELINE configs[3] (bfv_c4) and configs[4] (rotate_c5) timed by short child runs of this
               script after the headline (value, ms_per_step, verified_items, roofline of each); --no-children leaves them out
  config.batch_sweep  headline on one GPU only: the same step at SURVEY 8(d)'s other batch sizes (--sweep, default 1, 8, 64, 1024; batch 1
               eager and as a hipGraph replay = per-ciphertext latency) from one child process; --no-sweep leaves it out
  config.device_memory  bytes the library's pool held after the timed steps, the largest key-switch intermediate (the batch runs in
               chunks on forked streams: sealhip.h "Chunked key switching"), how many key switches ran chunked and in how many chunks
  rccl_ranks, per_rank  how many ranks the probe all-reduce reached before anything was timed (a mismatch stops the job with
               the count in the message) and every rank's own rate
  cpu_baseline the reference's own Evaluator (oracle/_ref = Microsoft SEAL 4.4.3, HEXL off) timed on this host's cores on a
               bounded sample, rank 0, N=1 only.  Checker/baseline only — never the thing measured.
"""
import ctypes as C
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from benchlib import counters, cpu, launcher, workloads  # noqa: E402
from benchlib.counters import HBM_PEAK_GBS  # noqa: E402
from benchlib.workloads import EMU  # noqa: E402


def main():
    args = launcher.parse()
    if args.pmc_child:
        return counters.pmc_child(args)
    if args.step_child:
        args.no_cpu_baseline = args.no_pmc = args.no_verify = args.no_children = True
    if args.sweep_child:
        return sweep_child(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launcher.launch_ranks(args, os.path.abspath(__file__)))
    r = launcher.init_ranks(args)
    torch, dist, world, rank, device, dev_sync, group = r.torch, r.dist, r.world, r.rank, r.device, r.dev_sync, r.group
    import seal_amd as S
    from seal_amd import shard
    if EMU:
        S.load(os.path.join(ROOT, "tests", "hipemu", "libsealhip_emu.so"))

    cdev = r.coll_device   # where the helper collectives' tensors live (the GPU with RCCL, the host with gloo)
    mem_check = launcher.check_device_memory(r, workloads.estimate_device_bytes(args, world, rank), "--workload %s" % args.workload)
    w = workloads.build(args, S, shard, torch, group, device, dev_sync, world, rank, shared_gpu=r.shared_gpu)
    n, K, L, B, scaling = w.n, w.K, w.L, w.B, w.scaling

    result = {}
    verified = None
    per_rank = None
    if not args.ntt_only:
        local = {}
        elapsed = shard.timed_steps(w.step, args.steps, args.warmup, group, dev_sync, torch, cdev, local=local)
        work = workloads.result_batch(w, args) if B > 0 else None
        assert B == 0 or (work.size() == 2 and work.coeff_modulus_size() == K - 1 and work.batch() == B)
        per_step = B if args.workload != "rotate_c5" else float(B) / world  # rotate_c5: all ranks worked on the same B items
        rate = shard.whole_job_rate(per_step, args.steps, elapsed, group, torch, cdev)
        result = dict(value=rate, ms_per_step=1e3 * elapsed / args.steps)
        per_rank = launcher.gather_per_rank(r, per_step * args.steps / local["elapsed"], 1e3 * local["elapsed"] / args.steps)
        if group is not None and w.want_verify is False and not args.no_verify and workloads.reference_available():
            verified = 0   # a rank with an empty shard has nothing to compare but joins the sum below
        if w.want_verify and B > 0:
            verified = workloads.verify_items(args.workload, w.scheme, n, w.primes, w.t_plain, w.key_host, w.xs, w.ys, work, B, w.scale)
        if group is not None and verified is not None:
            v = torch.tensor([verified], dtype=torch.int64, device=cdev)
            dist.all_reduce(v)
            verified = int(v.item())
        del work
    w.key_host = None

    # ---- roofline leg: the batched forward NTT over the resident batch (2*B polys x K comps)
    roofline = None
    if rank == 0 and not EMU and not args.step_child:
        roofline = counters.ntt_leg(S, w.ctx, w.xs, B, K, n)

    ntt_c1 = None
    if rank == 0 and world == 1 and args.workload == "headline" and not EMU and not args.step_child:
        ntt_c1 = counters.ntt_configs1(S, torch, device)

    # what the step held in HBM: the pool's blocks and the largest key-switch intermediate of the timed steps
    ks_calls, ks_chunks, ks_scratch = S.ks_chunk_stats()
    memory_note = dict(pool_bytes_held=S.pool_stats()[0], key_switch_scratch_bytes_max=ks_scratch,
                       key_switch_chunked_calls=ks_calls, key_switch_chunks=ks_chunks, **(mem_check or {}))
    # free the device before the PMC child processes, the appended workloads and the CPU baseline start
    if world > 1:
        dist.barrier()
    tail_note = None
    if args.workload != "bfv_c4":
        folded, plain, dropped = S.tail_stats()
        tail_note = ("deferred: relinearize leaves the mod-down by the special prime to the rescale that follows, which does both "
                     "rounding divisions with one transform per component (same words; %d folded / %d separate / %d discarded in "
                     "this process)" % (folded, plain, dropped)) if folded else "separate pass (%d tails completed on their own)" % plain
    product_note = None
    if args.workload == "headline":
        fused, formed, dropped = S.product_stats()
        product_note = ("deferred: multiply into the work object leaves the tensor product pending and the relinearize that follows forms it "
                        "inside its own kernels, in the same timed step (all of the product's arithmetic runs, none of it is cached or skipped; same words; "
                        "%d fused / %d formed on their own / %d discarded in this process)"
                        % (fused, formed, dropped)) if fused else "separate kernel"
    lanes_note = ""
    if w.lanes:
        lanes_note = ", %d evaluators x %d-item sub-batches on %d HIP streams" % (len(w.lanes), w.lanes[0]["cnt"], len(w.lanes))
    n_lanes = len(w.lanes) if w.lanes else 0
    metric, description, par = workloads.describe(args, world, w.dp)
    key_resident = w.keys.device_bytes() if hasattr(w.keys, "device_bytes") else None
    solo = rank == 0 and world == 1
    if solo and not EMU and (not args.no_pmc or not args.no_children):
        w.free()
        del w
        torch.cuda.empty_cache()
        S.release_pool()
        w = None
    if solo and roofline is not None and not args.no_pmc:
        roofline.update(counters.pmc_traffic(args, B, K, n))

    roofline_step = None
    if rank == 0 and result and not EMU and not args.step_child:
        # the whole step against the memory roofline in both accountings, and the kernels that actually bound it (vector ALU)
        sb = counters.step_bytes(args.workload, K, L, n)
        cts_per_s_per_gpu = result["value"] / world
        roofline_step = dict(
            bound="valu", note="the step is bound by vector-ALU issue in the key-switch (and BEHZ) kernels, not by HBM: the memory "
            "fractions below are what the algorithmic bytes amount to, the kernel table says how busy the vector ALU is",
            algorithmic_bytes_per_ciphertext=sb, peak=HBM_PEAK_GBS, unit="GB/s",
            achieved={k: round(v * cts_per_s_per_gpu / 1e9, 1) for k, v in sb.items()},
            frac={k: round(v * cts_per_s_per_gpu / 1e9 / HBM_PEAK_GBS, 4) for k, v in sb.items()})
        if world == 1 and not args.no_pmc:
            roofline_step.update(counters.step_counters(args, B))
            if roofline_step.get("hbm_counter_bytes_per_ciphertext"):
                # the step's REAL fabric traffic next to the algorithmic figure: what fraction of the HBM peak the step draws
                real = roofline_step["hbm_counter_bytes_per_ciphertext"]
                roofline_step["hbm_counter_over_key_amortised"] = round(real / sb["key_amortised"], 3)
                roofline_step["hbm_counter_gbs"] = round(real * cts_per_s_per_gpu / 1e9, 1)
                roofline_step["hbm_counter_frac_of_peak"] = round(real * cts_per_s_per_gpu / 1e9 / HBM_PEAK_GBS, 4)

    # ---- BASELINE configs[3] and configs[4] in the same line: short child runs of this script, one GPU
    appended = None
    if solo and args.workload == "headline" and not args.no_children and not args.ntt_only:
        appended = {name: child_workload(name, args) for name in ("bfv_c4", "rotate_c5")}

    # ---- SURVEY 8(d)'s other batch sizes (1 = latency; 64, 1024 = throughput) in the same line: one child process
    batch_sweep = None
    if solo and args.workload == "headline" and not args.no_sweep and not args.no_children and not args.ntt_only and not args.step_child:
        batch_sweep = child_sweep(args)

    cpu_line = None
    if solo and not args.no_cpu_baseline and not args.ntt_only and not EMU:
        scheme, _, bits, tbits, _ = workloads.WORKLOADS[args.workload]
        primes = S.CoeffModulus.Create(n, bits)
        cpu_line = cpu.cpu_baseline(args.workload, scheme, n, primes, S.PlainModulus.Batching(n, tbits) if scheme != "ckks" else 0, args)

    if rank == 0:
        line = dict(
            metric=metric, value=round(result.get("value", 0.0), 2), unit="ciphertexts/s", n_gpus=world, steps=args.steps,
            warmup=args.warmup, ms_per_step=round(result.get("ms_per_step", 0.0), 3), higher_is_better=True,
            scaling=scaling, vs_baseline=None, dtype="u64", data="synthetic",
            # rotate_c5 splits ONE key switch over the ranks: what it buys is time per ciphertext, so the line says that too
            **(dict(latency_ms_per_ciphertext=round(result.get("ms_per_step", 0.0) / max(1, B), 4)) if args.workload == "rotate_c5" and B else {}),
            verified_items=verified, rccl_ranks=r.collective_ranks, collective_backend=r.backend, per_rank=per_rank,
            config=dict(workload=description + (" [EMULATED KERNELS, CPU test]" if EMU else ""),
                        batch_per_gpu=B, key_switch_tail=tail_note, tensor_product=product_note, batch_sweep=batch_sweep,
                        cpus_per_rank=len(r.cpu_affinity) if r.cpu_affinity else None,
                        **(dict(shared_gpu="TEST MODE: the %d ranks share one device over gloo (SEALHIP_BENCH_SHARE_GPU) - not a measurement" % world)
                           if r.shared_gpu else {}), launch=("hipGraph replay" if args.graph else "eager") + lanes_note,
                        parallelism=par,
                        arithmetic="64-bit residues: exact double-precision (error-free FMA) arithmetic for primes below "
                                   "2^50, 64-bit Shoup/Barrett integer arithmetic for larger primes; results canonical u64",
                        key_bytes=2 * K * L * n * 8, key_bytes_resident=key_resident, device_memory=memory_note,
                        algorithmic_bytes_per_ciphertext=(2 * K * K + 18 * K - 2) * 8 * n,
                        **(dict(exchange_bytes_per_ciphertext=2 * (K + 1) * n * 8,
                                exchange_overlap=("%d sub-batches on %d streams sharing the communicator: the exchange of one runs while the "
                                                  "next is in its key-switch kernels" % (n_lanes, n_lanes)) if n_lanes else
                                "none: one batch, kernels and exchange in stream order (--streams S pipelines S sub-batches)")
                           if args.workload == "rotate_c5" else {})),
            roofline=roofline, roofline_step=roofline_step, roofline_configs1=ntt_c1, workloads=appended, cpu_baseline=cpu_line)
    # RCCL prints a version banner through C stdio when a communicator comes up; it sits in the C buffer until exit.  Tear
    # the process group down and flush the C streams first, so that the JSON line is the LAST thing on stdout.
    w = None
    if world > 1:
        dist.destroy_process_group()
    try:
        C.CDLL(None).fflush(None)
    except Exception:
        pass
    if rank == 0:
        print(json.dumps(line), flush=True)


def sweep_child(args):
    """`--sweep-child`: the headline step at the batch sizes of --sweep, one after the other in this process (a context, a key and
    a pair of resident input batches per size, freed before the next); batch 1 eager and as a hipGraph replay.  Prints one JSON
    list.  Throughput / latency only: the parity of these batch sizes is the GPU test suite's business."""
    import time
    args.no_cpu_baseline = args.no_pmc = args.no_verify = args.no_children = args.no_sweep = True
    args.workload, args.gpus = "headline", 1
    r = launcher.init_ranks(args)
    import seal_amd as S
    from seal_amd import shard
    out = []
    for b in [int(x) for x in args.sweep.split(",") if x.strip()]:
        for graph in ((False, True) if b == 1 else (False,)):
            args.batch, args.graph = b, graph
            steps = max(4, min(400, 4000 // b))
            try:
                w = workloads.build(args, S, shard, r.torch, r.group, r.device, r.dev_sync, 1, 0)
                elapsed = shard.timed_steps(w.step, steps, max(2, steps // 10), r.group, r.dev_sync, r.torch, r.coll_device)
                work = workloads.result_batch(w, args)
                assert work.size() == 2 and work.coeff_modulus_size() == w.K - 1 and work.batch() == b
                ms = 1e3 * elapsed / steps
                out.append(dict(batch=b, launch="hipGraph replay" if graph else "eager", steps=steps, ms_per_step=round(ms, 4),
                                ct_per_s=round(b / (ms * 1e-3), 1), latency_ms_per_ciphertext=round(ms, 4) if b == 1 else None))
                del work
                w.free()
                del w
            except BaseException as e:  # a size that does not fit or fails is reported, the others still run
                out.append(dict(batch=b, launch="hipGraph replay" if graph else "eager", error=repr(e)[:300]))
            r.torch.cuda.empty_cache()
            S.release_pool()
            time.sleep(0.05)
    try:
        C.CDLL(None).fflush(None)
    except Exception:
        pass
    print(json.dumps(out), flush=True)


def child_sweep(args):
    """the batch sweep as a child run of this script (the parent has released its HBM): [{batch, launch, ms_per_step, ct_per_s}]"""
    cmd = [sys.executable, os.path.abspath(__file__), "--sweep-child", "--sweep", args.sweep]
    try:
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ))
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith("[")]
        if p.returncode != 0 or not lines:
            return dict(error="sweep child failed (rc %d): %s" % (p.returncode, (p.stderr or p.stdout)[-400:]))
        return json.loads(lines[-1])
    except Exception as e:
        return dict(error="sweep child failed: %r" % (e,))


def child_workload(name, args):
    """one of the other BASELINE workloads as a child run of this script (its own process: the parent's 55 GB are released
    first); the child's line reduced to what identifies and qualifies the number.  Never takes the headline down."""
    cmd = [sys.executable, os.path.abspath(__file__), "--workload", name, "--steps", str(args.child_steps), "--warmup", "1",
           "--no-cpu-baseline", "--no-pmc", "--no-children"]
    if args.no_verify:
        cmd.append("--no-verify")
    try:
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ))
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        if p.returncode != 0 or not lines:
            return dict(error="child run failed (rc %d): %s" % (p.returncode, (p.stderr or p.stdout)[-400:]))
        d = json.loads(lines[-1])
        keep = ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "verified_items", "scaling", "dtype", "rccl_ranks")
        out = {k: d.get(k) for k in keep}
        out["config"] = {k: d["config"].get(k) for k in ("workload", "batch_per_gpu", "parallelism", "key_switch_tail")}
        out["roofline"] = d.get("roofline")
        rs = d.get("roofline_step") or {}
        out["roofline_step"] = {k: rs.get(k) for k in ("bound", "achieved", "frac", "unit", "peak")}
        return out
    except Exception as e:
        return dict(error="child run failed: %r" % (e,))


if __name__ == "__main__":
    main()
