"""bench.py, workload side: the three timed workloads (BASELINE headline, configs[3], configs[4]) - parameters, synthetic keys and
ciphertext batches resident in HBM, the step that is timed - and the reference check of sampled output items."""
import os

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


def device_uniform(torch, primes, shape_prefix, n, device):
    """uniform residues per RNS component, generated on the device: [*prefix][len(primes)][n] int64"""
    comps = [torch.randint(0, int(q), tuple(shape_prefix) + (1, n), dtype=torch.int64, device=device) for q in primes]
    return torch.cat(comps, dim=len(shape_prefix)).contiguous()


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
    cpus = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)   # this rank's share of the host
    with ThreadPoolExecutor(max_workers=max(1, min(len(items), cpus))) as pool:
        results = list(pool.map(expected, items))
    for (b, got), (exp, ref_scale) in zip(inputs_ready, results):
        if got.shape != exp.shape or not np.array_equal(got, exp):
            raise SystemExit("bench.py: item %d of the timed batch differs from the reference Evaluator" % b)
        if scheme == "ckks" and work.scale() != ref_scale:
            raise SystemExit("bench.py: scale metadata differs from the reference (%r vs %r)" % (work.scale(), ref_scale))
    return len(items)


class Workload:
    """everything the timed region needs, resident before it starts: context, evaluator, key, input batches, step()"""

    def free(self):
        """drop every device object (before child processes need the HBM)"""
        self.__dict__.clear()


def estimate_device_bytes(args, world, rank):
    """HBM this rank will hold for the workload, to within ~20 %: the torch input tensors, their ciphertext copies, the result, the
    key-switch sums and its chunked intermediate (at most the cap), the largest transform scratch, keys and tables."""
    scheme, n, bits, tbits, default_batch = WORKLOADS[args.workload]
    L, K = len(bits), len(bits) - 1
    if args.workload == "bfv_c4":
        total = args.total_batch if not EMU else 4
        B = total // world + (1 if rank < total % world else 0)
    else:
        B = args.batch or default_batch
    poly = K * n * 8                                   # one polynomial of one item at the first level
    operands = (1 if args.workload == "rotate_c5" else 2) * 2 * B * poly
    cap = float(os.environ.get("SEALHIP_KS_SCRATCH_CAP_MIB", "16384")) * 2 ** 20
    ks_mid = min(B * (K + 1) * K * n * 8, cap)
    behz = 6 * B * (K + 3) * n * 8 * 3 if scheme == "bfv" else 0   # lifted operands in q + Bsk + m~ and the tensor product there
    return int(2 * operands + 3 * B * poly + 2 * B * (K + 1) * n * 8 + B * poly + ks_mid + 3 * B * poly + behz + 2 * K * L * n * 8 * 2 + 2 ** 30)


def build(args, S, shard, torch, group, device, dev_sync, world, rank, shared_gpu=False):
    w = Workload()
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
        # (ranks sharing one device - the one-GPU test mode - cannot use RCCL: the partial sums go through torch.distributed)
        dp = shard.DigitParallel(ev, torch, group, device, exchange=args.exchange, native=False if shared_gpu else (True if args.native_comm else None))
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
    if B == 0:
        # more ranks than items of a sharded total batch (bfv_c4 --total-batch 4 on 8 ranks): this rank owns nothing, times an empty
        # step and still takes part in every barrier and helper collective
        w.__dict__.update(scheme=scheme, n=n, primes=primes, L=L, K=K, t_plain=t_plain, ctx=ctx, ev=ev, B=0, scaling=scaling, keys=keys,
                          dp=dp, xs=None, ys=None, x=None, y=None, work=None, lanes=None, step=lambda: None, holder={}, key_host=None,
                          scale=scale, want_verify=False)
        return w
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
    holder = {}
    if lanes and args.workload == "rotate_c5":
        rot_scale = float(primes[K - 1]) * 2.0 ** 10

        def step():
            for ln in lanes:
                wk = ln["ev"].copy_to(ln["x"], ln["work"])   # the rotation works in place: stage the resident input on the lane's stream
                wk.set_scale(rot_scale)
                ln["dp"].rotate_vector_inplace(wk, 1, keys)
                ln["ev"].rescale_to_next_inplace(wk)
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
    elif world == 1 and (dp is None or dp.comm is None):
        x.set_scale(float(primes[K - 1]) * 2.0 ** 10)
        holder["work"] = work

        def step():
            ev.rotate_vector(x, 1, keys, work)   # one rank: nothing to split - the out-of-place rotation reads the resident batch
            ev.rescale_to_next_inplace(work)
    else:
        rot_scale = float(primes[K - 1]) * 2.0 ** 10

        def step():
            wk = x.copy()                    # device-to-device copy of the resident batch (the digit-parallel rotation works in place)
            wk.set_scale(rot_scale)
            dp.rotate_vector_inplace(wk, 1, keys)
            ev.rescale_to_next_inplace(wk)
            holder["work"] = wk

    if args.graph and lanes:
        raise SystemExit("bench.py: --graph captures one evaluator's stream; not combined with --streams")
    if args.graph and not args.ntt_only and args.workload != "rotate_c5":
        step()  # eager once: lazily built tables, pool warm-up
        dev_sync()
        w.graph = ev.capture(step)
        step = w.graph.launch

    w.__dict__.update(scheme=scheme, n=n, primes=primes, L=L, K=K, t_plain=t_plain, ctx=ctx, ev=ev, B=B, scaling=scaling, keys=keys,
                      dp=dp, xs=xs, ys=ys, x=x, y=y, work=work, lanes=lanes, step=step, holder=holder, key_host=key_host,
                      scale=scale, want_verify=want_verify)
    return w


def result_batch(w, args):
    """the batch the timed steps left behind (the rotation workload makes a new one every step)"""
    if w.lanes:
        return LaneView(w.lanes)
    if args.workload == "rotate_c5":
        return w.holder["work"]
    return w.work


def describe(args, world, dp):
    """(metric, workload description, parallelism) of the JSON line"""
    names = {
        "headline": ("CKKS multiply+relinearize+rescale ciphertexts/sec @ N=2^16, L=16",
                     "CKKS N=65536, coeff_modulus {60,14x50,60} (L=16, K=15): multiply (x, y -> work, the operands stay resident) + relinearize_inplace + "
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
    return names[0], names[1], par
