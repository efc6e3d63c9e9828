"""Time-boxed soak of the shortest device sequences there are: fresh upload -> ONE evaluator operation -> download, compared
word for word with the real reference (oracle/_ref), thousands of times per minute.

Why: round 3 logged one wrong device result in about 1 800 random operation sequences (profiles/r03_fuzz_stress.txt: the square
of a fresh three-item CKKS batch at N = 4096, 256 consecutive words of the middle polynomial) and could not reproduce it.  The
operations below are the ones on that path and next to it:
  square_new       tensor_2x2's new-slab branch (evaluator.cpp here; reference evaluator.cpp:1022-1142 ckks_square)
  square_room      its in-place branch: the slab already has room for three polynomials
  multiply_new     two fresh operands, product into the first (reference evaluator.cpp:604-663)
  add_grow         size 2 + size 3: Ciphertext::resize's copy path - pool block + hipMemcpyAsync + free of the source + hipMemsetAsync
                   (reference ciphertext.cpp:101-116) - then the add kernel and a device-to-device copy of the third polynomial
  resize_grow      the C ABI's Ciphertext_Resize1 2 -> 3 on its own: copy + zero fill, nothing else
Every phase runs them with direct host copies (hipMemcpy on the caller's pageable buffer) or with staged ones (pinned bounce
buffers, SealHip_SetStagedHostCopies), on the NULL stream or on a non-blocking stream - a wrong kernel / ordering shows in every
copy mode, a wrong pageable-memory transfer only in one.  Every `churn_every` iterations the next operation runs on a context,
evaluator and key that were created just before it and replace the old ones (the failing operation of round 3 was the first after exactly that).  Expected words come from the reference once per input set; the device
side repeats.  On a mismatch the report says where the wrong words are, what they hold (the uploaded source word? zero? what
the reference expects for ANOTHER input set, i.e. stale data?), whether a second download shows them too, and whether the same
operation from a fresh upload is right the next time.  TEST INFRASTRUCTURE (imports the oracle)."""
import os
import time

import numpy as np

import seal_amd as S
from harness import DeviceSide
from oracle import Oracle, coeff_modulus_create, rand_ct

OPS = ("square_new", "square_room", "multiply_new", "add_grow", "resize_grow")


class _Case:
    """one context, a few input sets, the reference's result for every (operation, input set)"""

    def __init__(self, n, bits, batch, sets, seed):
        rng = np.random.default_rng(seed)
        self.n, self.bits, self.batch = n, list(bits), batch
        primes = coeff_modulus_create(n, bits)
        self.primes = primes
        self.K = K = len(primes) - 1
        self.o = Oracle("ckks", n, primes, 0)
        self.d = DeviceSide("ckks", n, primes, 0)
        self.rebuilds = 0
        ci = self.o._ci(K)
        self.scale = 2.0 ** 8
        self.sets = []
        for _ in range(sets):
            x = [rand_ct(rng, primes, K, n, size=2) for _ in range(batch)]
            y = [rand_ct(rng, primes, K, n, size=2) for _ in range(batch)]
            z = [rand_ct(rng, primes, K, n, size=3) for _ in range(batch)]
            ref = lambda s: self.o.ref.ct(ci, s, True, self.scale, 1)
            exp = {}
            exp["square_new"] = [self.o.ref.square_inplace(ref(s)).data() for s in x]
            exp["square_room"] = exp["square_new"]
            exp["multiply_new"] = [self.o.ref.multiply_inplace(ref(a), ref(b)).data() for a, b in zip(x, y)]
            exp["add_grow"] = [self.o.ref.add_inplace(ref(a), ref(c)).data() for a, c in zip(x, z)]
            exp["resize_grow"] = [np.concatenate([a, np.zeros((1, K, n), dtype=np.uint64)]) for a in x]
            self.sets.append({"x": x, "y": y, "z": z, "exp": {k: np.stack(v, axis=1) for k, v in exp.items()}})

    def rebuild_device(self, stream):
        """what the start of a fuzz sequence does (tests/fuzz_cases.py - the one wrong result of round 3 was the FIRST operation after it): a
        new SEALContext and Evaluator (hipMalloc / hipMemcpy of the tables), the relinearization key uploaded again (hipMalloc + host copy),
        the old context, evaluator and key destroyed (hipFree)"""
        self.d = DeviceSide("ckks", self.n, self.primes, 0)
        self.d.upload_keys(self.o)
        self.d.ev.set_stream(stream)
        self.rebuilds += 1

    def label(self):
        return "ckks n=%d bits=%s batch=%d" % (self.n, self.bits, self.batch)

    def device_run(self, op, s):
        """fresh objects every time: the pool hands their blocks round"""
        d, st = self.d, self.sets[s]
        if op == "square_new":
            c = d.ct(st["x"], scale=self.scale)
            d.ev.square_inplace(c)
        elif op == "square_room":
            pid = d.parms_id_for_K(self.K)
            c = S.Ciphertext(d.ctx, batch=self.batch)
            c.resize(pid, 3)
            c.resize(pid, 2)
            c.set_is_ntt_form(True)
            c.set_scale(self.scale)
            c.load(np.stack(st["x"], axis=1))
            d.ev.square_inplace(c)
        elif op == "multiply_new":
            c = d.ct(st["x"], scale=self.scale)
            y = d.ct(st["y"], scale=self.scale)
            d.ev.multiply_inplace(c, y)
        elif op == "add_grow":
            c = d.ct(st["x"], scale=self.scale)
            z = d.ct(st["z"], scale=self.scale)
            d.ev.add_inplace(c, z)
        elif op == "resize_grow":
            c = d.ct(st["x"], scale=self.scale)
            c.resize(d.parms_id_for_K(self.K), 3)
        else:
            raise ValueError(op)
        return c


def _describe(case, op, s, got, c):
    st = case.sets[s]
    exp = st["exp"][op]
    bad = np.argwhere(got != exp)
    flat = np.flatnonzero(got.reshape(-1) != exp.reshape(-1))
    runs = 1 + int(np.sum(np.diff(flat) != 1))
    wrong = got[tuple(bad.T)]
    src = np.stack(st["x"], axis=1)
    src3 = np.concatenate([src, np.zeros_like(src[:1])]) if src.shape[0] < got.shape[0] else src
    eq_src = int(np.sum(wrong == src3[tuple(bad.T)])) if src3.shape == got.shape else -1
    stale = {}
    for s2, other in enumerate(case.sets):
        for op2 in OPS:
            e2 = other["exp"][op2]
            if (s2, op2) != (s, op) and e2.shape == got.shape:
                hits = int(np.sum(wrong == e2[tuple(bad.T)]))
                if hits:
                    stale["set %d %s" % (s2, op2)] = hits
    again = c.to_numpy()
    second = "equals the first (the device memory holds the wrong words)" if np.array_equal(again, got) else (
        "is CORRECT (the first transfer was wrong)" if np.array_equal(again, exp) else "differs from both")
    retry = np.array_equal(case.device_run(op, s).to_numpy(), exp)
    return ("%s, %s, input set %d: %d of %d words differ in %d run(s); first at [poly, item, comp, word] = %s (flat word %d, byte %d of the "
            "slab), last at %s; %d of the wrong words are zero, %d equal the uploaded source word at that position, stale matches %s; "
            "first wrong words %s, expected %s; a second download %s; the same operation from a fresh upload is %s") % (
        case.label(), op, s, len(bad), exp.size, runs, [int(v) for v in bad[0]], int(flat[0]), int(flat[0]) * 8,
        [int(v) for v in bad[-1]], int(np.sum(wrong == 0)), eq_src, stale or "none",
        [hex(int(v)) for v in wrong[:3]], [hex(int(v)) for v in exp[tuple(bad.T)][:3]], second, "right" if retry else "WRONG again"), again


def default_cases(small=False):
    if small:   # emulated build: index arithmetic and host logic only
        return [_Case(256, [40, 30, 41], 3, 2, 31), _Case(128, [36, 37], 1, 2, 32)]
    return [
        _Case(4096, [54, 42, 55], 3, 3, 41),     # the parameters of the one wrong result
        _Case(2048, [36, 50], 2, 3, 42),          # K = 1
        _Case(8192, [50, 40, 60], 1, 3, 43),      # two-pass sizes, both arithmetic classes
        _Case(4096, [36, 36, 37], 2, 3, 44),
        _Case(8192, [58, 58, 59, 60], 3, 2, 45),
    ]


def run_soak(seconds, cases=None, phases=None, seed=1, dump_dir=None, max_iterations=None, churn_every=97):
    """-> dict of counters; raises AssertionError at the first mismatch (after writing its words to dump_dir)"""
    cases = cases or default_cases()
    if phases is None:
        phases = [("direct copies, NULL stream", False, False), ("staged copies, NULL stream", True, False),
                  ("direct copies, non-blocking stream", False, True)]
    rng = np.random.default_rng(seed)
    stats = {"iterations": 0, "per_phase": {}, "per_op": {op: 0 for op in OPS}, "seconds": 0.0}
    t_begin = time.time()
    stream = None
    try:
        for name, staged, nonblocking in phases:
            S.set_staged_host_copies(staged)
            if nonblocking:
                stream = S.Stream(non_blocking=True)
            for case in cases:
                case.d.ev.set_stream(stream.handle if nonblocking else None)
            t_end = time.time() + seconds / len(phases)
            count = 0
            while time.time() < t_end and (max_iterations is None or count < max_iterations):
                case = cases[int(rng.integers(0, len(cases)))]
                if churn_every and count % churn_every == churn_every - 1:
                    case.rebuild_device(stream.handle if nonblocking else None)
                op = OPS[int(rng.integers(0, len(OPS)))]
                s = int(rng.integers(0, len(case.sets)))
                c = case.device_run(op, s)
                got = c.to_numpy()
                if not np.array_equal(got, case.sets[s]["exp"][op]):
                    text, again = _describe(case, op, s, got, c)
                    text = "soak phase '%s', iteration %d: %s" % (name, stats["iterations"] + count, text)
                    if dump_dir:
                        os.makedirs(dump_dir, exist_ok=True)
                        np.savez_compressed(os.path.join(dump_dir, "soak_mismatch_%d.npz" % os.getpid()), got=got, again=again,
                                            expected=case.sets[s]["exp"][op], x=np.stack(case.sets[s]["x"], axis=1))
                        with open(os.path.join(dump_dir, "soak_mismatch_%d.txt" % os.getpid()), "w") as f:
                            f.write(text + "\n")
                    raise AssertionError(text)
                count += 1
                stats["per_op"][op] += 1
            stats["per_phase"][name] = count
            stats["iterations"] += count
    finally:
        S.set_staged_host_copies(False)
        for case in cases:
            case.d.ev.set_stream(None)
    stats["seconds"] = time.time() - t_begin
    stats["context_rebuilds"] = sum(c.rebuilds for c in cases)
    return stats
