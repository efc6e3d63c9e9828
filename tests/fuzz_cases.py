"""Randomised parity against the REAL reference (oracle/_ref): random parameter sets (degree, chain length, prime sizes
on both sides of the 2^50 back-end split), random batches and random sequences of Evaluator operations; after every
operation the device words, the scale, the correction factor and the level must equal the reference's.  Used by the
CPU suite (small sizes, fiber emulator) and the GPU suite (all sizes)."""
import numpy as np

import seal_amd as S
import sealref
from harness import DeviceSide
from oracle import Oracle, coeff_modulus_create, plain_modulus_batching, rand_ct


def _same(dev_ct, ref_cts, what):
    got = DeviceSide.out(dev_ct)
    for b, r in enumerate(ref_cts):
        exp, info = r.data(), r.info()
        assert got[b].shape == exp.shape, "%s: shape %s vs %s" % (what, got[b].shape, exp.shape)
        if not np.array_equal(got[b], exp):
            bad = np.argwhere(got[b] != exp)
            raise AssertionError("%s item %d: %d of %d words differ, first at %s" % (what, b, len(bad), exp.size, tuple(bad[0])))
        assert dev_ct.is_ntt_form() == info["is_ntt_form"], what
        assert dev_ct.scale() == info["scale"], "%s: scale %r vs %r" % (what, dev_ct.scale(), info["scale"])
        assert dev_ct.correction_factor() == info["correction_factor"], what
        assert dev_ct.coeff_modulus_size() == info["coeff_modulus_size"] and dev_ct.size() == info["size"], what


def run_sequence(scheme, n, bits, tb, batch, nops, seed, check_prob=1.0, scale0=None):
    """check_prob < 1: the device result is compared with the reference's only after some of the operations (always after the
    last), so that state the library defers between calls - the key-switch tail, sealhip.h: SealHip_TailStats - survives into
    the next operation instead of being completed by the comparison's read"""
    rng = np.random.default_rng(seed)
    check_rng = np.random.default_rng(seed + 77)
    primes = coeff_modulus_create(n, bits)
    t = plain_modulus_batching(n, tb) if scheme != "ckks" else 0
    L, K = len(primes), len(primes) - 1
    probe = Oracle(scheme, n, primes, t)
    steps = [1, -1] if n >= 8 else [1]
    elts = sorted({probe.galois_elt_from_step(s) for s in steps} | {2 * n - 1})
    o = Oracle(scheme, n, primes, t, galois_elts=elts)
    d = DeviceSide(scheme, n, primes, t)
    d.upload_keys(o)
    ntt = scheme != "bfv"
    if scale0 is None:
        scale0 = 2.0 ** 8 if scheme == "ckks" else 1.0
    ci = o._ci(K)

    def fresh(size=2):
        slabs = [rand_ct(rng, primes, K, n, size=size) for _ in range(batch)]
        c = d.ct(slabs, scale=scale0, is_ntt=ntt)
        return c, [o.ref.ct(ci, s, ntt, scale0, 1) for s in slabs]

    x, rx = fresh()
    log = []
    for op_index in range(nops):
        Kc = x.coeff_modulus_size()
        ops = ["add", "sub", "negate"]
        if x.size() == 2:
            ops += ["rotate", "conj", "square", "multiply"]
            if scheme != "ckks" or x.scale() * x.scale() < 2.0 ** (min(bits) * Kc - 4):
                pass
        if x.size() == 3:
            ops += ["relinearize", "relinearize", "multiply32"]
        if Kc >= 2:
            ops += ["mod_switch"]
            if scheme == "ckks":
                ops += ["rescale"]
        op = ops[rng.integers(0, len(ops))]
        if check_prob < 1.0 and log and log[-1] in ("relinearize", "rotate", "conj") and "rescale" in ops and rng.random() < 0.7:
            op = "rescale"   # deferred-state runs: a key switch is often followed directly by a rescale
        # keep CKKS scales inside the level's modulus: skip products that would overflow it
        if scheme == "ckks" and op in ("square", "multiply", "multiply32"):
            budget = sum(bits[:Kc]) - 2
            if 2 * np.log2(x.scale()) >= budget:
                op = "negate"
        if op in ("multiply32",) and x.size() + 1 > 4:
            op = "negate"
        if op == "rescale" and x.scale() < float(primes[Kc - 1]) * 2.0:
            op = "mod_switch"   # rescaling would push the scale below 1
        log.append(op)
        state = {"rx": rx}

        def apply_both():
            rx = state["rx"]
            if op in ("add", "sub"):
                y, ry = fresh(size=int(rng.integers(2, 4)))
                # bring the fresh operand to x's level / form / scale
                while y.coeff_modulus_size() > Kc:
                    d.ev.mod_switch_to_next_inplace(y)
                    for r in ry:
                        o.ref.mod_switch_to_next_inplace(r)
                if scheme == "ckks":
                    y.set_scale(x.scale())
                    ry = [o.ref.ct(o._ci(Kc), r.data(), True, x.scale(), 1) for r in ry]
                getattr(d.ev, op + "_inplace")(x, y)
                for r, q in zip(rx, ry):
                    getattr(o.ref, op + "_inplace")(r, q)
            elif op == "negate":
                d.ev.negate_inplace(x)
                for r in rx:
                    o.ref.negate_inplace(r)
            elif op in ("multiply", "multiply32"):
                y, ry = fresh(size=2)
                while y.coeff_modulus_size() > Kc:
                    d.ev.mod_switch_to_next_inplace(y)
                    for r in ry:
                        o.ref.mod_switch_to_next_inplace(r)
                if scheme == "ckks":
                    y.set_scale(x.scale())
                    ry = [o.ref.ct(o._ci(Kc), r.data(), True, x.scale(), 1) for r in ry]
                d.ev.multiply_inplace(x, y)
                for r, q in zip(rx, ry):
                    o.ref.multiply_inplace(r, q)
            elif op == "square":
                d.ev.square_inplace(x)
                for r in rx:
                    o.ref.square_inplace(r)
            elif op == "relinearize":
                d.ev.relinearize_inplace(x, d.rlk)
                for r in rx:
                    o.ref.relinearize_inplace(r)
            elif op == "rotate":
                s = steps[rng.integers(0, len(steps))]
                if scheme == "ckks":
                    d.ev.rotate_vector_inplace(x, s, d.glk)
                    for r in rx:
                        o.ref.rotate_vector_inplace(r, s)
                else:
                    d.ev.rotate_rows_inplace(x, s, d.glk)
                    for r in rx:
                        o.ref.rotate_rows_inplace(r, s)
            elif op == "conj":
                if scheme == "ckks":
                    d.ev.complex_conjugate_inplace(x, d.glk)
                    for r in rx:
                        o.ref.complex_conjugate_inplace(r)
                else:
                    d.ev.rotate_columns_inplace(x, d.glk)
                    for r in rx:
                        o.ref.rotate_columns_inplace(r)
            elif op == "mod_switch":
                d.ev.mod_switch_to_next_inplace(x)
                for r in rx:
                    o.ref.mod_switch_to_next_inplace(r)
            elif op == "rescale":
                d.ev.rescale_to_next_inplace(x)
                for r in rx:
                    o.ref.rescale_to_next_inplace(r)

        try:
            apply_both()
        except (S.InvalidArgument, S.LogicError) as dev_exc:
            # a call the device rejects must be rejected by the reference with the same exception class (run on fresh
            # copies of the last agreed state is not possible for in-place ops, so the sequence simply ends here)
            return log + ["device raised %s: %s" % (type(dev_exc).__name__, dev_exc.message)]
        except sealref.RefError as ref_exc:
            raise AssertionError("the reference raised %s but the device accepted the call after %s" % (ref_exc, " > ".join(log)))
        rx = state["rx"]
        if op_index == nops - 1 or check_rng.random() < check_prob:
            _same(x, rx, "%s n=%d bits=%s seed=%d after %s" % (scheme, n, bits, seed, " > ".join(log)))
    return log
