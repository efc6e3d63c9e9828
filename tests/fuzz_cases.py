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
            # what a rare failure looks like matters more than that it happened: where the words are, what they hold, and whether
            # a second read of the same device memory sees them too (a wrong kernel result stays, a wrong transfer does not)
            wrong = got[b][tuple(bad.T)]
            again = DeviceSide.out(dev_ct)[b]
            raise AssertionError(
                "%s item %d: %d of %d words differ, first at %s, last at %s; %d of them are zero; first wrong words %s (expected %s); "
                "a second read of the device memory %s" % (
                    what, b, len(bad), exp.size, tuple(int(v) for v in bad[0]), tuple(int(v) for v in bad[-1]), int(np.sum(wrong == 0)),
                    [hex(int(v)) for v in wrong[:3]], [hex(int(v)) for v in exp[tuple(bad.T)][:3]],
                    "equals the first" if np.array_equal(again, got[b]) else ("is CORRECT" if np.array_equal(again, exp) else "differs from both")))
        assert dev_ct.is_ntt_form() == info["is_ntt_form"], what
        assert dev_ct.scale() == info["scale"], "%s: scale %r vs %r" % (what, dev_ct.scale(), info["scale"])
        assert dev_ct.correction_factor() == info["correction_factor"], what
        assert dev_ct.coeff_modulus_size() == info["coeff_modulus_size"] and dev_ct.size() == info["size"], what


def run_sequence(scheme, n, bits, tb, batch, nops, seed, check_prob=1.0, scale0=None, wild_prob=0.0, three_object_prob=0.0):
    """three_object_prob > 0 (CKKS): that share of the products is Evaluator::multiply(x, y, destination) instead of multiply_inplace -
    the form whose tensor product the library may leave pending (sealhip.h: SealHip_ProductStats); the operands stay alive for a
    while, or are dropped, or are written to, while the destination goes on through the sequence.
    check_prob < 1: the device result is compared with the reference's only after some of the operations (always after the
    last), so that state the library defers between calls - the key-switch tail, sealhip.h: SealHip_TailStats - survives into
    the next operation instead of being completed by the comparison's read"""
    rng = np.random.default_rng(seed)
    check_rng = np.random.default_rng(seed + 77)
    wild_rng = np.random.default_rng(seed + 78)
    form_rng = np.random.default_rng(seed + 79)
    keep = []   # operands of three-object products that are kept alive (a pending product reads them)
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
        elif check_prob < 1.0 and scheme == "bfv" and log and log[-1] in ("relinearize", "rotate", "conj") and "mod_switch" in ops and rng.random() < 0.7:
            op = "mod_switch"   # ... in BFV by a mod switch (the folded tail of round 4)
        if three_object_prob > 0 and scheme == "ckks":
            # sequences about pending products: a product is often followed directly by its relinearisation, and products are frequent
            if log and "(x,y->w" in log[-1] and x.size() == 3 and form_rng.random() < 0.75:
                op = "relinearize"
            elif x.size() == 2 and form_rng.random() < 0.3:
                op = "multiply"
        # keep CKKS scales inside the level's modulus: skip products that would overflow it
        if scheme == "ckks" and op in ("square", "multiply", "multiply32"):
            budget = sum(bits[:Kc]) - 2
            if 2 * np.log2(x.scale()) >= budget:
                op = "negate"
        if op in ("multiply32",) and x.size() + 1 > 4:
            op = "negate"
        if op == "rescale" and x.scale() < float(primes[Kc - 1]) * 2.0:
            op = "mod_switch"   # rescaling would push the scale below 1
        # wild_prob > 0: now and then an operation chosen with none of the guards above (a rotation of a three-part ciphertext,
        # a rescale at the end of the chain or of a BFV ciphertext, a product whose scale does not fit, operands at different
        # levels or scales): whatever the device says about it - accept or reject - the reference must say the same
        wild = wild_prob > 0 and wild_rng.random() < wild_prob
        mismatch = None
        if wild:
            op = ["rotate", "conj", "square", "multiply", "relinearize", "mod_switch", "rescale", "add", "sub", "multiply32"][wild_rng.integers(0, 10)]
            if op in ("add", "sub", "multiply") and wild_rng.random() < 0.5:
                mismatch = ["level", "scale"][wild_rng.integers(0, 2)]
        log.append(op + ("!" if wild else "") + ("~" + mismatch if mismatch else ""))
        # the operation as two closures - device side, reference side - over operands prepared beforehand, so that a call the
        # device rejects can be replayed on the reference (which must reject it with the same exception class)
        def second_operand(size):
            y, ry = fresh(size=size)
            while y.coeff_modulus_size() > Kc:     # bring the fresh operand to x's level / form / scale
                d.ev.mod_switch_to_next_inplace(y)
                for r in ry:
                    o.ref.mod_switch_to_next_inplace(r)
            if mismatch == "level" and Kc >= 2:
                d.ev.mod_switch_to_next_inplace(y)
                for r in ry:
                    o.ref.mod_switch_to_next_inplace(r)
            if scheme == "ckks":
                ysc = x.scale() * (2.0 if mismatch == "scale" else 1.0)
                y.set_scale(ysc)
                ry = [o.ref.ct(r.info()["chain_index"], r.data(), True, ysc, 1) for r in ry]
            return y, ry

        def each(name, *args):
            def run():
                for r in rx:
                    getattr(o.ref, name)(r, *args)
            return run

        if op in ("add", "sub"):
            y, ry = second_operand(int(rng.integers(2, 4)))
            dev_call = lambda: getattr(d.ev, op + "_inplace")(x, y)
            ref_call = lambda: [getattr(o.ref, op + "_inplace")(r, q) for r, q in zip(rx, ry)]
        elif op == "negate":
            dev_call, ref_call = (lambda: d.ev.negate_inplace(x)), each("negate_inplace")
        elif op in ("multiply", "multiply32"):
            y, ry = second_operand(2)
            if scheme == "ckks" and x.size() == 2 and form_rng.random() < three_object_prob:
                fate = int(form_rng.integers(0, 4))   # of the operands: 0 / 1 kept alive, 2 dropped at once, 3 written to while pending

                def dev_call():
                    nonlocal x
                    w = S.Ciphertext(d.ctx, batch=batch)
                    d.ev.multiply(x, y, w)
                    if fate <= 1:
                        keep.append((x, y))
                        if len(keep) > 3:
                            keep.pop(0)
                    elif fate == 3:
                        d.ev.negate_inplace(y)
                        d.ev.add_inplace(x, y)
                        keep.append((x, y))
                    x = w
                log[-1] += "(x,y->w:%d)" % fate
            else:
                dev_call = lambda: d.ev.multiply_inplace(x, y)
            ref_call = lambda: [o.ref.multiply_inplace(r, q) for r, q in zip(rx, ry)]
        elif op == "square":
            dev_call, ref_call = (lambda: d.ev.square_inplace(x)), each("square_inplace")
        elif op == "relinearize":
            dev_call, ref_call = (lambda: d.ev.relinearize_inplace(x, d.rlk)), each("relinearize_inplace")
        elif op == "rotate":
            s = steps[rng.integers(0, len(steps))]
            if scheme == "ckks":
                dev_call, ref_call = (lambda: d.ev.rotate_vector_inplace(x, s, d.glk)), each("rotate_vector_inplace", s)
            else:
                dev_call, ref_call = (lambda: d.ev.rotate_rows_inplace(x, s, d.glk)), each("rotate_rows_inplace", s)
        elif op == "conj":
            if scheme == "ckks":
                dev_call, ref_call = (lambda: d.ev.complex_conjugate_inplace(x, d.glk)), each("complex_conjugate_inplace")
            else:
                dev_call, ref_call = (lambda: d.ev.rotate_columns_inplace(x, d.glk)), each("rotate_columns_inplace")
        elif op == "mod_switch":
            dev_call, ref_call = (lambda: d.ev.mod_switch_to_next_inplace(x)), each("mod_switch_to_next_inplace")
        elif op == "rescale":
            dev_call, ref_call = (lambda: d.ev.rescale_to_next_inplace(x)), each("rescale_to_next_inplace")
        else:
            raise AssertionError(op)

        try:
            dev_call()
        except (S.InvalidArgument, S.LogicError) as dev_exc:
            # replay on the reference: it must reject the same call, with the same exception class (RefError codes: 1
            # invalid_argument, 2 logic_error); the sequence ends here - the operand may be half-way for in-place forms
            want = 1 if isinstance(dev_exc, S.InvalidArgument) else 2
            try:
                ref_call()
            except sealref.RefError as ref_exc:
                assert ref_exc.code == want, "device raised %s (%s), the reference %s, after %s" % (
                    type(dev_exc).__name__, dev_exc.message, ref_exc, " > ".join(log))
                return log + ["both raised %s: %s" % (type(dev_exc).__name__, dev_exc.message)]
            raise AssertionError("the device raised %s (%s) but the reference accepted the call after %s" % (
                type(dev_exc).__name__, dev_exc.message, " > ".join(log)))
        try:
            ref_call()
        except sealref.RefError as ref_exc:
            raise AssertionError("the reference raised %s but the device accepted the call after %s" % (ref_exc, " > ".join(log)))
        if op_index == nops - 1 or check_rng.random() < check_prob:
            _same(x, rx, "%s n=%d bits=%s seed=%d after %s" % (scheme, n, bits, seed, " > ".join(log)))
    return log
