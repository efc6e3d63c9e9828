"""Parity cases shared by the CPU (fiber-emulated, index arithmetic only) and GPU (the real thing) test
files.  Every comparison is bit-exact (np.array_equal on uint64 slabs): the path is integer work.
Each case mirrors a reference test or bench case (cited)."""
import os

import numpy as np

import seal_amd as S
from harness import DeviceSide
from oracle import Oracle, coeff_modulus_create, plain_modulus_batching, rand_ct


def _eq(got, exp, what):
    assert got.shape == exp.shape, "%s: shape %s vs %s" % (what, got.shape, exp.shape)
    if not np.array_equal(got, exp):
        bad = np.argwhere(got != exp)
        raise AssertionError("%s: %d of %d words differ, first at %s: got %d expected %d" % (
            what, len(bad), got.size, tuple(bad[0]), got[tuple(bad[0])], exp[tuple(bad[0])]))


# ---- NTT: native/tests/seal/util/ntt.cpp:103-133 (round trip) + ciphertext-level parity (SURVEY §4 ii)
def case_ntt(n, bits, polys=2, seed=1):
    primes = coeff_modulus_create(n, bits)
    o = Oracle("ckks", n, primes)
    d = DeviceSide("ckks", n, primes)
    L = len(primes)
    for i in range(L):
        assert d.ctx.ntt_root(i) == o.ntt_root(i), "minimal primitive root of prime %d" % i
    rng = np.random.default_rng(seed)
    x = np.stack([np.stack([rng.integers(0, primes[i], n, dtype=np.uint64) for i in range(L)]) for _ in range(polys)])
    buf = S.DeviceBuffer.from_numpy(x)
    S.ntt_forward(d.ctx, buf, polys, L)
    fwd = buf.to_numpy(x.shape)
    for p in range(polys):
        _eq(fwd[p], o.ntt(0, x[p], "fwd"), "ntt_negacyclic_harvey poly %d" % p)
    S.ntt_inverse(d.ctx, buf, polys, L)
    _eq(buf.to_numpy(x.shape), x, "inverse(forward(x)) == x")
    buf = S.DeviceBuffer.from_numpy(x)
    S.ntt_inverse(d.ctx, buf, polys, L)
    inv = buf.to_numpy(x.shape)
    for p in range(polys):
        _eq(inv[p], o.ntt(0, x[p], "inv"), "inverse_ntt_negacyclic_harvey poly %d" % p)
    # lazy variants: same residues, inside the reference's lazy ranges (ntt.h:30-61)
    q = np.array(primes, dtype=np.uint64)[None, :, None]
    buf = S.DeviceBuffer.from_numpy(x)
    S.ntt_forward(d.ctx, buf, polys, L, lazy=True)
    lz = buf.to_numpy(x.shape)
    assert (lz < 4 * q).all()
    _eq(lz % q, fwd, "lazy forward mod q")
    buf = S.DeviceBuffer.from_numpy(x)
    S.ntt_inverse(d.ctx, buf, polys, L, lazy=True)
    lz = buf.to_numpy(x.shape)
    assert (lz < 2 * q).all()
    _eq(lz % q, inv, "lazy inverse mod q")


# ---- dyadic product: native/tests/seal/util/polyarithsmallmod.cpp:545-641
def case_dyadic(n, bits, seed=2):
    primes = coeff_modulus_create(n, bits)
    o = Oracle("ckks", n, primes)
    d = DeviceSide("ckks", n, primes)
    L = len(primes)
    rng = np.random.default_rng(seed)
    a = np.stack([rng.integers(0, primes[i], n, dtype=np.uint64) for i in range(L)])
    b = np.stack([rng.integers(0, primes[i], n, dtype=np.uint64) for i in range(L)])
    da, db, dr = S.DeviceBuffer.from_numpy(a), S.DeviceBuffer.from_numpy(b), S.DeviceBuffer(a.size)
    S.dyadic_product(d.ctx, da, db, dr, 1, L)
    got = dr.to_numpy(a.shape)
    for i in range(L):
        _eq(got[i], o.dyadic(i, a[i], b[i]), "dyadic_product_coeffmod prime %d" % i)


# ---- CKKS: CKKSEncryptMultiplyRelinRescaleDecrypt / CKKSEncryptRotateDecrypt
#      (native/tests/seal/evaluator.cpp:3513, 4326) at ciphertext level; bench cases native/bench/ckks.cpp
def case_ckks_pipeline(n, bits, batch=2, steps=(1,), seed=3, check_transforms=True):
    primes = coeff_modulus_create(n, bits)
    L = len(primes)
    K = L - 1
    probe = Oracle("ckks", n, primes)
    elts = [probe.galois_elt_from_step(s) for s in steps]
    o = Oracle("ckks", n, primes, galois_elts=elts)
    d = DeviceSide("ckks", n, primes)
    d.upload_keys(o)
    for s, e in zip(steps, elts):
        assert d.ctx.galois_elt_from_step(s) == e
    rng = np.random.default_rng(seed)
    xs = [rand_ct(rng, primes, K, n) for _ in range(batch)]
    ys = [rand_ct(rng, primes, K, n) for _ in range(batch)]
    cx, cy = d.ct(xs, scale=2.0 ** 10), d.ct(ys, scale=2.0 ** 10)

    d.ev.multiply_inplace(cx, cy)  # ckks_multiply, evaluator.cpp:569
    assert cx.size() == 3 and cx.is_ntt_form() and cx.scale() == 2.0 ** 20
    cur = d.out(cx)
    for b in range(batch):
        _eq(cur[b], o.multiply(xs[b], ys[b]), "multiply item %d" % b)

    d.ev.relinearize_inplace(cx, d.rlk)  # evaluator.cpp:1144
    assert cx.size() == 2
    nxt = d.out(cx)
    for b in range(batch):
        _eq(nxt[b], o.relinearize(cur[b]), "relinearize item %d" % b)
    cur = nxt

    if K >= 2:
        cx.set_scale(float(primes[K - 1]) * 2.0 ** 10)
        d.ev.rescale_to_next_inplace(cx)  # evaluator.cpp:1503
        assert cx.coeff_modulus_size() == K - 1 and cx.scale() == (float(primes[K - 1]) * 2.0 ** 10) / float(primes[K - 1])
        nxt = d.out(cx)
        for b in range(batch):
            _eq(nxt[b], o.rescale(cur[b]), "rescale_to_next item %d" % b)
        cur = nxt

    for s in steps:
        # out of place first (evaluator.h:1246: destination = encrypted; rotate_vector_inplace(destination) - here the operand is
        # read where it lies): the destination gets the rotated words and the operand's metadata, the operand keeps its words
        e = o.galois_elt_from_step(s)
        dst = S.Ciphertext(d.ctx, batch=batch)
        assert d.ev.rotate_vector(cx, s, d.glk, dst) is dst
        assert dst.size() == 2 and dst.is_ntt_form() and dst.scale() == cx.scale() and dst.coeff_modulus_size() == cx.coeff_modulus_size()
        got, kept = d.out(dst), d.out(cx)
        for b in range(batch):
            _eq(got[b], o.apply_galois(cur[b], e), "rotate_vector(%d) out of place, item %d" % (s, b))
            _eq(kept[b], cur[b], "the operand of an out-of-place rotation, item %d" % b)
        # a destination of another batch size takes the reference's two steps (copy, then in place) and its shape follows the operand
        other = S.Ciphertext(d.ctx, batch=batch + 1)
        d.ev.rotate_vector(cx, s, d.glk, other)
        got = d.out(other)
        assert len(got) == batch
        for b in range(batch):
            _eq(got[b], o.apply_galois(cur[b], e), "rotate_vector(%d) into a destination of another batch size, item %d" % (s, b))
        d.ev.rotate_vector_inplace(cx, s, d.glk)  # evaluator.h:1209
        nxt = d.out(cx)
        for b in range(batch):
            _eq(nxt[b], o.apply_galois(cur[b], e), "rotate_vector(%d) item %d" % (s, b))
        cur = nxt

    if cx.coeff_modulus_size() >= 2:
        d.ev.mod_switch_to_next_inplace(cx)  # CKKS: drop last component, evaluator.cpp:1296
        nxt = d.out(cx)
        for b in range(batch):
            _eq(nxt[b], o.mod_switch_to_next(cur[b]), "mod_switch_to_next item %d" % b)
        cur = nxt

    if check_transforms:
        d.ev.transform_from_ntt_inplace(cx)  # evaluator.cpp:2337
        assert not cx.is_ntt_form()
        nxt = d.out(cx)
        for b in range(batch):
            _eq(nxt[b], o.transform(cur[b], False), "transform_from_ntt item %d" % b)
        d.ev.transform_to_ntt_inplace(cx)
        back = d.out(cx)
        for b in range(batch):
            _eq(back[b], cur[b], "transform_to_ntt(transform_from_ntt(x)) item %d" % b)

    # twice: with the digit loop cut into in-launch groups where the batch is small (the default; the pass that adds the groups
    # also adds the ciphertext's words) and as ONE group - the form large batches run, in which ks2's epilogue leaves c + S P^-1
    # behind (KsFusedArgs::fold_c0).  Both tails read one operand either way
    for one_group in ((False, True) if K >= 2 else ()):
        saved = os.environ.get("SEALHIP_KS_SPLIT")
        if one_group:
            os.environ["SEALHIP_KS_SPLIT"] = "1"
        try:
            _deferred_tails(d, o, xs, ys, primes, K, n, batch, steps, elts)
        finally:
            if one_group:
                if saved is None:
                    del os.environ["SEALHIP_KS_SPLIT"]
                else:
                    os.environ["SEALHIP_KS_SPLIT"] = saved


def _deferred_tails(d, o, xs, ys, primes, K, n, batch, steps, elts):
    if True:
        # Deferred key-switch tails (evaluator.h: LazyTail): the key switch leaves its mod-down undone, and a rescale that follows
        # without anything reading the ciphertext in between does both rounding divisions in one pass.  Same words as the
        # reference's two separate steps (evaluator.cpp:2806-2864, rns.cpp:830-901).
        sc = float(primes[K - 1]) * 2.0 ** 10
        defers = 13 <= n.bit_length() - 1 <= 16 and not os.environ.get("SEALHIP_KS_EAGER_TAIL")  # the two-pass sizes
        folded0, plain0, dropped0 = S.tail_stats()
        cz, cw = d.ct(xs, scale=2.0 ** 10), d.ct(ys, scale=2.0 ** 10)
        d.ev.multiply_inplace(cz, cw)
        d.ev.relinearize_inplace(cz, d.rlk)
        cz.set_scale(sc)
        d.ev.rescale_to_next_inplace(cz)
        assert S.tail_stats()[0] - folded0 == (1 if defers else 0), "the folded pass did not run where it should"
        assert cz.size() == 2 and cz.coeff_modulus_size() == K - 1 and cz.scale() == sc / float(primes[K - 1])
        got = d.out(cz)
        for b in range(batch):
            _eq(got[b], o.rescale(o.relinearize(o.multiply(xs[b], ys[b]))), "relinearize + rescale folded, item %d" % b)
        if steps:
            cr = d.ct(xs, scale=sc)
            d.ev.rotate_vector_inplace(cr, steps[0], d.glk)
            d.ev.rescale_to_next_inplace(cr)
            got = d.out(cr)
            for b in range(batch):
                _eq(got[b], o.rescale(o.apply_galois(xs[b], elts[0])), "rotate + rescale folded, item %d" % b)
        # a deferred tail is completed by anything else that needs the words: a copy, another operation, a second key switch
        ca, cb = d.ct(xs, scale=2.0 ** 10), d.ct(ys, scale=2.0 ** 10)
        d.ev.multiply_inplace(ca, cb)
        d.ev.relinearize_inplace(ca, d.rlk)
        cc = ca.copy()
        d.ev.add_inplace(ca, cc)
        got, one = d.out(ca), d.out(cc)
        for b in range(batch):
            r = o.relinearize(o.multiply(xs[b], ys[b]))
            _eq(one[b], r, "copy of a ciphertext with a deferred tail, item %d" % b)
            qk = np.array(primes[:K], dtype=np.uint64)[None, :, None]
            _eq(got[b], (r + r) % qk, "add after a deferred tail, item %d" % b)
        # ... or never runs when the object is overwritten or destroyed first
        folded1, plain1, dropped1 = S.tail_stats()
        cd, ce = d.ct(xs, scale=2.0 ** 10), d.ct(ys, scale=2.0 ** 10)
        d.ev.multiply_inplace(cd, ce)
        d.ev.relinearize_inplace(cd, d.rlk)
        d.ev.multiply(d.ct(xs, scale=2.0 ** 10), ce, cd)   # cd is the destination: its pending tail is discarded
        d.ev.relinearize_inplace(cd, d.rlk)
        del cd
        import gc
        gc.collect()
        folded2, plain2, dropped2 = S.tail_stats()
        assert (folded2, plain2) == (folded1, plain1) and dropped2 - dropped1 == (2 if defers else 0)


# ---- chunked key switching (sealhip.h: SealHip_KsChunkStats): the batch cut into chunks dealt to lanes
class _Env:
    """set / restore environment variables the library reads per call"""

    def __init__(self, **kv):
        self.kv, self.saved = kv, {}

    def __enter__(self):
        for k, v in self.kv.items():
            self.saved[k] = os.environ.get(k)
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = str(v)

    def __exit__(self, *a):
        for k, v in self.saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def case_ks_chunked(scheme, n, bits_or_primes, batch, chunk, lanes, t=None, seed=5, cap_mib=None):
    """relinearize (+ rescale, CKKS: the folded tail; + mod_switch, BFV) and a rotation over a batch that the key switch cuts
    into ceil(batch / chunk) chunks on `lanes` streams - ragged last chunk included - against the reference item by item, and
    the same words as the unchunked run (evaluator.cpp:2561-2867)."""
    primes = bits_or_primes if scheme != "ckks" else coeff_modulus_create(n, list(bits_or_primes))
    K = len(primes) - 1
    t = t or 0
    probe = Oracle(scheme, n, primes, t)
    elt = probe.galois_elt_from_step(1)
    o = Oracle(scheme, n, primes, t, galois_elts=[elt])
    d = DeviceSide(scheme, n, primes, t)
    d.upload_keys(o)
    rng = np.random.default_rng(seed)
    xs = [rand_ct(rng, primes, K, n) for _ in range(batch)]
    ys = [rand_ct(rng, primes, K, n) for _ in range(batch)]
    env = dict(SEALHIP_KS_SPLIT=1, SEALHIP_KS_CHUNK=chunk, SEALHIP_KS_LANES=lanes)
    if cap_mib is not None:
        env["SEALHIP_KS_SCRATCH_CAP_MIB"] = cap_mib

    def run():
        if scheme == "ckks":
            sc = float(primes[K - 1]) * 2.0 ** 10
            a, b = d.ct(xs, scale=2.0 ** 10), d.ct(ys, scale=2.0 ** 10)
            d.ev.multiply_inplace(a, b)
            d.ev.relinearize_inplace(a, d.rlk)
            a.set_scale(sc)
            d.ev.rescale_to_next_inplace(a)          # the folded tail reads the chunks' sums
            r = d.ct(xs, scale=sc)
            d.ev.rotate_vector_inplace(r, 1, d.glk)  # completed on its own (plain tail)
            return d.out(a), d.out(r)
        a, b = d.ct(xs), d.ct(ys)
        d.ev.multiply_inplace(a, b)
        d.ev.relinearize_inplace(a, d.rlk)
        d.ev.mod_switch_to_next_inplace(a)
        r = d.ct(xs)
        d.ev.apply_galois_inplace(r, elt, d.glk)
        return d.out(a), d.out(r)

    c0, k0, _ = S.ks_chunk_stats()
    with _Env(**env):
        got_a, got_r = run()
    c1, k1, held = S.ks_chunk_stats()
    nchunks = (batch + chunk - 1) // chunk
    if cap_mib is None:
        assert (c1 - c0, k1 - k0) == (2, 2 * nchunks), ("the key switch did not run in chunks", c1 - c0, k1 - k0, nchunks)
    else:
        assert c1 - c0 == 2 and k1 - k0 >= 2 * nchunks, (c1 - c0, k1 - k0)
        assert held <= cap_mib * 1048576, (held, cap_mib)
    with _Env(SEALHIP_KS_SPLIT=1, SEALHIP_KS_CHUNK=0):
        ref_a, ref_r = run()
    assert S.ks_chunk_stats()[0] == c1, "SEALHIP_KS_CHUNK=0 must keep the batch whole"
    for b in range(batch):
        _eq(got_a[b], ref_a[b], "chunked vs whole key switch, relinearize, item %d" % b)
        _eq(got_r[b], ref_r[b], "chunked vs whole key switch, rotation, item %d" % b)
        if scheme == "ckks":
            _eq(got_a[b], o.rescale(o.relinearize(o.multiply(xs[b], ys[b]))), "chunked relinearize + rescale, item %d" % b)
            _eq(got_r[b], o.apply_galois(xs[b], elt), "chunked rotation, item %d" % b)
        else:
            _eq(got_a[b], o.mod_switch_to_next(o.relinearize(o.multiply(xs[b], ys[b]))), "chunked relinearize + mod_switch, item %d" % b)
            _eq(got_r[b], o.apply_galois(xs[b], elt), "chunked apply_galois, item %d" % b)


# ---- deferred key-switch tails: who completes them (sealhip.h: SealHip_TailStats)
def case_deferred_tail_lifecycle(n=8192, bits=(50, 40, 60)):
    """K = 2 (the folded pass has ONE component to produce); a pending tail completed by a second evaluator, by the owner's
    destruction, by serialisation; the folded rescale after a rotation of a relinearised ciphertext (tail completed, new one
    deferred, then folded)."""
    import gc
    primes = coeff_modulus_create(n, list(bits))
    K = len(primes) - 1
    probe = Oracle("ckks", n, primes)
    elt = probe.galois_elt_from_step(1)
    o = Oracle("ckks", n, primes, galois_elts=[elt])
    d = DeviceSide("ckks", n, primes)
    d.upload_keys(o)
    rng = np.random.default_rng(11)
    x, y = rand_ct(rng, primes, K, n), rand_ct(rng, primes, K, n)
    ref_relin = o.relinearize(o.multiply(x, y))
    sc = float(primes[K - 1]) * 2.0 ** 10

    def relinearised(ev):
        a, b = d.ct([x], scale=2.0 ** 10), d.ct([y], scale=2.0 ** 10)
        ev.multiply_inplace(a, b)
        ev.relinearize_inplace(a, d.rlk)
        return a

    f0, p0, _ = S.tail_stats()
    # K = 2: relinearize + rescale folded
    a = relinearised(d.ev)
    a.set_scale(sc)
    d.ev.rescale_to_next_inplace(a)
    _eq(d.out(a)[0], o.rescale(ref_relin), "folded pass with one remaining component")
    # another evaluator touches the pending ciphertext: the owner completes it, the second evaluator then works on the result
    ev2 = S.Evaluator(d.ctx)
    a = relinearised(d.ev)
    a.set_scale(sc)
    ev2.rescale_to_next_inplace(a)   # not the owner: plain tail, then an ordinary rescale
    _eq(d.out(a)[0], o.rescale(ref_relin), "rescale by a second evaluator")
    # the owner goes away first
    ev3 = S.Evaluator(d.ctx)
    a = relinearised(ev3)
    del ev3
    gc.collect()
    _eq(d.out(a)[0], ref_relin, "tail completed when its evaluator was destroyed")
    # rotate a relinearised ciphertext, then rescale: the first tail runs alone, the rotation's is folded
    a = relinearised(d.ev)
    a.set_scale(sc)
    d.ev.rotate_vector_inplace(a, 1, d.glk)
    d.ev.rescale_to_next_inplace(a)
    _eq(d.out(a)[0], o.rescale(o.apply_galois(ref_relin, elt)), "relinearize, rotate, rescale")
    f1, p1, _ = S.tail_stats()
    if not os.environ.get("SEALHIP_KS_EAGER_TAIL"):
        assert (f1 - f0, p1 - p0) == (2, 3), (f1 - f0, p1 - p0)


# ---- large device-resident batches (the shapes bench.py times): inputs are generated on the device, a sample of items is
#      downloaded and compared with the reference's multiply + relinearize + rescale (+ rotate) on the same words
def host_uniform_words(primes, polys, batch, n, seed):
    """[polys][batch][len(primes)][n] uniform residues, drawn on several host threads (numpy releases the GIL); torch is not
    used: its bundled HIP runtime cannot be initialised next to the one libsealhip.so has already loaded"""
    from concurrent.futures import ThreadPoolExecutor
    out = np.empty((polys, batch, len(primes), n), dtype=np.uint64)

    def fill(job):
        p, k = job
        rng = np.random.default_rng([seed, p, k])
        out[p, :, k, :] = rng.integers(0, primes[k], size=(batch, n), dtype=np.uint64)
    with ThreadPoolExecutor(max_workers=16) as ex:
        list(ex.map(fill, [(p, k) for p in range(polys) for k in range(len(primes))]))
    return out


def _tiled_ciphertext(ctx, host, batch, pid, scale):
    """a batch of `batch` items whose item b holds host[:, b % unique] (host: [polys][unique][K][n]): the host block is uploaded
    batch / unique times into the planes, so that a large batch needs neither a large host array nor its generation time"""
    import ctypes as C
    from seal_amd import _native as N
    polys, unique, K, n = host.shape
    assert batch % unique == 0
    ct = S.Ciphertext(ctx, batch=batch)
    ct.resize(pid, polys)
    ct.set_is_ntt_form(True)
    ct.set_scale(scale)
    ptr, words = ct.device_ptr()
    assert words >= polys * batch * K * n
    for p in range(polys):
        block = np.ascontiguousarray(host[p])
        for r in range(batch // unique):
            dst = C.c_void_p(ptr + ((p * batch + r * unique) * K * n) * 8)
            N.check(N.lib().shl_memcpy_h2d(dst, block.ctypes.data_as(C.c_void_p), C.c_uint64(block.nbytes)))
    return ct


def case_ckks_big_batch(n, bits, batch, check_items, seed=11, rotate=True, unique=None):
    """unique: only that many distinct items are generated; item b repeats item b % unique (see _tiled_ciphertext)"""
    primes = coeff_modulus_create(n, bits)
    L = len(primes)
    K = L - 1
    probe = Oracle("ckks", n, primes)
    elt = probe.galois_elt_from_step(1)
    o = Oracle("ckks", n, primes, galois_elts=[elt])
    d = DeviceSide("ckks", n, primes)
    d.upload_keys(o)
    pid = d.parms_id_for_K(K)
    if unique:
        xu = host_uniform_words(primes[:K], 2, unique, n, seed)
        yu = host_uniform_words(primes[:K], 2, unique, n, seed + 1)
        cx = _tiled_ciphertext(d.ctx, xu, batch, pid, 2.0 ** 10)
        cy = _tiled_ciphertext(d.ctx, yu, batch, pid, 2.0 ** 10)

        class _Rep:  # xs[:, b] of the full batch
            def __init__(self, a):
                self.a = a

            def __getitem__(self, key):
                return self.a[:, key[1] % unique]
        xs, ys = _Rep(xu), _Rep(yu)
    else:
        xs = host_uniform_words(primes[:K], 2, batch, n, seed)
        ys = host_uniform_words(primes[:K], 2, batch, n, seed + 1)
        cx = S.Ciphertext.from_numpy(d.ctx, xs, pid, True, 2.0 ** 10)
        cy = S.Ciphertext.from_numpy(d.ctx, ys, pid, True, 2.0 ** 10)
    work = S.Ciphertext(d.ctx, batch=batch)
    d.ev.multiply(cx, cy, work)
    d.ev.relinearize_inplace(work, d.rlk)
    work.set_scale(float(primes[K - 1]) * 2.0 ** 10)
    d.ev.rescale_to_next_inplace(work)
    mid = {b: work.item_to_numpy(b) for b in check_items}
    if rotate:
        d.ev.rotate_vector_inplace(work, 1, d.glk)
    assert work.size() == 2 and work.coeff_modulus_size() == K - 1 and work.batch() == batch
    for b in check_items:
        x, y = np.ascontiguousarray(xs[:, b]), np.ascontiguousarray(ys[:, b])
        exp = o.rescale(o.relinearize(o.multiply(x, y)))
        _eq(mid[b], exp, "multiply+relinearize+rescale item %d of %d" % (b, batch))
        if rotate:
            _eq(work.item_to_numpy(b), o.apply_galois(exp, elt), "rotate_vector item %d of %d" % (b, batch))
    del cx, cy, work
    S.release_pool()


# ---- BFV: BFVEncryptMultiplyDecrypt / BFVRelinearize / BFVEncryptModSwitchToNextDecrypt /
#      BFVEncryptRotateMatrixDecrypt (native/tests/seal/evaluator.cpp:1356, 2430, 5722, 5670)
def case_bfv_pipeline(n, primes, t, batch=2, seed=4):
    L = len(primes)
    K = L - 1
    probe = Oracle("bfv", n, primes, t)
    elts = [probe.galois_elt_from_step(1), 2 * n - 1]
    o = Oracle("bfv", n, primes, t, galois_elts=elts)
    d = DeviceSide("bfv", n, primes, t)
    d.upload_keys(o)
    ci = d.chain_index_for_K(K)
    assert d.ctx.base_bsk(ci) == o.base_bsk(K), "BEHZ base Bsk"
    rng = np.random.default_rng(seed)
    xs = [rand_ct(rng, primes, K, n) for _ in range(batch)]
    ys = [rand_ct(rng, primes, K, n) for _ in range(batch)]
    cx, cy = d.ct(xs), d.ct(ys)

    # out of place first: the product goes straight into a third object (no copy of encrypted1), the operands stay what they were
    dest = S.Ciphertext(d.ctx, batch=batch)
    d.ev.multiply(cx, cy, dest)
    assert dest.size() == 3 and not dest.is_ntt_form() and dest.scale() == cx.scale() and dest.correction_factor() == 1
    away = d.out(dest)
    kept = d.out(cx)
    for b in range(batch):
        _eq(away[b], o.multiply(xs[b], ys[b]), "bfv multiply (out of place) item %d" % b)
        _eq(kept[b], xs[b], "bfv multiply (out of place) left encrypted1 alone, item %d" % b)
    d.ev.multiply_inplace(cx, cy)  # bfv_multiply, evaluator.cpp:395
    assert cx.size() == 3 and not cx.is_ntt_form()
    cur = d.out(cx)
    for b in range(batch):
        _eq(cur[b], o.multiply(xs[b], ys[b]), "bfv multiply item %d" % b)

    d.ev.relinearize_inplace(cx, d.rlk)
    nxt = d.out(cx)
    for b in range(batch):
        _eq(nxt[b], o.relinearize(cur[b]), "bfv relinearize item %d" % b)
    cur = nxt

    # out of place: the destination (here one that held something else) gets the result, the operand keeps its words
    dst = cx.copy()
    d.ev.rotate_rows(cx, 1, d.glk, dst)
    got, kept = d.out(dst), d.out(cx)
    for b in range(batch):
        _eq(got[b], o.apply_galois(cur[b], elts[0]), "rotate_rows(1) out of place, item %d" % b)
        _eq(kept[b], cur[b], "the operand of an out-of-place rotation, item %d" % b)
    d.ev.rotate_columns(cx, d.glk, dst)
    got = d.out(dst)
    for b in range(batch):
        _eq(got[b], o.apply_galois(cur[b], elts[1]), "rotate_columns out of place, item %d" % b)
    d.ev.apply_galois(cx, elts[0], d.glk, dst)
    got = d.out(dst)
    for b in range(batch):
        _eq(got[b], o.apply_galois(cur[b], elts[0]), "apply_galois out of place, item %d" % b)

    d.ev.rotate_rows_inplace(cx, 1, d.glk)
    nxt = d.out(cx)
    for b in range(batch):
        _eq(nxt[b], o.apply_galois(cur[b], elts[0]), "rotate_rows(1) item %d" % b)
    cur = nxt

    d.ev.rotate_columns_inplace(cx, d.glk)
    nxt = d.out(cx)
    for b in range(batch):
        _eq(nxt[b], o.apply_galois(cur[b], elts[1]), "rotate_columns item %d" % b)
    cur = nxt

    if K >= 2:
        d.ev.mod_switch_to_next_inplace(cx)  # divide_and_round_q_last_inplace, rns.cpp:789
        assert cx.coeff_modulus_size() == K - 1
        nxt = d.out(cx)
        for b in range(batch):
            _eq(nxt[b], o.mod_switch_to_next(cur[b]), "bfv mod_switch_to_next item %d" % b)
        cur = nxt

    d.ev.square_inplace(cx)  # bfv_square, evaluator.cpp:878
    nxt = d.out(cx)
    for b in range(batch):
        _eq(nxt[b], o.multiply(cur[b], cur[b]), "bfv square item %d" % b)

    if K >= 2:
        # Deferred key-switch tail, BFV (round 4): relinearize / rotate_rows followed directly by mod_switch_to_next - nothing reads the
        # ciphertext in between - does the mod-down by the special prime and the division by q_last in ONE element-wise pass
        # (evaluator.cpp:2806-2864 then rns.cpp:789-828: the same words as the reference's two steps); a reader in between gets the
        # completed ciphertext, a destination that is overwritten drops the pending sums.
        defers = 13 <= n.bit_length() - 1 <= 16 and not os.environ.get("SEALHIP_KS_EAGER_TAIL")  # the two-pass sizes
        folded0, plain0, dropped0 = S.tail_stats()
        cz, cw = d.ct(xs), d.ct(ys)
        d.ev.multiply_inplace(cz, cw)
        d.ev.relinearize_inplace(cz, d.rlk)
        d.ev.mod_switch_to_next_inplace(cz)
        assert S.tail_stats()[0] - folded0 == (1 if defers else 0), "the folded BFV pass did not run where it should"
        assert cz.size() == 2 and cz.coeff_modulus_size() == K - 1 and not cz.is_ntt_form()
        got = d.out(cz)
        for b in range(batch):
            _eq(got[b], o.mod_switch_to_next(o.relinearize(o.multiply(xs[b], ys[b]))), "bfv relinearize + mod_switch folded, item %d" % b)
        cr = d.ct(xs)
        d.ev.rotate_rows_inplace(cr, 1, d.glk)
        d.ev.mod_switch_to_next_inplace(cr)
        got = d.out(cr)
        for b in range(batch):
            _eq(got[b], o.mod_switch_to_next(o.apply_galois(xs[b], elts[0])), "bfv rotate_rows + mod_switch folded, item %d" % b)
        # completed by a copy, then the ordinary mod switch; and dropped unrun when the destination is overwritten
        ca, cb = d.ct(xs), d.ct(ys)
        d.ev.multiply_inplace(ca, cb)
        d.ev.relinearize_inplace(ca, d.rlk)
        cc = ca.copy()
        d.ev.mod_switch_to_next_inplace(ca)
        got, one = d.out(ca), d.out(cc)
        for b in range(batch):
            r = o.relinearize(o.multiply(xs[b], ys[b]))
            _eq(one[b], r, "bfv copy of a ciphertext with a deferred tail, item %d" % b)
            _eq(got[b], o.mod_switch_to_next(r), "bfv mod_switch after the tail was completed by a copy, item %d" % b)
        f1, p1, dr1 = S.tail_stats()
        ce, cf = d.ct(xs), d.ct(ys)
        d.ev.multiply_inplace(ce, cf)
        d.ev.relinearize_inplace(ce, d.rlk)
        d.ev.multiply(d.ct(xs), cf, ce)        # ce is the destination: its pending tail is discarded
        got = d.out(ce)
        for b in range(batch):
            _eq(got[b], o.multiply(xs[b], ys[b]), "bfv product written over a ciphertext with a deferred tail, item %d" % b)
        if defers:
            assert S.tail_stats()[2] - dr1 == 1, "the overwritten ciphertext's tail was not discarded"


# ---- BGV: BGVEncryptMultiplyDecrypt / BGVRelinearize / BGVEncryptModSwitchToNextDecrypt / BGVEncryptRotateMatrixDecrypt /
#      BGVEncryptAddDecrypt with unequal correction factors (native/tests/seal/evaluator.cpp, BGV cases) at ciphertext level.
#      Reference oracle only (the plain-C restatement does not cover BGV).
def case_bgv_pipeline(n, primes, t, batch=2, seed=6):
    L = len(primes)
    K = L - 1
    probe = Oracle("bgv", n, primes, t)
    assert probe.kind == "reference", "BGV parity needs oracle/_ref"
    elts = [probe.galois_elt_from_step(1), 2 * n - 1]
    o = Oracle("bgv", n, primes, t, galois_elts=elts)
    d = DeviceSide("bgv", n, primes, t)
    d.upload_keys(o)
    rng = np.random.default_rng(seed)
    xs = [rand_ct(rng, primes, K, n) for _ in range(batch)]
    ys = [rand_ct(rng, primes, K, n) for _ in range(batch)]
    cfx, cfy = 3, 5 % t if t > 5 else 1

    def dev(slabs, cf):
        c = d.ct(slabs, is_ntt=True)
        c.set_correction_factor(cf)
        return c

    # add / sub with unequal correction factors: balance_correction_factors (evaluator.cpp:50-117, 173-192)
    for op in ("add_inplace", "sub_inplace"):
        cx, cy = dev(xs, cfx), dev(ys, cfy)
        getattr(d.ev, op)(cx, cy)
        got = d.out(cx)
        for b in range(batch):
            exp, info = o.run(op, [(xs[b], cfx), (ys[b], cfy)])
            _eq(got[b], exp, "bgv %s item %d" % (op, b))
            assert cx.correction_factor() == info["correction_factor"], op

    cx, cy = dev(xs, cfx), dev(ys, cfy)
    d.ev.multiply_inplace(cx, cy)  # bgv_multiply, evaluator.cpp:710
    assert cx.size() == 3 and cx.is_ntt_form()
    cur = d.out(cx)
    for b in range(batch):
        exp, info = o.run("multiply_inplace", [(xs[b], cfx), (ys[b], cfy)])
        _eq(cur[b], exp, "bgv multiply item %d" % b)
        assert cx.correction_factor() == info["correction_factor"]
    cf = cx.correction_factor()

    d.ev.relinearize_inplace(cx, d.rlk)  # switch_key_inplace BGV branch, evaluator.cpp:2762
    nxt = d.out(cx)
    for b in range(batch):
        exp, info = o.run("relinearize_inplace", [(cur[b], cf)])
        _eq(nxt[b], exp, "bgv relinearize item %d" % b)
    cur = nxt

    d.ev.rotate_rows_inplace(cx, 1, d.glk)
    nxt = d.out(cx)
    for b in range(batch):
        exp, _ = o.run("apply_galois_inplace", [(cur[b], cf)], elts[0])
        _eq(nxt[b], exp, "bgv rotate_rows(1) item %d" % b)
    cur = nxt

    d.ev.rotate_columns_inplace(cx, d.glk)
    nxt = d.out(cx)
    for b in range(batch):
        exp, _ = o.run("apply_galois_inplace", [(cur[b], cf)], elts[1])
        _eq(nxt[b], exp, "bgv rotate_columns item %d" % b)
    cur = nxt

    if K >= 2:
        d.ev.mod_switch_to_next_inplace(cx)  # mod_t_and_divide_q_last_ntt_inplace, rns.cpp:1193
        assert cx.coeff_modulus_size() == K - 1
        nxt = d.out(cx)
        for b in range(batch):
            exp, info = o.run("mod_switch_to_next_inplace", [(cur[b], cf)])
            _eq(nxt[b], exp, "bgv mod_switch_to_next item %d" % b)
            assert cx.correction_factor() == info["correction_factor"]
        cur, cf = nxt, cx.correction_factor()

    d.ev.square_inplace(cx)  # bgv_square, evaluator.cpp:1079
    nxt = d.out(cx)
    for b in range(batch):
        exp, info = o.run("square_inplace", [(cur[b], cf)])
        _eq(nxt[b], exp, "bgv square item %d" % b)
        assert cx.correction_factor() == info["correction_factor"]


# ---- digit-parallel key switching (sealhip.h section 1b) emulated in ONE process: the digits are split
#      over `parts` virtual ranks, each with only its key slice resident; the partial sums are added on the host (what the
#      all-reduce does) and every result must equal the single-GPU relinearize / rotate AND the oracle, bit for bit.
def case_digit_parallel(scheme, n, primes, t=0, parts=2, batch=2, seed=7):
    from seal_amd import shard
    L = len(primes)
    K = L - 1
    probe = Oracle(scheme, n, primes, t)
    elt = probe.galois_elt_from_step(1)
    o = Oracle(scheme, n, primes, t, galois_elts=[elt])
    d = DeviceSide(scheme, n, primes, t)
    d.upload_keys(o)
    rlk_words, glk_words = o.relin_key(), o.galois_key(elt)
    rng = np.random.default_rng(seed)
    x3 = [rand_ct(rng, primes, K, n, size=3) for _ in range(batch)]
    x2 = [rand_ct(rng, primes, K, n, size=2) for _ in range(batch)]
    ranges = [shard.split(K, parts, r) for r in range(parts)]
    assert sum(c for _, c in ranges) == K
    # one key object per virtual rank, holding only that rank's digits
    rlks, glks = [], []
    for first, count in ranges:
        rk, gk = S.RelinKeys(d.ctx), S.GaloisKeys(d.ctx)
        if count:
            rk.set_key_digits(0, first, rlk_words[first:first + count])
            gk.set_key_digits(S.GaloisKeys.get_index(elt), first, glk_words[first:first + count])
        else:  # a rank without digits still takes part in the exchange
            rk.set_key(0, rlk_words)
            gk.set_key(S.GaloisKeys.get_index(elt), glk_words)
        rlks.append(rk)
        glks.append(gk)

    def run(make_ct, partial, finish, single):
        ref_ct = make_ct()
        single(ref_ct)
        want = d.out(ref_ct)
        words = d.ev.switch_key_acc_words(make_ct())
        total = np.zeros(words, dtype=np.uint64)
        cts = []
        for r, (first, count) in enumerate(ranges):
            c = make_ct()
            acc = S.DeviceBuffer(words)
            partial(c, r, first, count, acc)
            total += acc.to_numpy((words,))
            cts.append(c)
        for c in cts:  # every rank finishes on the same sum
            acc = S.DeviceBuffer.from_numpy(total)
            finish(c, acc)
            got = d.out(c)
            for b in range(batch):
                _eq(got[b], want[b], "digit-parallel == single GPU, item %d" % b)
        return want

    want = run(lambda: d.ct(x3, is_ntt=scheme != "bfv"),
               lambda c, r, f, cnt, acc: d.ev.relinearize_partial(c, rlks[r], f, cnt, acc.ptr),
               lambda c, acc: d.ev.relinearize_finish(c, acc.ptr, parts),
               lambda c: d.ev.relinearize_inplace(c, d.rlk))
    is_ntt = scheme != "bfv"
    for b in range(batch):
        exp = o.run("relinearize_inplace", [(x3[b], 1)])[0] if scheme == "bgv" else o.relinearize(x3[b])
        _eq(want[b], exp, "relinearize item %d vs oracle" % b)
    want = run(lambda: d.ct(x2, is_ntt=is_ntt),
               lambda c, r, f, cnt, acc: d.ev.apply_galois_partial(c, elt, glks[r], f, cnt, acc.ptr),
               lambda c, acc: d.ev.apply_galois_finish(c, acc.ptr, parts),
               lambda c: d.ev.apply_galois_inplace(c, elt, d.glk))
    for b in range(batch):
        exp = o.run("apply_galois_inplace", [(x2[b], 1)], elt)[0] if scheme == "bgv" else o.apply_galois(x2[b], elt)
        _eq(want[b], exp, "apply_galois item %d vs oracle" % b)
    if scheme == "ckks" and K >= 2:
        # BASELINE configs[4] is rotate + rescale: at the two-pass sizes `*Finish` leaves the mod-down pending like the single-GPU
        # key switch does and the rescale folds both divisions (sealhip.h section 1b); smaller sizes finish at once
        defers = 13 <= n.bit_length() - 1 <= 16 and not os.environ.get("SEALHIP_KS_EAGER_TAIL")
        sc = float(primes[K - 1]) * 2.0 ** 10
        words = d.ev.switch_key_acc_words(d.ct(x2, scale=sc))
        total = np.zeros(words, dtype=np.uint64)
        cts = []
        for r, (first, count) in enumerate(ranges):
            c = d.ct(x2, scale=sc)
            acc = S.DeviceBuffer(words)
            d.ev.apply_galois_partial(c, elt, glks[r], first, count, acc.ptr)
            total += acc.to_numpy((words,))
            cts.append(c)
        for c in cts:
            folded0 = S.tail_stats()[0]
            acc = S.DeviceBuffer.from_numpy(total)
            d.ev.apply_galois_finish(c, acc.ptr, parts)
            del acc  # the caller's buffer is not referenced by the pending tail
            d.ev.rescale_to_next_inplace(c)
            assert S.tail_stats()[0] - folded0 == (1 if defers else 0), "digit-parallel finish + rescale: folded pass expected %s" % defers
            got = d.out(c)
            for b in range(batch):
                _eq(got[b], o.rescale(o.apply_galois(x2[b], elt)), "digit-parallel rotate + rescale, item %d" % b)


def case_digit_parallel_reduce_scatter(n, primes, parts=2, batch=2, seed=9):
    """The reduce-scatter shape of the exchange (sealhip.h section 1c, exchange 1; CKKS) with the ranks emulated in one process:
    partial sums per rank -> pack by owner -> [reduce-scatter + all-reduce of the special component, here numpy sums] ->
    mod-down of the rank's own moduli -> [all-gather, here a concatenation] -> add.  Every rank's result equals the
    single-GPU result and the reference; then the same through the library's own driver on a one-rank communicator."""
    from seal_amd import shard
    L = len(primes)
    K = L - 1
    probe = Oracle("ckks", n, primes)
    elt = probe.galois_elt_from_step(1)
    o = Oracle("ckks", n, primes, galois_elts=[elt])
    d = DeviceSide("ckks", n, primes)
    d.upload_keys(o)
    rlk_words, glk_words = o.relin_key(), o.galois_key(elt)
    rng = np.random.default_rng(seed)
    x3 = [rand_ct(rng, primes, K, n, size=3) for _ in range(batch)]
    x2 = [rand_ct(rng, primes, K, n, size=2) for _ in range(batch)]
    G = parts
    ranges = [shard.split(K, G, r) for r in range(G)]
    rlks, glks = [], []
    for first, count in ranges:
        rk, gk = S.RelinKeys(d.ctx), S.GaloisKeys(d.ctx)
        if count:  # a rank without digits holds no key at all
            rk.set_key_digits(0, first, rlk_words[first:first + count])
            gk.set_key_digits(S.GaloisKeys.get_index(elt), first, glk_words[first:first + count])
        rlks.append(rk)
        glks.append(gk)

    def run(make_ct, partial, single, planes_kept):
        ref_ct = make_ct()
        single(ref_ct)
        want = d.out(ref_ct)
        probe_ct = make_ct()
        words = d.ev.switch_key_acc_words(probe_ct)
        m = d.ev.switch_key_slots(probe_ct, G)
        assert m == -(-K // G)
        chunk = m * batch * 2 * n
        cts, sends, sps = [], [], []
        for r, (first, count) in enumerate(ranges):
            c = make_ct()
            acc, send, sp = S.DeviceBuffer(words), S.DeviceBuffer(G * chunk), S.DeviceBuffer(batch * 2 * n)
            partial(c, r, first, count, acc)
            d.ev.switch_key_pack_targets(c, acc.ptr, G, send.ptr, sp.ptr)
            sends.append(send.to_numpy((G, chunk)))
            sps.append(sp.to_numpy((batch * 2 * n,)))
            cts.append(c)
        sp_sum = np.sum(np.stack(sps), axis=0, dtype=np.uint64)
        owns = []
        for r in range(G):
            recv = np.sum(np.stack([sends[src][r] for src in range(G)]), axis=0, dtype=np.uint64)  # reduce-scatter
            own, recv_d, sp_d = S.DeviceBuffer(chunk), S.DeviceBuffer.from_numpy(recv), S.DeviceBuffer.from_numpy(sp_sum)
            d.ev.switch_key_finish_owned(cts[r], recv_d.ptr, sp_d.ptr, G, r, own.ptr)
            owns.append(own.to_numpy((chunk,)))  # (synchronises: the inputs may be released now)
        gathered = S.DeviceBuffer.from_numpy(np.concatenate(owns))  # all-gather
        for r in range(G):
            d.ev.switch_key_add_gathered(cts[r], gathered.ptr, G)
            got = d.out(cts[r])
            for b in range(batch):
                _eq(got[b][:planes_kept], want[b][:planes_kept], "reduce-scatter exchange, rank %d of %d, item %d" % (r, G, b))
        return want

    want = run(lambda: d.ct(x3), lambda c, r, f, cnt, acc: d.ev.relinearize_partial(c, rlks[r], f, cnt, acc.ptr),
               lambda c: d.ev.relinearize_inplace(c, d.rlk), 2)
    for b in range(batch):
        _eq(want[b], o.relinearize(x3[b]), "relinearize item %d vs oracle" % b)
    want = run(lambda: d.ct(x2), lambda c, r, f, cnt, acc: d.ev.apply_galois_partial(c, elt, glks[r], f, cnt, acc.ptr),
               lambda c: d.ev.apply_galois_inplace(c, elt, d.glk), 2)
    for b in range(batch):
        _eq(want[b], o.apply_galois(x2[b], elt), "apply_galois item %d vs oracle" % b)
    # the library's own driver on a one-rank communicator (real RCCL on the GPU box, loopback under emulation), both shapes
    comm = S.Comm(S.Comm.unique_id(), 1, 0)
    assert comm.digit_range(K) == (0, K)
    for exchange in (S.Comm.ALL_REDUCE, S.Comm.REDUCE_SCATTER):
        c3, c2, cr = d.ct(x3), d.ct(x2), d.ct(x2)
        d.ev.relinearize_inplace_dp(c3, d.rlk, comm, exchange)
        d.ev.apply_galois_inplace_dp(c2, elt, d.glk, comm, exchange)
        d.ev.rotate_vector_inplace_dp(cr, 1, d.glk, comm, exchange)
        assert c3.size() == 2
        g3, g2, gr = d.out(c3), d.out(c2), d.out(cr)
        for b in range(batch):
            _eq(g3[b], o.relinearize(x3[b]), "driver relinearize (exchange %d) item %d" % (exchange, b))
            _eq(g2[b], o.apply_galois(x2[b], elt), "driver apply_galois (exchange %d) item %d" % (exchange, b))
            _eq(gr[b], o.apply_galois(x2[b], elt), "driver rotate_vector (exchange %d) item %d" % (exchange, b))
    # one-time key distribution: the full key is staged on the device, "broadcast" over the one-rank communicator, and the
    # evaluator keeps this rank's digits (all of them here)
    stage = S.DeviceBuffer.from_numpy(rlk_words)
    rk = S.RelinKeys(d.ctx)
    d.ev.broadcast_key_digits(rk, 0, stage.ptr, comm, 0)
    c3 = d.ct(x3)
    d.ev.relinearize_inplace_dp(c3, rk, comm, S.Comm.REDUCE_SCATTER)
    g3 = d.out(c3)
    for b in range(batch):
        _eq(g3[b], o.relinearize(x3[b]), "relinearize with a broadcast key, item %d" % b)
    return comm.loopback()


# ---- plaintext operands and many-operand forms: add_plain / sub_plain / multiply_plain in every form
#      combination, transform_to_ntt(Plaintext), mod_switch_to_next(Plaintext), add_many, multiply_many, exponentiate
#      (native/tests/seal/evaluator.cpp: *AddPlain*, *SubPlain*, *MultiplyPlain*, *MultiplyMany*, *Exponentiate*,
#      TransformPlainToNTT) at ciphertext level against the real reference.
def case_plain_ops(scheme, n, primes, t=0, batch=2, seed=8):
    L = len(primes)
    K = L - 1
    o = Oracle(scheme, n, primes, t)
    assert o.kind == "reference", "plaintext-operand parity needs oracle/_ref"
    d = DeviceSide(scheme, n, primes, t)
    d.upload_keys(o)
    rng = np.random.default_rng(seed)
    ci = o._ci(K)
    pid = d.parms_id_for_K(K)
    is_ntt = scheme != "bfv"
    xs = [rand_ct(rng, primes, K, n) for _ in range(batch)]
    cf = 3 if scheme == "bgv" else 1
    scale = 2.0 ** 10 if scheme == "ckks" else 1.0

    def dev_ct(slabs=None, ntt=None):
        c = d.ct(xs if slabs is None else slabs, scale=scale, is_ntt=is_ntt if ntt is None else ntt)
        c.set_correction_factor(cf)
        return c

    def ref_ct(slab, ntt=None):
        return o.ref.ct(ci, slab, is_ntt if ntt is None else ntt, scale, cf)

    def check(name, cdev, ref_fn, slabs=None, ntt=None):
        got = d.out(cdev)
        for b in range(batch):
            r = ref_ct((xs if slabs is None else slabs)[b], ntt)
            ref_fn(r)
            _eq(got[b], r.data(), "%s item %d" % (name, b))
            i = r.info()
            assert cdev.scale() == i["scale"] and cdev.correction_factor() == i["correction_factor"], name
            assert cdev.is_ntt_form() == i["is_ntt_form"], name

    if scheme == "ckks":
        pr = np.stack([rng.integers(0, primes[i], n, dtype=np.uint64) for i in range(K)])
        pl_d = S.Plaintext.from_numpy(d.ctx, pr, pid, scale)
        pl_r = lambda: o.ref.pt(pr, ci, scale)
        for op in ("add_plain_inplace", "sub_plain_inplace", "multiply_plain_inplace"):
            c = dev_ct()
            getattr(d.ev, op)(c, pl_d)
            check(op, c, lambda r: getattr(o.ref, op)(r, pl_r()))
        # mod_switch_to_next(Plaintext): drop the last component
        if K >= 2:
            pm = pl_d.copy()
            d.ev.mod_switch_plain_to_next_inplace(pm)
            rp = o.ref.pt_mod_switch_to_next_inplace(pl_r())
            assert pm.coeff_count() == rp.info()["coeff_count"] and np.array_equal(pm.to_numpy(), rp.data())
    else:
        # coefficient-form plaintexts: full length with values on both sides of the (t+1)/2 threshold, a short one
        # (coeff_count < N) and a monomial (the reference's negacyclic_multiply_poly_mono shortcut)
        full = rng.integers(0, t, n, dtype=np.uint64)
        short = rng.integers(0, t, max(1, n // 4), dtype=np.uint64)
        mono = np.zeros(5, dtype=np.uint64)
        mono[4] = t - 2
        for name, m in (("full", full), ("short", short), ("mono", mono)):
            pl_d = S.Plaintext.from_numpy(d.ctx, m)
            pl_r = lambda m=m: o.ref.pt(m)
            for op in ("add_plain_inplace", "sub_plain_inplace", "multiply_plain_inplace"):
                c = dev_ct()
                getattr(d.ev, op)(c, pl_d)
                check("%s(%s)" % (op, name), c, lambda r: getattr(o.ref, op)(r, pl_r()))
            # transform_to_ntt_inplace(Plaintext, parms_id), then multiply_plain on the other form combinations
            pn = pl_d.copy()
            d.ev.transform_plain_to_ntt_inplace(pn, pid)
            rn = o.ref.pt_transform_to_ntt_inplace(pl_r(), ci)
            assert pn.is_ntt_form() and np.array_equal(pn.to_numpy(), rn.data()), "transform_to_ntt(Plaintext) %s" % name
            c = dev_ct(ntt=not is_ntt)  # the other ciphertext form
            if scheme == "bfv":  # BFV ciphertext in NTT form x NTT plain, and x coefficient-form plain
                d.ev.multiply_plain_inplace(c, pn)
                check("multiply_plain ntt x ntt (%s)" % name, c, lambda r: o.ref.multiply_plain_inplace(r, rn), ntt=True)
                c = dev_ct(ntt=True)
                d.ev.multiply_plain_inplace(c, pl_d)
                check("multiply_plain ntt x coeff (%s)" % name, c, lambda r: o.ref.multiply_plain_inplace(r, pl_r()), ntt=True)
                c = dev_ct()
                d.ev.multiply_plain_inplace(c, pn)
                check("multiply_plain coeff x ntt (%s)" % name, c, lambda r: o.ref.multiply_plain_inplace(r, rn))
            else:        # BGV ciphertext (NTT form) x NTT plain
                c = dev_ct()
                d.ev.multiply_plain_inplace(c, pn)
                check("multiply_plain ntt x ntt (%s)" % name, c, lambda r: o.ref.multiply_plain_inplace(r, rn))

    # add_many
    ys = [rand_ct(rng, primes, K, n) for _ in range(batch)]
    zs = [rand_ct(rng, primes, K, n) for _ in range(batch)]
    dest = S.Ciphertext(d.ctx, batch=batch)
    d.ev.add_many([dev_ct(xs), dev_ct(ys), dev_ct(zs)], dest)
    got = d.out(dest)
    for b in range(batch):
        r = o.ref.add_many([ref_ct(xs[b]), ref_ct(ys[b]), ref_ct(zs[b])])
        _eq(got[b], r.data(), "add_many item %d" % b)
    if scheme != "ckks":
        # multiply_many (5 operands: products, relinearizations, the odd one carried) and exponentiate (x^3)
        ops = [[rand_ct(rng, primes, K, n) for _ in range(batch)] for _ in range(5)]
        dest = S.Ciphertext(d.ctx, batch=batch)
        d.ev.multiply_many([dev_ct(s) for s in ops], d.rlk, dest)
        got = d.out(dest)
        for b in range(batch):
            r = o.ref.multiply_many([ref_ct(s[b]) for s in ops])
            _eq(got[b], r.data(), "multiply_many item %d" % b)
            assert dest.correction_factor() == r.info()["correction_factor"]
        c = dev_ct()
        d.ev.exponentiate_inplace(c, 3, d.rlk)
        check("exponentiate(3)", c, lambda r: o.ref.exponentiate_inplace(r, 3))


# ---- BEHZ stages: native/tests/seal/util/rns.cpp:460-854
def case_rns_stages(n, primes, t, seed=5):
    L = len(primes)
    K = L - 1
    o = Oracle("bfv", n, primes, t)
    d = DeviceSide("bfv", n, primes, t)
    ci = d.chain_index_for_K(K)
    nBsk = len(o.base_bsk(K))
    rng = np.random.default_rng(seed)
    x = np.stack([rng.integers(0, primes[i], n, dtype=np.uint64) for i in range(K)])

    def run(which, inp, out_comps):
        src, dst = S.DeviceBuffer.from_numpy(inp), S.DeviceBuffer(out_comps * n)
        S.rns_stage(d.ctx, ci, which, src, dst, 1)
        got = dst.to_numpy((out_comps, n))
        _eq(got, o.rns_stage(K, which, inp, out_comps), which)
        return got

    e0 = run("fastbconv_m_tilde", x, nBsk + 1)
    e1 = run("sm_mrq", e0, nBsk)
    e2 = run("fast_floor", np.concatenate([x, e1]), nBsk)
    run("fastbconv_sk", e2, K)


def default_bfv_params(n, bits, t_bits):
    return coeff_modulus_create(n, bits), plain_modulus_batching(n, t_bits)


# ---- digests of the REAL reference's outputs at two-pass-engine sizes (tests/golden/make_golden_engine.py):
#      inputs and keys are regenerated from the stored seeds, outputs compared by SHA-256
def case_golden_engine(name):
    import hashlib
    import json
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    g = json.load(open(os.path.join(here, "golden", "engine_digests.json")))[name]
    n, bits = g["n"], g["bits"]
    primes = coeff_modulus_create(n, bits)
    L, K = len(primes), len(primes) - 1
    rng = np.random.default_rng(g["seed"])
    ct = lambda: np.stack([np.stack([rng.integers(0, primes[i], n, dtype=np.uint64) for i in range(K)]) for _ in range(2)])
    a, b = ct(), ct()
    key = lambda: np.stack([np.stack([np.stack([rng.integers(0, primes[i], n, dtype=np.uint64) for i in range(L)])
                                      for _ in range(2)]) for _ in range(K)])
    rlk, glk = key(), key()
    dg = lambda arr: hashlib.sha256(np.ascontiguousarray(arr, dtype=np.uint64).tobytes()).hexdigest()
    want = g["digests"]
    d = DeviceSide("ckks", n, primes)
    elt = g["galois_elt"]
    assert d.ctx.galois_elt_from_step(1) == elt
    d.rlk = S.RelinKeys(d.ctx)
    d.rlk.set_key(0, rlk)
    d.glk = S.GaloisKeys(d.ctx)
    d.glk.set_key(S.GaloisKeys.get_index(elt), glk)
    buf = S.DeviceBuffer.from_numpy(a[0])
    S.ntt_forward(d.ctx, buf, 1, K)
    assert dg(buf.to_numpy(a[0].shape)) == want["ntt_fwd_a0"], "ntt_negacyclic_harvey"
    buf = S.DeviceBuffer.from_numpy(a[0])
    S.ntt_inverse(d.ctx, buf, 1, K)
    assert dg(buf.to_numpy(a[0].shape)) == want["ntt_inv_a0"], "inverse_ntt_negacyclic_harvey"
    x, y = d.ct(a, scale=2.0 ** 10), d.ct(b, scale=2.0 ** 10)
    d.ev.multiply_inplace(x, y)
    assert dg(d.out(x)[0]) == want["multiply"], "multiply"
    d.ev.relinearize_inplace(x, d.rlk)
    assert dg(d.out(x)[0]) == want["relinearize"], "relinearize"
    x.set_scale(float(primes[K - 1]) * 2.0 ** 10)
    d.ev.rescale_to_next_inplace(x)
    assert dg(d.out(x)[0]) == want["rescale"], "rescale_to_next"
    d.ev.rotate_vector_inplace(x, 1, d.glk)
    assert dg(d.out(x)[0]) == want["rotate1"], "rotate_vector(1)"


# ---- mod_reduce_to_next / mod_reduce_to (evaluator.cpp:1597-1647): CKKS drops the last prime without scaling
def case_mod_reduce(n, bits, batch=2, seed=31):
    import sealref
    primes = coeff_modulus_create(n, bits)
    K = len(primes) - 1
    ref = sealref.RefContext("ckks", n, primes, 0)
    d = DeviceSide("ckks", n, primes)
    rng = np.random.default_rng(seed)
    xs = [rand_ct(rng, primes, K, n) for _ in range(batch)]
    cx = d.ct(xs, scale=2.0 ** 30)
    rs = [ref.ct(ref.first_chain_index, x, True, 2.0 ** 30) for x in xs]
    d.ev.mod_reduce_to_next_inplace(cx)
    for b in range(batch):
        ref.mod_reduce_to_next_inplace(rs[b])
        _eq(d.out(cx)[b], rs[b].data(), "mod_reduce_to_next item %d" % b)
    assert cx.scale() == rs[0].info()["scale"] and cx.coeff_modulus_size() == K - 1
    # all the way down with mod_reduce_to
    d.ev.mod_reduce_to_inplace(cx, d.ctx.parms_id_at(0))
    for b in range(batch):
        while rs[b].info()["chain_index"] > 0:
            ref.mod_reduce_to_next_inplace(rs[b])
        _eq(d.out(cx)[b], rs[b].data(), "mod_reduce_to item %d" % b)
    assert cx.coeff_modulus_size() == 1
    import seal_amd as S
    for bad in (lambda: d.ev.mod_reduce_to_next_inplace(cx),                       # end of chain
                lambda: d.ev.mod_reduce_to_inplace(cx, d.ctx.parms_id_at(1)),      # higher level
                lambda: d.ev.mod_reduce_to_inplace(cx, (1, 2, 3, 4))):             # unknown parms_id
        try:
            bad()
            raise AssertionError("expected InvalidArgument")
        except S.InvalidArgument:
            pass


# ---- the multi-level forms against the reference's own (evaluator.cpp:1451-1473 mod_switch_to_inplace(Ciphertext),
#      1543-1595 rescale_to_inplace): three and more levels in one call, directly and right after a key switch (CKKS at the
#      two-pass sizes: the key switch's mod-down is still pending - LazyTail - when the call arrives)
def case_multi_level_ckks(n, bits, batch=2, seed=41, step=1):
    import sealref
    primes = coeff_modulus_create(n, bits)
    L = len(primes)
    K = L - 1
    assert K >= 4
    probe = Oracle("ckks", n, primes)
    elt = probe.galois_elt_from_step(step)
    o = Oracle("ckks", n, primes, galois_elts=[elt], kind="reference")
    ref = o.ref
    d = DeviceSide("ckks", n, primes)
    d.upload_keys(o)
    rng = np.random.default_rng(seed)
    xs = [rand_ct(rng, primes, K, n) for _ in range(batch)]
    ys = [rand_ct(rng, primes, K, n) for _ in range(batch)]
    defers = 13 <= n.bit_length() - 1 <= 16 and not os.environ.get("SEALHIP_KS_EAGER_TAIL")
    levels = 3
    target_K = K - levels
    dropped = 1.0
    for i in range(target_K, K):
        dropped *= float(primes[i])
    sc = dropped * 2.0 ** 10     # lands on 2^10 (up to rounding) after three divisions
    pid = d.parms_id_for_K(target_K)
    tci = o._ci(target_K)

    def same(dev, refs, what):
        got = d.out(dev)
        for b in range(batch):
            _eq(got[b], refs[b].data(), "%s item %d" % (what, b))
        info = refs[0].info()
        assert dev.coeff_modulus_size() == info["coeff_modulus_size"] == target_K and dev.size() == info["size"], what
        assert dev.scale() == info["scale"], "%s: scale %r vs %r" % (what, dev.scale(), info["scale"])
        assert dev.parms_id() == tuple(ref.parms_id(tci)), what

    # 1. rescale_to over three levels, in place and out of place
    cx = d.ct(xs, scale=sc)
    rs = [ref.rescale_to_inplace(o._ct(x, sc), tci) for x in xs]
    dst = S.Ciphertext(d.ctx)
    d.ev.rescale_to(cx, pid, dst)
    same(dst, rs, "rescale_to (out of place)")
    assert cx.coeff_modulus_size() == K
    d.ev.rescale_to_inplace(cx, pid)
    same(cx, rs, "rescale_to_inplace")
    # 2. mod_switch_to over three levels (CKKS drops components), in place and out of place
    cm = d.ct(xs, scale=2.0 ** 30)
    rm = [ref.mod_switch_to_inplace(o._ct(x, 2.0 ** 30), tci) for x in xs]
    dst = S.Ciphertext(d.ctx)
    d.ev.mod_switch_to(cm, pid, dst)
    same(dst, rm, "mod_switch_to (out of place)")
    d.ev.mod_switch_to_inplace(cm, pid)
    same(cm, rm, "mod_switch_to_inplace")
    # 3. right after relinearize / rotate: the first of the three divisions is folded with the pending mod-down
    f0, p0, x0 = S.tail_stats()
    ca, cb = d.ct(xs, scale=2.0 ** 10), d.ct(ys, scale=2.0 ** 10)
    d.ev.multiply_inplace(ca, cb)
    d.ev.relinearize_inplace(ca, d.rlk)
    ca.set_scale(sc)
    d.ev.rescale_to_inplace(ca, pid)
    f1, p1, x1 = S.tail_stats()
    assert (f1 - f0, p1 - p0, x1 - x0) == ((1, 0, 0) if defers else (0, 0, 0)), (f1 - f0, p1 - p0, x1 - x0)
    ra = []
    for b in range(batch):
        a, bb = o._ct(xs[b], 2.0 ** 10), o._ct(ys[b], 2.0 ** 10)
        ref.multiply_inplace(a, bb)
        ref.relinearize_inplace(a)
        a = o._ct(a.data(), sc)
        ra.append(ref.rescale_to_inplace(a, tci))
    same(ca, ra, "relinearize then rescale_to")
    cr = d.ct(xs, scale=sc)
    d.ev.rotate_vector_inplace(cr, step, d.glk)
    d.ev.rescale_to_inplace(cr, pid)
    f2, p2, x2 = S.tail_stats()
    assert (f2 - f1, p2 - p1) == ((1, 0) if defers else (0, 0))
    rr = [ref.rescale_to_inplace(ref.apply_galois_inplace(o._ct(x, sc), elt), tci) for x in xs]
    same(cr, rr, "rotate then rescale_to")
    # ... and mod_switch_to after a key switch: the tail is completed on its own, then the components are dropped
    cs = d.ct(xs, scale=2.0 ** 30)
    d.ev.rotate_vector_inplace(cs, step, d.glk)
    d.ev.mod_switch_to_inplace(cs, pid)
    f3, p3, x3 = S.tail_stats()
    assert (f3 - f2, p3 - p2) == ((0, 1) if defers else (0, 0))
    rsw = [ref.mod_switch_to_inplace(ref.apply_galois_inplace(o._ct(x, 2.0 ** 30), elt), tci) for x in xs]
    same(cs, rsw, "rotate then mod_switch_to")
    # 4. the calls the reference rejects are rejected the same way
    rejected = (
            ("rescale_to a higher level", lambda: d.ev.rescale_to_inplace(cs, d.parms_id_for_K(K)), lambda c: ref.rescale_to_inplace(c, o._ci(K))),
            ("mod_switch_to a higher level", lambda: d.ev.mod_switch_to_inplace(cs, d.parms_id_for_K(K)), lambda c: ref.mod_switch_to_inplace(c, o._ci(K))),
            ("mod_switch_to a level the scale does not fit", lambda: d.ev.mod_switch_to_inplace(cs, d.parms_id_for_K(1)), lambda c: ref.mod_switch_to_inplace(c, o._ci(1))))
    for what, dev_call, ref_call in rejected[:3 if target_K >= 2 else 2]:   # the third needs a level below the current one
        big = 2.0 ** (sum(bits[:target_K]) - 5)     # fits the current level, not the one below
        cs.set_scale(big)
        try:
            ref_call(o._ct(rsw[0].data(), big))
            raise AssertionError("the reference accepts: " + what)
        except sealref.RefError as e:
            assert e.code == 1, what
        try:
            dev_call()
            raise AssertionError("the device accepts: " + what)
        except S.InvalidArgument:
            pass
    cs.set_scale(2.0 ** 30)
    same(cs, rsw, "operand unchanged by the rejected calls")


def case_multi_level_bfv_bgv(scheme, n, primes, t, batch=2, seed=43):
    """mod_switch_to_inplace(Ciphertext) over three levels for BFV (coefficient form, divide_and_round_q_last) and BGV (NTT form,
    correction factor tracked), against the reference's own multi-level call"""
    import sealref
    o = Oracle(scheme, n, primes, t, kind="reference")
    ref = o.ref
    d = DeviceSide(scheme, n, primes, t)
    K = len(primes) - 1
    assert K >= 4
    target_K = K - 3
    rng = np.random.default_rng(seed)
    xs = [rand_ct(rng, primes, K, n) for _ in range(batch)]
    cx = d.ct(xs, is_ntt=scheme == "bgv")
    rs = [ref.mod_switch_to_inplace(o._ct(x), o._ci(target_K)) for x in xs]
    dst = S.Ciphertext(d.ctx)
    d.ev.mod_switch_to(cx, d.parms_id_for_K(target_K), dst)
    d.ev.mod_switch_to_inplace(cx, d.parms_id_for_K(target_K))
    for c, what in ((dst, "out of place"), (cx, "in place")):
        got = d.out(c)
        for b in range(batch):
            _eq(got[b], rs[b].data(), "%s mod_switch_to (%s) item %d" % (scheme, what, b))
        info = rs[0].info()
        assert c.coeff_modulus_size() == target_K == info["coeff_modulus_size"]
        if scheme == "bgv":
            assert c.correction_factor() == info["correction_factor"]
    try:
        d.ev.rescale_to_inplace(cx, d.parms_id_for_K(1))
        raise AssertionError("rescale_to is CKKS only")
    except S.InvalidArgument:
        pass
    try:
        ref.rescale_to_inplace(rs[0], o._ci(1))
        raise AssertionError("rescale_to is CKKS only (reference)")
    except sealref.RefError as e:
        assert e.code == 1


# ---- a pending key-switch tail and two threads that read the same ciphertext at once (settle() runs from const accessors)
def case_deferred_tail_two_readers(n=8192, bits=(50, 40, 40, 60), rounds=6):
    """The reference lets several threads use one ciphertext as an operand at the same time.  Here the operand carries a
    deferred key-switch tail: exactly one of the readers may run it (the counters say so), both must see the completed words."""
    import threading
    primes = coeff_modulus_create(n, list(bits))
    K = len(primes) - 1
    o = Oracle("ckks", n, primes)
    d = DeviceSide("ckks", n, primes)
    d.upload_keys(o)
    rng = np.random.default_rng(17)
    x, y = rand_ct(rng, primes, K, n), rand_ct(rng, primes, K, n)
    z1, z2 = rand_ct(rng, primes, K, n), rand_ct(rng, primes, K, n)
    ref_relin = o.relinearize(o.multiply(x, y))
    qk = np.array(primes[:K], dtype=np.uint64)[None, :, None]
    defers = 13 <= n.bit_length() - 1 <= 16 and not os.environ.get("SEALHIP_KS_EAGER_TAIL")
    ev2 = S.Evaluator(d.ctx)
    for r in range(rounds):
        a, b = d.ct([x], scale=2.0 ** 10), d.ct([y], scale=2.0 ** 10)
        d.ev.multiply_inplace(a, b)
        d.ev.relinearize_inplace(a, d.rlk)            # tail pending on d.ev
        c1, c2 = d.ct([z1], scale=2.0 ** 20), d.ct([z2], scale=2.0 ** 20)
        out1, out2 = S.Ciphertext(d.ctx), S.Ciphertext(d.ctx)
        f0, p0, x0 = S.tail_stats()
        start = threading.Barrier(2)
        errs = []

        def reader(ev, other, out):
            try:
                start.wait()
                ev.add(a, other, out)
            except Exception as e:   # noqa: BLE001 - reported below
                errs.append(e)
        # odd rounds: the same evaluator from both threads; even rounds: two evaluators (two streams)
        t1 = threading.Thread(target=reader, args=(d.ev, c1, out1))
        t2 = threading.Thread(target=reader, args=(d.ev if r & 1 else ev2, c2, out2))
        t1.start(); t2.start(); t1.join(); t2.join()
        assert not errs, errs
        S.device_synchronize()
        f1, p1, x1 = S.tail_stats()
        assert (f1 - f0, p1 - p0, x1 - x0) == ((0, 1, 0) if defers else (0, 0, 0)), "the pending tail must run exactly once"
        _eq(d.out(out1)[0], (ref_relin + z1) % qk, "reader 1, round %d" % r)
        _eq(d.out(out2)[0], (ref_relin + z2) % qk, "reader 2, round %d" % r)
        _eq(d.out(a)[0], ref_relin, "the shared operand, round %d" % r)


def case_pending_product_threads(n=8192, bits=(50, 40, 40, 60), rounds=4):
    """Pending tensor products and host threads (the reference lets several threads use one ciphertext as an operand at the same time):
    (a) two threads read one destination whose product is pending - exactly one of them forms it (SealHip_ProductStats), both see the
    words; (b) two threads, each on its own evaluator / stream, multiply their own operand with ONE shared second operand and relinearise
    (the shared operand's reader list is touched from both), fused on both."""
    import threading
    primes = coeff_modulus_create(n, list(bits))
    K = len(primes) - 1
    o = Oracle("ckks", n, primes)
    d = DeviceSide("ckks", n, primes)
    d.upload_keys(o)
    rng = np.random.default_rng(23)
    x, x2, y = rand_ct(rng, primes, K, n), rand_ct(rng, primes, K, n), rand_ct(rng, primes, K, n)
    z1, z2 = rand_ct(rng, primes, K, n, size=3), rand_ct(rng, primes, K, n, size=3)
    ref_prod, ref_prod2 = o.multiply(x, y), o.multiply(x2, y)
    qk = np.array(primes[:K], dtype=np.uint64)[None, :, None]
    defers = 13 <= n.bit_length() - 1 <= 16 and not os.environ.get("SEALHIP_KS_EAGER_TAIL")
    ev2 = S.Evaluator(d.ctx)
    with _Env(SEALHIP_KS_SPLIT=1, SEALHIP_LAZY_PRODUCT_MIN_WGS=0, SEALHIP_LAZY_PRODUCT=None):
        for r in range(rounds):
            # (a)
            a, b, w = d.ct([x], scale=2.0 ** 10), d.ct([y], scale=2.0 ** 10), S.Ciphertext(d.ctx)
            if r & 1:
                d.ev.multiply(a, b, w)
            else:
                d.ev.multiply_inplace(a, b)
                w = a
            c1, c2 = d.ct([z1], scale=2.0 ** 20), d.ct([z2], scale=2.0 ** 20)
            out1, out2 = S.Ciphertext(d.ctx), S.Ciphertext(d.ctx)
            f0, m0, x0 = S.product_stats()
            start = threading.Barrier(2)
            errs = []

            def reader(ev, other, out):
                try:
                    start.wait()
                    ev.add(w, other, out)
                except Exception as e:   # noqa: BLE001 - reported below
                    errs.append(e)
            t1 = threading.Thread(target=reader, args=(d.ev, c1, out1))
            t2 = threading.Thread(target=reader, args=(d.ev if r & 2 else ev2, c2, out2))
            t1.start(); t2.start(); t1.join(); t2.join()
            assert not errs, errs
            S.device_synchronize()
            f1, m1, x1 = S.product_stats()
            assert (f1 - f0, m1 - m0, x1 - x0) == ((0, 1, 0) if defers else (0, 0, 0)), "the pending product must be formed exactly once"
            _eq(d.out(out1)[0], (ref_prod + z1) % qk, "reader 1, round %d" % r)
            _eq(d.out(out2)[0], (ref_prod + z2) % qk, "reader 2, round %d" % r)
            _eq(d.out(w)[0], ref_prod, "the shared destination, round %d" % r)
            # (b)
            a1, a2, b = d.ct([x], scale=2.0 ** 10), d.ct([x2], scale=2.0 ** 10), d.ct([y], scale=2.0 ** 10)
            w1, w2 = S.Ciphertext(d.ctx), S.Ciphertext(d.ctx)
            f0, m0, x0 = S.product_stats()
            start = threading.Barrier(2)

            def worker(ev, xa, wa):
                try:
                    start.wait()
                    ev.multiply(xa, b, wa)
                    ev.relinearize_inplace(wa, d.rlk)
                except Exception as e:   # noqa: BLE001
                    errs.append(e)
            t1 = threading.Thread(target=worker, args=(d.ev, a1, w1))
            t2 = threading.Thread(target=worker, args=(ev2, a2, w2))
            t1.start(); t2.start(); t1.join(); t2.join()
            assert not errs, errs
            S.device_synchronize()
            f1, m1, x1 = S.product_stats()
            assert (f1 - f0, m1 - m0) == ((2, 0) if defers else (0, 0)), "both products fused: %r" % ((f1 - f0, m1 - m0),)
            _eq(d.out(w1)[0], o.relinearize(ref_prod), "worker 1, round %d" % r)
            _eq(d.out(w2)[0], o.relinearize(ref_prod2), "worker 2, round %d" % r)
            _eq(d.out(b)[0], y, "the shared operand, round %d" % r)
    del ev2


# ---- the 2 x 2 tensor product grows its first operand in two ways (evaluator.cpp: tensor_2x2)
def case_product_growth(scheme, n, bits, tbits=20, batch=3, seed=51):
    """multiply_inplace / square_inplace of a size-2 ciphertext: into a new slab when the operand's slab holds two polynomials
    (a fresh ciphertext), in place when it has room for three (a ciphertext that was relinearized before) - word for word the
    reference's product both times, for distinct operands and for the square (evaluator.cpp:626-707, 878-1142)"""
    primes = coeff_modulus_create(n, list(bits))
    t = plain_modulus_batching(n, tbits) if scheme != "ckks" else 0
    K = len(primes) - 1
    o = Oracle(scheme, n, primes, t)
    d = DeviceSide(scheme, n, primes, t)
    d.upload_keys(o)
    rng = np.random.default_rng(seed)
    ntt = scheme != "bfv"
    scale = 2.0 ** 10 if scheme == "ckks" else 1.0
    xs = [rand_ct(rng, primes, K, n) for _ in range(batch)]
    ys = [rand_ct(rng, primes, K, n) for _ in range(batch)]
    for square in (False, True):
        a = d.ct(xs, scale=scale, is_ntt=ntt)
        if square:
            d.ev.square_inplace(a)                                    # fresh slab of two polynomials: new slab
            want = [o.multiply(x, x) for x in xs]
        else:
            d.ev.multiply_inplace(a, d.ct(ys, scale=scale, is_ntt=ntt))
            want = [o.multiply(x, y) for x, y in zip(xs, ys)]
        got = DeviceSide.out(a)
        for b in range(batch):
            _eq(got[b], want[b], "%s product into a new slab (square=%s) item %d" % (scheme, square, b))
        d.ev.relinearize_inplace(a, d.rlk)                             # back to two polynomials, the slab keeps room for three
        if scheme == "ckks":
            a.set_scale(scale)
        two = [o.relinearize(w) for w in want]
        if square:
            d.ev.square_inplace(a)
            want2 = [o.multiply(w, w) for w in two]
        else:
            d.ev.multiply_inplace(a, d.ct(ys, scale=scale, is_ntt=ntt))
            want2 = [o.multiply(w, y) for w, y in zip(two, ys)]
        got = DeviceSide.out(a)
        for b in range(batch):
            _eq(got[b], want2[b], "%s product in place (square=%s) item %d" % (scheme, square, b))


# ---- structured worst-case inputs.  Every other case draws uniform residues; the reference's own unit tests use zeros,
#      ones and q - 1 (native/tests/seal/util/ntt.cpp:53-133, polyarithsmallmod.cpp:545-641, rns.cpp:904-1073), and the lazy-reduction
#      bounds of both back ends (field.h: fix() placements, IntBounds) are worst-case only on structured data.
EXTREME_PATTERNS = ("qm1", "zero", "alt", "half", "half1", "spike", "mix4", "one")


def extreme_slab(primes, n, pattern, seed=0):
    """[len(primes)][n] words: all q-1 / all 0 / alternating 0, q-1 / all floor(q/2) / all floor(q/2)+1 / a single q-1 at N-1 /
    a random mix of {0, q-1, floor(q/2), floor(q/2)+1} / all 1"""
    out = np.zeros((len(primes), n), dtype=np.uint64)
    for i, q in enumerate(primes):
        q = int(q)
        if pattern == "qm1":
            out[i, :] = q - 1
        elif pattern == "zero":
            pass
        elif pattern == "alt":
            out[i, 1::2] = q - 1
        elif pattern == "half":
            out[i, :] = q // 2
        elif pattern == "half1":
            out[i, :] = q // 2 + 1
        elif pattern == "spike":
            out[i, n - 1] = q - 1
        elif pattern == "mix4":
            vals = np.array([0, q - 1, q // 2, q // 2 + 1], dtype=np.uint64)
            out[i, :] = vals[np.random.default_rng([seed, i]).integers(0, 4, n)]
        elif pattern == "one":
            out[i, :] = 1
        else:
            raise ValueError(pattern)
    return out


def case_extremes_ntt(n, bits, polys):
    """ntt_negacyclic_harvey[_lazy] / inverse (ntt.cpp:394-475) on the extreme slabs; polynomial p holds pattern p mod 8, so a large
    `polys` runs them through the looping workgroups (and, at 2^16, the packed intermediate)"""
    primes = coeff_modulus_create(n, bits)
    o = Oracle("ckks", n, primes)
    d = DeviceSide("ckks", n, primes)
    L = len(primes)
    uniq = [extreme_slab(primes, n, pat, seed=7) for pat in EXTREME_PATTERNS]
    U = len(uniq)
    x = np.stack([uniq[p % U] for p in range(polys)])
    q = np.array(primes, dtype=np.uint64)[None, :, None]
    fwd_exp = [o.ntt(0, u, "fwd") for u in uniq]
    inv_exp = [o.ntt(0, u, "inv") for u in uniq]
    for lazy in (False, True):
        buf = S.DeviceBuffer.from_numpy(x)
        S.ntt_forward(d.ctx, buf, polys, L, lazy=lazy)
        got = buf.to_numpy(x.shape)
        if lazy:
            assert (got < 4 * q).all()
            got = got % q
        for p in range(polys):
            _eq(got[p], fwd_exp[p % U], "forward%s of pattern %s (poly %d)" % (" lazy" if lazy else "", EXTREME_PATTERNS[p % U], p))
        buf = S.DeviceBuffer.from_numpy(x)
        S.ntt_inverse(d.ctx, buf, polys, L, lazy=lazy)
        got = buf.to_numpy(x.shape)
        if lazy:
            assert (got < 2 * q).all()
            got = got % q
        for p in range(polys):
            _eq(got[p], inv_exp[p % U], "inverse%s of pattern %s (poly %d)" % (" lazy" if lazy else "", EXTREME_PATTERNS[p % U], p))
    # and the round trip of the forward results (values spread over the whole range after one transform of a flat input)
    buf = S.DeviceBuffer.from_numpy(np.stack([fwd_exp[p % U] for p in range(polys)]))
    S.ntt_inverse(d.ctx, buf, polys, L)
    _eq(buf.to_numpy(x.shape), x, "inverse(forward(pattern)) == pattern")


def _extreme_items(primes, K, n):
    """eight size-2 ciphertexts [2][K][n]: item i = (pattern i, pattern i + 1); a zero second polynomial would make products transparent
    (the reference throws), so it is replaced by the all-ones pattern"""
    U = len(EXTREME_PATTERNS)
    items = []
    for i in range(U):
        p0, p1 = EXTREME_PATTERNS[i], EXTREME_PATTERNS[(i + 1) % U]
        if p1 == "zero":
            p1 = "one"
        items.append(np.stack([extreme_slab(primes[:K], n, p0, seed=11 + i), extreme_slab(primes[:K], n, p1, seed=29 + i)]))
    return items


def case_extremes_ckks(n, bits, batch=8, check=None):
    """multiply -> relinearize -> rescale and rotate -> rescale (evaluator.cpp:569, 1144, 1503, 2504) on the extreme slabs; with
    batch > 8 the eight items are tiled over the batch (chunked key switch, looping workgroups) and `check` items are compared"""
    primes = coeff_modulus_create(n, bits)
    K = len(primes) - 1
    probe = Oracle("ckks", n, primes)
    elt = probe.galois_elt_from_step(1)
    o = Oracle("ckks", n, primes, galois_elts=[elt])
    d = DeviceSide("ckks", n, primes)
    d.upload_keys(o)
    pid = d.parms_id_for_K(K)
    xs = _extreme_items(primes, K, n)
    U = len(xs)
    ys = [xs[(i + 3) % U] for i in range(U)]
    assert batch % U == 0
    hx = np.stack([np.stack([xs[i][p] for i in range(U)]) for p in range(2)])  # [polys][unique][K][n]
    hy = np.stack([np.stack([ys[i][p] for i in range(U)]) for p in range(2)])
    cx = _tiled_ciphertext(d.ctx, hx, batch, pid, 2.0 ** 10)
    cy = _tiled_ciphertext(d.ctx, hy, batch, pid, 2.0 ** 10)
    check = list(range(U)) if check is None else list(check)
    # multiply -> relinearize -> rescale
    work = S.Ciphertext(d.ctx, batch=batch)
    d.ev.multiply(cx, cy, work)
    prod = {b: work.item_to_numpy(b) for b in check}
    d.ev.relinearize_inplace(work, d.rlk)
    relin = {b: work.item_to_numpy(b) for b in check}
    work.set_scale(float(primes[K - 1]) * 2.0 ** 10)
    d.ev.rescale_to_next_inplace(work)
    exp_cache = {}
    for b in check:
        u = b % U
        if u not in exp_cache:
            e0 = o.multiply(xs[u], ys[u])
            e1 = o.relinearize(e0)
            exp_cache[u] = (e0, e1, o.rescale(e1))
        e0, e1, e2 = exp_cache[u]
        _eq(prod[b], e0, "multiply of extreme item %d (%s)" % (b, EXTREME_PATTERNS[u]))
        _eq(relin[b], e1, "relinearize of extreme item %d (%s)" % (b, EXTREME_PATTERNS[u]))
        _eq(work.item_to_numpy(b), e2, "rescale of extreme item %d (%s)" % (b, EXTREME_PATTERNS[u]))
    # rotate -> rescale on the inputs themselves
    d.ev.rotate_vector_inplace(cx, 1, d.glk)
    rot = {b: cx.item_to_numpy(b) for b in check}
    cx.set_scale(float(primes[K - 1]) * 2.0 ** 10)
    d.ev.rescale_to_next_inplace(cx)
    rot_cache = {}
    for b in check:
        u = b % U
        if u not in rot_cache:
            r0 = o.apply_galois(xs[u], elt)
            rot_cache[u] = (r0, o.rescale(r0))
        _eq(rot[b], rot_cache[u][0], "rotate_vector of extreme item %d (%s)" % (b, EXTREME_PATTERNS[u]))
        _eq(cx.item_to_numpy(b), rot_cache[u][1], "rotate + rescale of extreme item %d (%s)" % (b, EXTREME_PATTERNS[u]))
    del cx, cy, work
    S.release_pool()


def case_extremes_bfv(n, primes, t):
    """bfv_multiply -> relinearize -> mod_switch_to_next (evaluator.cpp:395, 1144, 1404; rns.cpp:789-1131) on the extreme slabs: the
    BEHZ base conversions see residues at the ends and the middle of every range"""
    K = len(primes) - 1
    o = Oracle("bfv", n, primes, t)
    d = DeviceSide("bfv", n, primes, t)
    d.upload_keys(o)
    xs = _extreme_items(primes, K, n)
    U = len(xs)
    ys = [xs[(i + 3) % U] for i in range(U)]
    cx, cy = d.ct(xs), d.ct(ys)
    d.ev.multiply_inplace(cx, cy)
    cur = d.out(cx)
    exp = [o.multiply(xs[b], ys[b]) for b in range(U)]
    for b in range(U):
        _eq(cur[b], exp[b], "bfv multiply of extreme item %d (%s)" % (b, EXTREME_PATTERNS[b]))
    d.ev.relinearize_inplace(cx, d.rlk)
    cur = d.out(cx)
    exp = [o.relinearize(e) for e in exp]
    for b in range(U):
        _eq(cur[b], exp[b], "bfv relinearize of extreme item %d (%s)" % (b, EXTREME_PATTERNS[b]))
    if K >= 2:
        d.ev.mod_switch_to_next_inplace(cx)
        cur = d.out(cx)
        for b in range(U):
            _eq(cur[b], o.mod_switch_to_next(exp[b]), "bfv mod_switch_to_next of extreme item %d (%s)" % (b, EXTREME_PATTERNS[b]))
    # squares of the inputs (bfv_square, evaluator.cpp:878)
    cz = d.ct(xs)
    d.ev.square_inplace(cz)
    cur = d.out(cz)
    for b in range(U):
        _eq(cur[b], o.multiply(xs[b], xs[b]), "bfv square of extreme item %d (%s)" % (b, EXTREME_PATTERNS[b]))


# ---- deferred tensor products (sealhip.h: SealHip_ProductStats; evaluator.h: LazyProduct): Evaluator::multiply into a third object does
#      not form the product at once; a relinearize that follows forms it inside its kernels.  Every way the caller can observe the
#      words - of the destination or of the operands - must give the reference's words (evaluator.cpp:569-708, 1144-1199).
def case_lazy_product(n, bits, batch=3, seed=61):
    primes = coeff_modulus_create(n, bits)
    K = len(primes) - 1
    o = Oracle("ckks", n, primes)
    d = DeviceSide("ckks", n, primes)
    d.upload_keys(o)
    rng = np.random.default_rng(seed)
    xs = [rand_ct(rng, primes, K, n) for _ in range(batch)]
    ys = [rand_ct(rng, primes, K, n) for _ in range(batch)]
    zs = [rand_ct(rng, primes, K, n) for _ in range(batch)]
    qk = np.array(primes[:K], dtype=np.uint64)[None, :, None]
    sc = float(primes[K - 1]) * 2.0 ** 10
    prod = [o.multiply(xs[b], ys[b]) for b in range(batch)]
    relin = [o.relinearize(p) for p in prod]
    resc = [o.rescale(r) for r in relin]
    import gc
    # the launcher's rule keeps small batches eager; the test reaches the deferred path with the two knobs it documents
    with _Env(SEALHIP_KS_SPLIT=1, SEALHIP_LAZY_PRODUCT_MIN_WGS=0, SEALHIP_LAZY_PRODUCT=None):
        defers = 13 <= n.bit_length() - 1 <= 16 and not os.environ.get("SEALHIP_KS_EAGER_TAIL")

        def fresh():
            return d.ct(xs, scale=2.0 ** 10), d.ct(ys, scale=2.0 ** 10), S.Ciphertext(d.ctx, batch=batch)

        def check3(w, what):
            got = d.out(w)
            for b in range(batch):
                _eq(got[b], prod[b], what + ", item %d" % b)

        def check_relin(w, what):
            got = d.out(w)
            for b in range(batch):
                _eq(got[b], relin[b], what + ", item %d" % b)

        # 1. the fused path: multiply -> relinearize -> rescale with nothing in between
        f0, m0, x0 = S.product_stats()
        cx, cy, w = fresh()
        d.ev.multiply(cx, cy, w)
        assert w.size() == 3 and w.is_ntt_form() and w.scale() == 2.0 ** 20
        d.ev.relinearize_inplace(w, d.rlk)
        assert w.size() == 2
        w.set_scale(sc)
        d.ev.rescale_to_next_inplace(w)
        f1, m1, x1 = S.product_stats()
        assert (f1 - f0, m1 - m0) == ((1, 0) if defers else (0, 0)), "the fused relinearisation did not run where it should: %r" % ((f1 - f0, m1 - m0),)
        got = d.out(w)
        for b in range(batch):
            _eq(got[b], resc[b], "multiply + relinearize (fused) + rescale, item %d" % b)
        kept_x, kept_y = d.out(cx), d.out(cy)
        for b in range(batch):
            _eq(kept_x[b], xs[b], "operand x after the fused path, item %d" % b)
            _eq(kept_y[b], ys[b], "operand y after the fused path, item %d" % b)
        # ... and relinearize alone (the tail completed by the read)
        cx, cy, w = fresh()
        d.ev.multiply(cx, cy, w)
        d.ev.relinearize_inplace(w, d.rlk)
        check_relin(w, "multiply + relinearize (fused), read")

        # 2. the product is read before anything else: it is formed then
        cx, cy, w = fresh()
        d.ev.multiply(cx, cy, w)
        check3(w, "a deferred product read at once")
        d.ev.relinearize_inplace(w, d.rlk)
        check_relin(w, "relinearize after the product was read")

        # 3. an operand is written while the product is pending: the product is of the OLD words
        cx, cy, w = fresh()
        cz = d.ct(zs, scale=2.0 ** 10)
        d.ev.multiply(cx, cy, w)
        d.ev.add_inplace(cx, cz)
        d.ev.negate_inplace(cy)
        d.ev.relinearize_inplace(w, d.rlk)
        check_relin(w, "relinearize after both operands were overwritten")
        got = d.out(cx)
        for b in range(batch):
            _eq(got[b], (xs[b] + zs[b]) % qk, "the overwritten operand, item %d" % b)

        # 4. an operand is destroyed / overwritten as a whole / re-shaped while the product is pending
        cx, cy, w = fresh()
        d.ev.multiply(cx, cy, w)
        del cy
        gc.collect()
        d.ev.relinearize_inplace(w, d.rlk)
        check_relin(w, "relinearize after an operand was destroyed")
        cx, cy, w = fresh()
        d.ev.multiply(cx, cy, w)
        d.ev.multiply(cy, cy, cx)          # cx is now a destination (its own product pending), w's product must be of the old cx
        d.ev.relinearize_inplace(w, d.rlk)
        check_relin(w, "relinearize after an operand became a destination")
        got = d.out(cx)
        for b in range(batch):
            _eq(got[b], o.multiply(ys[b], ys[b]), "the operand that became a product, item %d" % b)
        cx, cy, w = fresh()
        d.ev.multiply(cx, cy, w)
        cy.reserve(4)
        cx.resize(d.parms_id_for_K(K), 3)
        d.ev.relinearize_inplace(w, d.rlk)
        check_relin(w, "relinearize after the operands were re-shaped")

        # 5. the same operand twice; a pending product as an operand; a copy of a pending product
        cx, cy, w = fresh()
        d.ev.multiply(cx, cx, w)
        d.ev.relinearize_inplace(w, d.rlk)
        got = d.out(w)
        for b in range(batch):
            _eq(got[b], o.relinearize(o.multiply(xs[b], xs[b])), "x times x, fused, item %d" % b)
        cx, cy, w = fresh()
        d.ev.multiply(cx, cy, w)
        w2 = w.copy()
        check3(w2, "copy of a pending product")
        d.ev.relinearize_inplace(w, d.rlk)
        check_relin(w, "the original after it was copied")
        cx, cy, w = fresh()
        d.ev.multiply(cx, cy, w)
        d.ev.add_inplace(w, w2)            # a pending product as the in-place operand of something else
        got = d.out(w)
        for b in range(batch):
            _eq(got[b], (prod[b] + prod[b]) % qk, "add onto a pending product, item %d" % b)

        # 6. the destination is overwritten before anyone needs it: the product is never formed
        f2, m2, x2 = S.product_stats()
        cx, cy, w = fresh()
        d.ev.multiply(cx, cy, w)
        d.ev.multiply(cy, cy, w)
        d.ev.relinearize_inplace(w, d.rlk)
        got = d.out(w)
        for b in range(batch):
            _eq(got[b], o.relinearize(o.multiply(ys[b], ys[b])), "the second product into one destination, item %d" % b)
        cx, cy, w = fresh()
        d.ev.multiply(cx, cy, w)
        del w
        gc.collect()
        f3, m3, x3 = S.product_stats()
        assert x3 - x2 == (2 if defers else 0) and m3 == m2, "discarded products: %r" % ((f3 - f2, m3 - m2, x3 - x2),)

        # 7. an operand with a pending key-switch tail of its own: the tail completes first, then the product defers
        ca, cb, w = fresh()
        d.ev.multiply_inplace(ca, cb)
        d.ev.relinearize_inplace(ca, d.rlk)      # ca: deferred tail
        cz = d.ct(zs, scale=2.0 ** 20)
        d.ev.multiply(ca, cz, w)
        d.ev.relinearize_inplace(w, d.rlk)
        got = d.out(w)
        for b in range(batch):
            _eq(got[b], o.relinearize(o.multiply(relin[b], zs[b])), "product of a relinearised operand, fused, item %d" % b)

        # 8. another evaluator reads / writes while the first one's product is pending
        ev2 = S.Evaluator(d.ctx)
        cx, cy, w = fresh()
        d.ev.multiply(cx, cy, w)
        ev2.negate_inplace(cx)
        ev2.relinearize_inplace(w, d.rlk)        # not the owner: the product is formed, the ordinary path runs
        check_relin(w, "relinearize on another evaluator")
        del ev2
        # 10. the in-place forms (multiply_inplace, square_inplace): the destination's previous slab is the operand, kept by the record
        sq = [o.multiply(xs[b], xs[b]) for b in range(batch)]
        f5, m5, x5 = S.product_stats()
        cx, cy, _ = fresh()
        d.ev.multiply_inplace(cx, cy)
        assert cx.size() == 3 and cx.scale() == 2.0 ** 20
        d.ev.relinearize_inplace(cx, d.rlk)
        cx.set_scale(sc)
        d.ev.rescale_to_next_inplace(cx)
        got = d.out(cx)
        for b in range(batch):
            _eq(got[b], resc[b], "multiply_inplace + relinearize (fused) + rescale, item %d" % b)
        got = d.out(cy)
        for b in range(batch):
            _eq(got[b], ys[b], "the second operand after the fused in-place path, item %d" % b)
        cx, cy, _ = fresh()
        d.ev.square_inplace(cx)
        d.ev.relinearize_inplace(cx, d.rlk)
        got = d.out(cx)
        for b in range(batch):
            _eq(got[b], o.relinearize(sq[b]), "square_inplace + relinearize (fused), item %d" % b)
        f6, m6, x6 = S.product_stats()
        assert (f6 - f5, m6 - m5) == ((2, 0) if defers else (0, 0)), "in-place products fused / formed: %r" % ((f6 - f5, m6 - m5),)
        # read at once; the second operand written / destroyed while pending; the destination an operand of something else while pending
        cx, cy, _ = fresh()
        d.ev.multiply_inplace(cx, cy)
        check3(cx, "an in-place product read at once")
        cx, cy, _ = fresh()
        d.ev.multiply_inplace(cx, cy)
        d.ev.negate_inplace(cy)
        d.ev.relinearize_inplace(cx, d.rlk)
        check_relin(cx, "in-place product, second operand written while pending")
        cx, cy, _ = fresh()
        d.ev.multiply_inplace(cx, cy)
        del cy
        gc.collect()
        d.ev.relinearize_inplace(cx, d.rlk)
        check_relin(cx, "in-place product, second operand destroyed while pending")
        cx, cy, w = fresh()
        d.ev.multiply_inplace(cx, cy)
        d.ev.add(cx, cx, w)                 # a pending in-place product read as an operand of another operation
        got = d.out(w)
        for b in range(batch):
            _eq(got[b], (prod[b] + prod[b]) % qk, "sum of a pending in-place product with itself, item %d" % b)
        # a pending in-place product as the OPERAND of a three-object product that is pending too, then both consumed
        cx, cy, w = fresh()
        cz = d.ct(zs, scale=2.0 ** 10)
        d.ev.multiply_inplace(cx, cy)
        c2 = cx.copy()                      # (forms it: the copy needs the words)
        check3(c2, "copy of a pending in-place product")
        d.ev.relinearize_inplace(cx, d.rlk)
        check_relin(cx, "the in-place original after it was copied")
        # discarded: overwritten as a whole / destroyed before anyone needs the words (the kept slab goes back to the pool)
        f7, m7, x7 = S.product_stats()
        cx, cy, _ = fresh()
        d.ev.multiply_inplace(cx, cy)
        d.ev.multiply(cy, cz, cx)
        d.ev.relinearize_inplace(cx, d.rlk)
        got = d.out(cx)
        for b in range(batch):
            _eq(got[b], o.relinearize(o.multiply(ys[b], zs[b])), "a three-object product over a pending in-place one, item %d" % b)
        cx, cy, _ = fresh()
        d.ev.square_inplace(cx)
        del cx
        gc.collect()
        f8, m8, x8 = S.product_stats()
        assert x8 - x7 == (2 if defers else 0) and m8 == m7, "discarded in-place products: %r" % ((f8 - f7, m8 - m7, x8 - x7),)
        # another evaluator relinearises: the product is formed, the ordinary path runs
        ev3 = S.Evaluator(d.ctx)
        cx, cy, _ = fresh()
        d.ev.multiply_inplace(cx, cy)
        ev3.relinearize_inplace(cx, d.rlk)
        check_relin(cx, "in-place product relinearised on another evaluator")
        del ev3
    # 9. and with the deferral switched off nothing is pending, same words
    with _Env(SEALHIP_LAZY_PRODUCT=0, SEALHIP_KS_SPLIT=1, SEALHIP_LAZY_PRODUCT_MIN_WGS=0):
        f4, m4, x4 = S.product_stats()
        cx, cy, w = d.ct(xs, scale=2.0 ** 10), d.ct(ys, scale=2.0 ** 10), S.Ciphertext(d.ctx, batch=batch)
        d.ev.multiply(cx, cy, w)
        d.ev.relinearize_inplace(w, d.rlk)
        got = d.out(w)
        for b in range(batch):
            _eq(got[b], relin[b], "SEALHIP_LAZY_PRODUCT=0, item %d" % b)
        assert S.product_stats() == (f4, m4, x4)


def case_rotate_gather(n, bits, batch=3, steps=(1, -1), seed=71):
    """Round 6: rotations at the batches whose key switch runs un-split with the addend folded in read c0 and c1 through the
    automorphism's index map inside the key switch's kernels (evaluator_levels.cpp: apply_galois; no permutation kernels, the operand's
    slab kept alive in the in-place form).  Out of place, in place, conjugation, each followed by a rescale (folded tail) or read at once
    (plain tail), against the reference; the switch SEALHIP_KS_SPLIT=1 brings small test batches onto that path."""
    primes = coeff_modulus_create(n, bits)
    K = len(primes) - 1
    probe = Oracle("ckks", n, primes)
    elts = sorted({probe.galois_elt_from_step(s) for s in steps} | {2 * n - 1})
    o = Oracle("ckks", n, primes, galois_elts=elts)
    d = DeviceSide("ckks", n, primes)
    d.upload_keys(o)
    rng = np.random.default_rng(seed)
    xs = [rand_ct(rng, primes, K, n) for _ in range(batch)]
    sc = float(primes[K - 1]) * 2.0 ** 10
    g0 = S.galois_stats()
    with _Env(SEALHIP_KS_SPLIT=1):
        for s in steps:
            e = o.galois_elt_from_step(s)
            want = [o.apply_galois(xs[b], e) for b in range(batch)]
            # out of place, read at once: the operand keeps its words
            cx = d.ct(xs, scale=sc)
            dst = S.Ciphertext(d.ctx, batch=batch)
            d.ev.rotate_vector(cx, s, d.glk, dst)
            got, kept = d.out(dst), d.out(cx)
            for b in range(batch):
                _eq(got[b], want[b], "rotate_vector(%d) out of place, item %d" % (s, b))
                _eq(kept[b], xs[b], "operand of the out-of-place rotation, item %d" % b)
            # out of place + rescale (folded tail), the operand destroyed right after the call
            cx = d.ct(xs, scale=sc)
            dst = S.Ciphertext(d.ctx, batch=batch)
            d.ev.rotate_vector(cx, s, d.glk, dst)
            del cx
            d.ev.rescale_to_next_inplace(dst)
            got = d.out(dst)
            for b in range(batch):
                _eq(got[b], o.rescale(want[b]), "rotate_vector(%d) + rescale, item %d" % (s, b))
            # in place twice (the second rotation reads a ciphertext whose tail is pending), then rescale
            cx = d.ct(xs, scale=sc)
            d.ev.rotate_vector_inplace(cx, s, d.glk)
            d.ev.rotate_vector_inplace(cx, s, d.glk)
            d.ev.rescale_to_next_inplace(cx)
            got = d.out(cx)
            for b in range(batch):
                _eq(got[b], o.rescale(o.apply_galois(want[b], e)), "rotate_vector(%d) in place twice + rescale, item %d" % (s, b))
        cx = d.ct(xs, scale=sc)
        d.ev.complex_conjugate_inplace(cx, d.glk)
        got = d.out(cx)
        for b in range(batch):
            _eq(got[b], o.apply_galois(xs[b], 2 * n - 1), "complex_conjugate in place, item %d" % b)
        # a product relinearised, rotated and rescaled: pending product -> pending tail -> gather -> folded tail
        ys = [rand_ct(rng, primes, K, n) for _ in range(batch)]
        cx, cy = d.ct(xs, scale=2.0 ** 10), d.ct(ys, scale=2.0 ** 10)
        d.ev.multiply_inplace(cx, cy)
        d.ev.relinearize_inplace(cx, d.rlk)
        d.ev.rotate_vector_inplace(cx, steps[0], d.glk)
        cx.set_scale(sc)
        d.ev.rescale_to_next_inplace(cx)
        got = d.out(cx)
        e0 = o.galois_elt_from_step(steps[0])
        for b in range(batch):
            _eq(got[b], o.rescale(o.apply_galois(o.relinearize(o.multiply(xs[b], ys[b])), e0)), "multiply + relinearize + rotate + rescale, item %d" % b)
    g1 = S.galois_stats()
    two_pass = 13 <= n.bit_length() - 1 <= 16 and not os.environ.get("SEALHIP_KS_EAGER_TAIL")
    assert (g1[0] - g0[0], g1[1] - g0[1]) == ((4 * len(steps) + 2, 0) if two_pass else (0, 4 * len(steps) + 2)), "rotation paths: %r" % ((g1[0] - g0[0], g1[1] - g0[1]),)
