"""CPU: the product's host logic (context tables, orchestration, metadata) and the kernels' index
arithmetic, exercised by running the SAME kernel sources under the fiber emulator (tests/hipemu) at
small sizes and checking them against the oracle.  This proves nothing about the GPU build — the
`-m gpu` tests do that — but it lets the host side and every index formula be tested without a GPU."""
import os

import numpy as np
import pytest

import parity_cases as P
from oracle import coeff_modulus_create, plain_modulus_batching


@pytest.mark.parametrize("n,bits", [
    (2, [30, 30]), (8, [30, 30]), (32, [30, 40]),           # small-kernel path (native/tests sizes)
    (64, [40, 50]), (128, [40, 50]), (256, [50, 60]),        # single pass, 2 phases
    (512, [50]), (1024, [50]), (2048, [60]), (4096, [50]),  # single pass, 2-3 phases
    (8192, [60]),                                            # single pass, 4 phases
    (16384, [50]), (32768, [55]), (65536, [60]),            # column pass + row pass
])
def test_ntt_all_plans(emu, n, bits):
    P.case_ntt(n, bits, polys=2 if n <= 4096 else 1)


# two-pass engine (ntt2_kernels.hip): every D1 geometry, both arithmetic back ends in one context
@pytest.mark.parametrize("n,bits", [
    (8192, [50, 30, 60]), (8192, [20, 25]), (16384, [50, 50, 45, 60]), (32768, [40, 50]), (65536, [50, 60, 36]),
    # integer back end, unguarded butterflies (field.h): 51 .. 60-bit user moduli (the 61-bit BEHZ base of the BFV cases is guarded)
    (8192, [60, 59, 58]), (16384, [58, 60]), (32768, [60, 55]), (65536, [60, 59, 51]),
])
def test_ntt_two_pass_engine_mixed_primes(emu, n, bits):
    P.case_ntt(n, bits, polys=2)


# single-launch kernels (ntt2_fwd_fused2 / ntt2_inv_fused2): the per-workgroup loop with the next transform in flight
@pytest.mark.parametrize("n,bits,polys,chunks", [(8192, [50, 36, 60], 5, 2), (8192, [40], 3, 1), (16384, [50, 45], 3, 1), (16384, [60, 50], 4, 3),
                                                 # integer back end in one launch (round 3): the three modulus classes' bodies
                                                 (8192, [60, 59, 57, 55], 5, 2), (16384, [58, 60, 51], 3, 1), (16384, [59, 40, 57], 4, 3)])
def test_ntt_single_launch_loop(emu, monkeypatch, n, bits, polys, chunks):
    monkeypatch.setenv("SEALHIP_NTT_FCHUNKS", str(chunks))
    P.case_ntt(n, bits, polys=polys)


def test_ntt_two_pass_engine_integer_only(emu, monkeypatch):
    """SEALHIP_NO_FP=1: the same primes on the 64-bit integer back end give the same words."""
    monkeypatch.setenv("SEALHIP_NO_FP", "1")
    P.case_ntt(8192, [50, 30, 60], polys=1)


def test_pipelines_integer_only_in_a_subprocess():
    """SEALHIP_NO_FP=1 (read when the context is built): every prime on the integer back end, so the key switch runs its Shoup-key
    sums for primes of 30 to 60 bits (both modulus classes, `hi32` estimates on and off) - CKKS and BFV pipelines, bit-exact"""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import seal_amd as S; S.load(%r); import parity_cases as P\n"
            "from oracle import coeff_modulus_create, plain_modulus_batching\n"
            "P.case_ckks_pipeline(8192, [50, 30, 40, 60], batch=1, steps=(1,))\n"
            "P.case_ckks_pipeline(8192, [36, 30, 33, 40], batch=2, steps=(-1,))\n"
            "P.case_bfv_pipeline(8192, coeff_modulus_create(8192, [50, 55, 56]), plain_modulus_batching(8192, 20), batch=1)\n"
            "print('integer-only ok')\n" % (here, os.path.dirname(here), os.path.join(here, "hipemu", "libsealhip_emu.so")))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SEALHIP_NO_FP="1"), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "integer-only ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_tails_without_the_addend_in_a_subprocess():
    """SEALHIP_KS_NO_FOLD=1 (development switch, read once; the emulated library is built with -DSEALHIP_AB_SWITCHES): the key switch
    leaves the bare sums and both tails add the ciphertext's words themselves - the round-3 form the A/B of
    profiles/r04_fold_addend.txt compares with, still the same words"""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import seal_amd as S; S.load(%r); import parity_cases as P\n"
            "P.case_ckks_pipeline(8192, [50, 30, 40, 60], batch=1, steps=(1,), check_transforms=False)\n"
            "P.case_deferred_tail_lifecycle()\n"
            "print('bare sums ok')\n" % (here, os.path.dirname(here), os.path.join(here, "hipemu", "libsealhip_emu.so")))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SEALHIP_KS_NO_FOLD="1", SEALHIP_KS_TRACE="1"),
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "bare sums ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
    assert "[ks] folded tail\n" in out.stderr and "addend in the sums" not in out.stderr, out.stderr[-1000:]


def test_key_switch_digit_resident_order_in_a_subprocess():
    """SEALHIP_KS1_ORDER=1 (development switch, read once): pass 1 of the fused key switch with the digit's tile resident and the
    targets in the loop (ks1t_kernel) - the order large batches take on the device (profiles/r04_ks1_order.txt) - at sizes the
    emulator finishes: CKKS (both arithmetic classes, every two-pass size, runs of small and large digits in either order), BFV (also
    N = 2^15, the size of BASELINE configs[3], where the tensor product is formed inside the two-pass inverse), a digit-parallel slice"""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import seal_amd as S; S.load(%r); import parity_cases as P\n"
            "from oracle import coeff_modulus_create, plain_modulus_batching\n"
            "P.case_ckks_pipeline(8192, [60, 40, 50, 60], batch=2, steps=(1,), check_transforms=False)\n"
            "P.case_ckks_pipeline(16384, [60, 45, 50, 60], batch=1, steps=(1,), check_transforms=False)\n"
            "P.case_ckks_pipeline(32768, [50, 60, 50, 60], batch=1, steps=(1,), check_transforms=False)\n"
            "P.case_ckks_pipeline(65536, [60, 50, 50, 60], batch=1, steps=(1,), check_transforms=False)\n"
            "P.case_bfv_pipeline(8192, coeff_modulus_create(8192, [50, 55, 56]), plain_modulus_batching(8192, 20), batch=1)\n"
            "P.case_bfv_pipeline(32768, coeff_modulus_create(32768, [55, 55, 56]), plain_modulus_batching(32768, 20), batch=1)\n"
            "P.case_digit_parallel('ckks', 8192, coeff_modulus_create(8192, [50, 40, 60, 50, 50]), parts=3, batch=1)\n"
            "print('digit-resident ok')\n" % (here, os.path.dirname(here), os.path.join(here, "hipemu", "libsealhip_emu.so")))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SEALHIP_KS1_ORDER="1"), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "digit-resident ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_ntt_two_pass_engine_mixed_kernel(emu, monkeypatch):
    """SEALHIP_NTT_NOSPLIT=1: one mixed-back-end launch instead of one launch per class run."""
    monkeypatch.setenv("SEALHIP_NTT_NOSPLIT", "1")
    P.case_ntt(8192, [60, 50, 30], polys=2)


# fused key switching + fused tails at engine sizes: CKKS (diagonal shortcut), BFV (no shortcut, BEHZ)
@pytest.mark.parametrize("n,bits,batch,steps", [
    (8192, [50, 40, 60, 50], 2, (1,)),
    (8192, [60, 40, 40, 60], 1, (-1,)),
    (16384, [60, 50, 50, 60], 1, (1,)),
    (8192, [60] + [30, 40, 50, 45, 36] * 4 + [60], 1, (1,)),   # 22 primes of unequal sizes: 21 digits, both classes interleaved
])
def test_ckks_pipeline_engine_sizes(emu, n, bits, batch, steps):
    P.case_ckks_pipeline(n, bits, batch=batch, steps=steps)


def test_bfv_pipeline_engine_size(emu):
    primes, t = P.default_bfv_params(8192, [50, 55, 56], 20)
    P.case_bfv_pipeline(8192, primes, t, batch=1)


def test_golden_engine_digests_index_arithmetic(emu):
    """same digests through the emulated kernels (index arithmetic + host logic only)"""
    P.case_golden_engine("ckks_n8192_fp_and_int")


def test_field_arithmetic_exact():
    """seal_amd/csrc/field.h: the double-precision residue arithmetic is exact and stays inside its
    documented ranges (checked against 128-bit integers, 2.8 M random and adversarial cases)."""
    import os
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    exe = os.path.join(here, "hipemu", "obj", "field_check")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-mfma", "-DSEALHIP_CHECK_BOUNDS", "-I" + os.path.join(here, "hipemu", "include"),
                           "-I" + os.path.join(root, "seal_amd", "csrc"), os.path.join(here, "field_check.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "field_check ok" in out.stdout, out.stdout + out.stderr


def test_dyadic(emu):
    P.case_dyadic(256, [60, 40, 30])


@pytest.mark.parametrize("n,bits,batch,steps", [
    (16, [30, 30, 30, 30], 2, (1, -1)),
    (128, [40, 30, 40], 1, (3,)),
    (1024, [50, 40, 40, 50], 3, (1,)),
])
def test_ckks_pipeline(emu, n, bits, batch, steps):
    P.case_ckks_pipeline(n, bits, batch=batch, steps=steps)


@pytest.mark.parametrize("n,bits,tb,batch", [
    (16, [30, 30, 30, 30], 12, 2),
    (256, [40, 40, 41], 16, 1),
    (4096, [36, 36, 37], 20, 1),      # BASELINE config 1: BFVDefault(4096) bit sizes, Batching(4096, 20)
])
def test_bfv_pipeline(emu, n, bits, tb, batch):
    primes, t = P.default_bfv_params(n, bits, tb)
    P.case_bfv_pipeline(n, primes, t, batch=batch)


@pytest.mark.parametrize("n,bits,tb,batch", [
    (16, [30, 30, 30, 30], 12, 2),
    (1024, [40, 40, 41, 42], 16, 1),
    (8192, [50, 55, 56], 20, 1),      # two-pass engine: fused epilogues with the BGV correction as the source
])
def test_bgv_pipeline(emu, n, bits, tb, batch):
    import sealref
    if not sealref.available():
        pytest.skip("BGV parity needs the real reference (oracle/_ref)")
    primes, t = P.default_bfv_params(n, bits, tb)
    P.case_bgv_pipeline(n, primes, t, batch=batch)


@pytest.mark.parametrize("scheme,n,bits,tb,parts", [
    ("ckks", 64, [40, 30, 30, 40], 0, 2),
    ("ckks", 8192, [50, 40, 60, 50, 50], 0, 3),      # fused kernels, both back ends, uneven split (2, 1, 1)
    ("ckks", 8192, [60, 40, 60], 0, 4),               # more ranks than digits: two ranks contribute zeros
    ("bfv", 256, [40, 40, 41], 16, 2),
    ("bgv", 128, [40, 40, 41, 42], 16, 2),
])
def test_digit_parallel_key_switch(emu, scheme, n, bits, tb, parts):
    primes = coeff_modulus_create(n, bits)
    t = plain_modulus_batching(n, tb) if tb else 0
    P.case_digit_parallel(scheme, n, primes, t, parts=parts, batch=2)


@pytest.mark.parametrize("scheme,n,bits,tb", [
    ("ckks", 64, [40, 30, 30, 40], 0),
    ("bfv", 64, [40, 40, 41], 13),
    ("bfv", 128, [30, 30, 30], 40),       # t above every q_i: the lift cannot use the "fast plain lift" shortcut
    ("bgv", 64, [40, 40, 41], 13),
    ("bfv", 8192, [50, 55, 56], 20),      # two-pass engine
])
def test_plain_operands_and_many(emu, scheme, n, bits, tb):
    import sealref
    if not sealref.available():
        pytest.skip("needs the real reference (oracle/_ref)")
    primes = coeff_modulus_create(n, bits)
    t = plain_modulus_batching(n, tb) if tb else 0
    P.case_plain_ops(scheme, n, primes, t, batch=2 if n < 4096 else 1)


def test_rns_stages(emu):
    primes, t = P.default_bfv_params(64, [40, 40, 40, 40], 13)
    P.case_rns_stages(64, primes, t)


def test_kats_on_device_path(emu):
    """The reference's NTT / Galois known answers (native/tests/seal/util/ntt.cpp:75-101,
    galois.cpp:86-120) through the device path."""
    import seal_amd as S
    from harness import DeviceSide
    d = DeviceSide("ckks", 2, [0xFFFFFFFFFFC0001])
    assert d.ctx.ntt_root(0) == 288794978602139552
    buf = S.DeviceBuffer.from_numpy(np.array([1, 1], dtype=np.uint64))
    S.ntt_forward(d.ctx, buf, 1, 1)
    assert list(buf.to_numpy((2,))) == [288794978602139553, 864126526004445282]
    d = DeviceSide("ckks", 8, [17])
    src = S.DeviceBuffer.from_numpy(np.arange(8, dtype=np.uint64))
    dst = S.DeviceBuffer(8)
    S.apply_galois(d.ctx, 0, False, 3, src, dst, 1)
    assert list(dst.to_numpy((8,))) == [0, 14, 6, 1, 13, 7, 2, 12]
    S.apply_galois(d.ctx, 0, True, 3, src, dst, 1)
    assert list(dst.to_numpy((8,))) == [4, 5, 7, 6, 1, 0, 2, 3]


def test_context_constants_match_oracle(emu):
    import seal_amd as S
    n = 1024
    assert S.CoeffModulus.Create(n, [60, 40, 40, 60]) == coeff_modulus_create(n, [60, 40, 40, 60])
    assert S.PlainModulus.Batching(n, 20) == plain_modulus_batching(n, 20)


def test_mod_reduce(emu):
    import sealref
    if not sealref.available():
        pytest.skip("oracle/_ref (the real reference) is not built")
    P.case_mod_reduce(1024, [40, 30, 30, 40])
    P.case_mod_reduce(8192, [50, 40, 60, 50], batch=1)


@pytest.mark.parametrize("scheme,n,bits", [("ckks", 4096, [54, 42, 55]), ("ckks", 8192, [50, 40, 60]), ("bgv", 2048, [40, 40, 45]),
                                            ("bfv", 1024, [36, 36, 37])])
def test_product_growth(emu, scheme, n, bits):
    """the 2 x 2 product into a new slab and in place (the shape of the one device fuzz mismatch of round 3: CKKS, N = 4096,
    three items, square of a fresh ciphertext)"""
    P.case_product_growth(scheme, n, bits)


@pytest.mark.parametrize("groups", ["auto", "1"])
def test_deferred_tail_two_readers(emu, monkeypatch, groups):
    """groups "1": the key switch runs its digits as one group (ks2 itself leaves c + S P^-1 behind: KsFusedArgs::fold_c0)"""
    if groups != "auto":
        monkeypatch.setenv("SEALHIP_KS_SPLIT", groups)
    P.case_deferred_tail_two_readers(8192, (50, 40, 60), rounds=2)


def test_multi_level_forms(emu):
    """rescale_to_inplace / mod_switch_to_inplace(Ciphertext) over three levels against the reference's own multi-level calls
    (evaluator.cpp:1451-1473, 1543-1595), also with a key switch's deferred tail pending (missing #3)"""
    import sealref
    if not sealref.available():
        pytest.skip("oracle/_ref (the real reference) is not built")
    P.case_multi_level_ckks(1024, [40, 30, 30, 30, 30, 40])
    P.case_multi_level_ckks(8192, [50, 30, 40, 30, 40, 50], batch=1)       # two-pass size: deferred tails
    primes = coeff_modulus_create(1024, [36, 36, 36, 36, 36, 37])
    t = plain_modulus_batching(1024, 20)
    P.case_multi_level_bfv_bgv("bfv", 1024, primes, t)
    P.case_multi_level_bfv_bgv("bgv", 1024, primes, t)


@pytest.mark.parametrize("n,bits,parts,batch", [
    (64, [40, 30, 30, 40], 2, 2),
    (1024, [50, 40, 40, 50], 4, 2),          # more ranks than digits: one rank has neither digits nor moduli
    (128, [40, 30, 30, 30, 30, 40], 3, 1),
    (8192, [60, 40, 40, 50, 60], 3, 2),      # two-pass engine, fused tail
])
def test_digit_parallel_reduce_scatter(emu, n, bits, parts, batch):
    """the reduce-scatter shape of the key-switch exchange (sealhip.h section 1c) with emulated ranks, partition arithmetic
    of pack / owned mod-down / add; and the library's driver + key broadcast on a one-rank (loopback) communicator"""
    primes = coeff_modulus_create(n, bits)
    assert P.case_digit_parallel_reduce_scatter(n, primes, parts=parts, batch=batch) is True  # loopback: no RCCL on this box


def test_ntt_two_pass_loop_packed_intermediate(emu, monkeypatch):
    """N = 2^16 / 2^15 with the per-workgroup loop forced at a small batch (SEALHIP_NTT_CHUNKS, development builds): pass 2 with hoisted
    twiddles and, at 2^16, the packed 52-bit intermediate of the double-precision components (ntt2_kernels.hip: kPackWords) next to an
    integer-class component that keeps the plain one"""
    monkeypatch.setenv("SEALHIP_NTT_CHUNKS", "1")
    P.case_ntt(65536, [50, 60, 40], polys=3)
    P.case_ntt(32768, [50, 45], polys=3)


def test_digit_parallel_key_switch_in_chunks(emu, monkeypatch):
    """the digit-parallel halves (switch_key_partial over a rank's digit range, *Finish with the deferred tail) when the batch is
    cut into chunks on lanes - what a multi-GPU rotate_c5 at batch >= 48 runs: every emulated rank's slice in chunks of one item"""
    monkeypatch.setenv("SEALHIP_KS_CHUNK", "1")
    monkeypatch.setenv("SEALHIP_KS_LANES", "2")
    c0, k0, _ = emu.ks_chunk_stats()
    P.case_digit_parallel("ckks", 8192, coeff_modulus_create(8192, [50, 40, 60, 50, 50]), parts=2, batch=3)
    c1, k1, _ = emu.ks_chunk_stats()
    assert c1 > c0 and k1 - k0 >= 3 * (c1 - c0), "the partial key switches did not run in chunks"


def test_ckks_tensor_product_wide_kernel(emu, monkeypatch):
    """the CKKS 2 x 2 product in its large-batch shape (one 4 KiB chunk per workgroup, 16 bytes per thread: poly_kernels.hip,
    ckks_multiply_2x2_wide_kernel) forced at a small batch: both arithmetic classes, the ragged last workgroup"""
    monkeypatch.setenv("SEALHIP_TENSOR_WIDE_MIN", "0")
    P.case_ckks_pipeline(8192, [60, 40, 50, 60], batch=1, steps=(), check_transforms=False)
    P.case_product_growth("ckks", 4096, [54, 42, 55])


def test_ks_chunked(emu):
    """chunked key switching (sealhip.h: SEALHIP_KS_CHUNK / SEALHIP_KS_LANES / SEALHIP_KS_SCRATCH_CAP_MIB): pointer arithmetic of the
    chunks, ragged last chunk, the folded CKKS tail over chunked sums, BFV's in-place target, the scratch cap"""
    P.case_ks_chunked("ckks", 8192, [50, 40, 40, 50], batch=5, chunk=2, lanes=2)
    pr = coeff_modulus_create(8192, [45, 40, 45])
    P.case_ks_chunked("bfv", 8192, pr, batch=4, chunk=1, lanes=3, t=plain_modulus_batching(8192, 20))
    P.case_ks_chunked("ckks", 8192, [50, 40, 40, 50], batch=7, chunk=4, lanes=2, cap_mib=2)


def test_ckks_pipeline_n65536_lean_key_switch(emu):
    """N = 2^16 is the one size whose key-switch kernels use the lean fix() placement (two per sixteen stages, balanced
    tables; ntt2_kernels.hip p1_tile / p2_tile): multiply + relinearize + rescale + rotate word for word, incl. 60-bit digits
    feeding double-precision targets and the other way round"""
    P.case_ckks_pipeline(65536, [60, 50, 50, 60], batch=1, steps=(1,), check_transforms=False)
    P.case_ckks_pipeline(65536, [60, 50, 40, 50, 45, 60], batch=1, steps=(1,), check_transforms=False)


@pytest.mark.parametrize("groups", ["auto", "1"])
def test_deferred_tail_lifecycle(emu, monkeypatch, groups):
    """deferred key-switch tails (sealhip.h): folded into a rescale by their owner, completed by anyone else who needs the words;
    groups "1": the digits as one group (ks2 itself adds the ciphertext's words: KsFusedArgs::fold_c0)"""
    if groups != "auto":
        monkeypatch.setenv("SEALHIP_KS_SPLIT", groups)
    P.case_deferred_tail_lifecycle()


# structured worst-case inputs (parity_cases.case_extremes_*): the emulated kernels assert the bounds of both back ends
# (SEALHIP_CHECK_BOUNDS) on all q-1 / alternating / floor(q/2) slabs - the GPU suite runs the same cases at every plan
@pytest.mark.parametrize("n,bits", [(1024, [36, 60, 55]), (8192, [50, 60, 58]), (65536, [49, 60, 57])])
def test_extremes_ntt(emu, n, bits):
    P.case_extremes_ntt(n, bits, 8)


def test_extremes_ckks(emu):
    P.case_extremes_ckks(8192, [60, 40, 58, 60], batch=8)


def test_extremes_bfv(emu):
    P.case_extremes_bfv(4096, coeff_modulus_create(4096, [36, 36, 37]), plain_modulus_batching(4096, 20))


def test_lazy_product_life_cycle(emu):
    """deferred tensor products (evaluator.h: LazyProduct): the fused relinearisation and every way the caller can observe the words of
    the destination or the operands while a product is pending - both arithmetic classes in the chain"""
    P.case_lazy_product(8192, [60, 40, 40, 60], batch=2)


def test_rotate_gather(emu):
    """rotations that read their operand through the automorphism's index map inside the key switch (both arithmetic classes)"""
    P.case_rotate_gather(8192, [60, 40, 40, 60], batch=2)


def test_pending_product_threads(emu):
    """two host threads around pending tensor products: one destination read by both, one operand shared by two fused relinearisations"""
    P.case_pending_product_threads(8192, (50, 40, 60), rounds=2)
