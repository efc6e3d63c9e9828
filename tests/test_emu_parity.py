"""CPU: the product's host logic (context tables, orchestration, metadata) and the kernels' index
arithmetic, exercised by running the SAME kernel sources under the fiber emulator (tests/hipemu) at
small sizes and checking them against the oracle.  This proves nothing about the GPU build — the
`-m gpu` tests do that — but it lets the host side and every index formula be tested without a GPU."""
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
