"""Randomised operation sequences against the real reference (tests/fuzz_cases.py).  CPU: small degrees on the fiber
emulator (host logic + index arithmetic); GPU: the same generator over every kernel family (small single-pass kernels,
the two-pass engine with both arithmetic back ends in one context, BEHZ, BGV)."""
import numpy as np
import pytest

import fuzz_cases as F
import sealref


def _configs(seed, count, degrees):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(count):
        scheme = ["ckks", "bfv", "bgv"][int(rng.integers(0, 3))]
        n = int(degrees[rng.integers(0, len(degrees))])
        L = int(rng.integers(3, 7))
        lo = 32 if n >= 4096 else (24 if n >= 64 else 20)
        bits = [int(b) for b in rng.integers(lo, 61, L)]
        if scheme != "ckks":
            bits = [max(b, 30) for b in bits]
        tb = 20 if n >= 1024 else int(rng.integers(13, 21))   # PlainModulus::Batching(n, 20) exists for every degree used here
        out.append((scheme, n, bits, tb, int(rng.integers(1, 4)), int(rng.integers(4, 9)), 1000 * seed + i))
    return out


@pytest.mark.parametrize("cfg", _configs(11, 12, [16, 64, 128, 256]), ids=lambda c: "%s-%d-%s" % (c[0], c[1], "_".join(map(str, c[2]))))
def test_random_sequences_emulated(emu, cfg):
    if not sealref.available():
        pytest.skip("needs the real reference (oracle/_ref)")
    try:
        F.run_sequence(*cfg)
    except sealref.RefError as e:   # a parameter set the reference itself rejects (e.g. no batching prime of that size)
        pytest.skip("reference rejected the parameters: %s" % e)


@pytest.mark.parametrize("cfg", _configs(12, 16, [16, 64, 128, 256]), ids=lambda c: "%s-%d-%s" % (c[0], c[1], "_".join(map(str, c[2]))))
def test_wild_sequences_emulated(emu, cfg):
    """operations drawn without the generator's guards: calls the device rejects are replayed on the reference, which must
    reject them with the same exception class - and calls the reference rejects must not be accepted"""
    if not sealref.available():
        pytest.skip("needs the real reference (oracle/_ref)")
    try:
        F.run_sequence(*cfg, wild_prob=0.3)
    except sealref.RefError as e:
        pytest.skip("reference rejected the parameters: %s" % e)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", _configs(13, 24, [32, 1024, 4096, 8192, 16384]), ids=lambda c: "%s-%d-%s" % (c[0], c[1], "_".join(map(str, c[2]))))
def test_wild_sequences_gpu(gpu, cfg):
    if not sealref.available():
        pytest.skip("needs the real reference (oracle/_ref)")
    try:
        F.run_sequence(*cfg, wild_prob=0.3)
    except sealref.RefError as e:
        pytest.skip("reference rejected the parameters: %s" % e)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", _configs(7, 36, [32, 512, 2048, 4096, 8192, 8192, 16384, 16384, 32768]) +
                         _configs(8, 24, [64, 1024, 8192, 16384, 32768, 65536]),
                         ids=lambda c: "%s-%d-%s" % (c[0], c[1], "_".join(map(str, c[2]))))
def test_random_sequences_gpu(gpu, cfg):
    if not sealref.available():
        pytest.skip("needs the real reference (oracle/_ref)")
    try:
        F.run_sequence(*cfg)
    except sealref.RefError as e:
        pytest.skip("reference rejected the parameters: %s" % e)


def _ckks_configs(seed, count, degrees):
    """CKKS only, more operations per sequence, a mix that favours key switches followed by rescales"""
    rng = np.random.default_rng(seed)
    out = []
    for i in range(count):
        n = int(degrees[rng.integers(0, len(degrees))])
        L = int(rng.integers(4, 8))
        bits = [int(b) for b in rng.integers(36, 61, L)]
        out.append(("ckks", n, bits, 20, int(rng.integers(1, 3)), int(rng.integers(8, 14)), 5000 * seed + i))
    return out


@pytest.mark.parametrize("cfg", _ckks_configs(3, 6, [8192, 8192, 16384]), ids=lambda c: "%s-%d-%s" % (c[0], c[1], "_".join(map(str, c[2]))))
def test_random_sequences_with_deferred_state_emulated(emu, cfg, monkeypatch):
    """the same sequences with the result read back only now and then: deferred key-switch tails reach the next operation.
    Every other sequence runs its key switches as ONE digit group, the form large batches take (ks2 adds the ciphertext's words)"""
    if not sealref.available():
        pytest.skip("needs the real reference (oracle/_ref)")
    if cfg[-1] % 2:
        monkeypatch.setenv("SEALHIP_KS_SPLIT", "1")
        monkeypatch.setenv("SEALHIP_LAZY_PRODUCT_MIN_WGS", "0")   # ... and leave three-object products pending (round 6) at these batches
    try:
        F.run_sequence(*cfg, check_prob=0.25, scale0=2.0 ** 30, three_object_prob=0.6)
    except sealref.RefError as e:
        pytest.skip("reference rejected the parameters: %s" % e)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", _ckks_configs(4, 24, [8192, 16384, 32768, 65536]), ids=lambda c: "%s-%d-%s" % (c[0], c[1], "_".join(map(str, c[2]))))
def test_random_sequences_with_deferred_state_gpu(gpu, cfg, monkeypatch):
    if not sealref.available():
        pytest.skip("needs the real reference (oracle/_ref)")
    if cfg[-1] % 2:
        monkeypatch.setenv("SEALHIP_KS_SPLIT", "1")
        monkeypatch.setenv("SEALHIP_LAZY_PRODUCT_MIN_WGS", "0")   # ... and leave three-object products pending (round 6) at these batches
    try:
        F.run_sequence(*cfg, check_prob=0.25, scale0=2.0 ** 30, three_object_prob=0.6)
    except sealref.RefError as e:
        pytest.skip("reference rejected the parameters: %s" % e)


def _bfv_configs(seed, count, degrees):
    """BFV only, sequences that favour key switches followed by mod switches (the deferred BFV tail of round 4)"""
    rng = np.random.default_rng(seed)
    out = []
    for i in range(count):
        n = int(degrees[rng.integers(0, len(degrees))])
        L = int(rng.integers(4, 7))
        bits = [int(b) for b in rng.integers(36, 61, L)]
        out.append(("bfv", n, bits, 20, int(rng.integers(1, 3)), int(rng.integers(8, 13)), 7000 * seed + i))
    return out


@pytest.mark.parametrize("cfg", _bfv_configs(5, 4, [8192, 8192, 16384]), ids=lambda c: "%s-%d-%s" % (c[0], c[1], "_".join(map(str, c[2]))))
def test_bfv_sequences_with_deferred_state_emulated(emu, cfg):
    if not sealref.available():
        pytest.skip("needs the real reference (oracle/_ref)")
    try:
        F.run_sequence(*cfg, check_prob=0.25)
    except sealref.RefError as e:
        pytest.skip("reference rejected the parameters: %s" % e)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", _bfv_configs(6, 16, [8192, 16384, 32768, 65536]), ids=lambda c: "%s-%d-%s" % (c[0], c[1], "_".join(map(str, c[2]))))
def test_bfv_sequences_with_deferred_state_gpu(gpu, cfg):
    if not sealref.available():
        pytest.skip("needs the real reference (oracle/_ref)")
    try:
        F.run_sequence(*cfg, check_prob=0.25)
    except sealref.RefError as e:
        pytest.skip("reference rejected the parameters: %s" % e)
