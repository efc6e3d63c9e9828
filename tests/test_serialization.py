"""CPU: the reference's wire format through the C ABI with the kernels emulated — parser, seed expansion
(Blake2xb / SHAKE256 restated in seal_amd/csrc), save, validity checks and exception classes against the REAL reference
(oracle/_ref) where it is built, and against committed golden streams (tests/golden/serial_*.bin, made by
tests/golden/make_serial_golden.py from the real reference) everywhere."""
import hashlib
import json
import os

import numpy as np
import pytest

import sealref

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
needs_ref = pytest.mark.skipif(not sealref.available(), reason="oracle/_ref (the real reference) is not built")

SMALL = [("ckks", 1024, [40, 30, 40]), ("bfv", 1024, [36, 36, 37]), ("bgv", 2048, [40, 40, 45])]


@needs_ref
@pytest.mark.parametrize("scheme,n,bits", SMALL + [("ckks", 4096, [50, 40, 40, 50]), ("ckks", 64, [30]), ("bfv", 1024, [30, 30, 30, 30, 40])])
def test_parms_ids_are_the_references_hashes(emu, scheme, n, bits):
    import serial_cases as SC
    SC.case_parms_ids(scheme, n, bits)


@needs_ref
@pytest.mark.parametrize("scheme,n,bits", SMALL)
def test_ciphertext_streams(emu, scheme, n, bits):
    import serial_cases as SC
    SC.case_ciphertext_streams(scheme, n, bits)


@needs_ref
@pytest.mark.parametrize("scheme,n,bits", SMALL)
def test_evaluated_ciphertext_roundtrip(emu, scheme, n, bits):
    import serial_cases as SC
    SC.case_evaluated_ciphertext_roundtrip(scheme, n, bits)


@needs_ref
def test_bgv_coefficient_form_stream(emu):
    import serial_cases as SC
    SC.case_bgv_coefficient_form_stream(2048, [40, 40, 45])


@needs_ref
@pytest.mark.parametrize("scheme,n,bits", SMALL[:2])
def test_batch_items(emu, scheme, n, bits):
    import serial_cases as SC
    SC.case_batch_items(scheme, n, bits)


@needs_ref
@pytest.mark.parametrize("scheme,n,bits", [("ckks", 1024, [40, 30, 40]), ("bfv", 1024, [36, 36, 37]), ("ckks", 8192, [60, 40, 59])])
def test_key_save(emu, scheme, n, bits):
    import serial_cases as SC
    SC.case_key_save(scheme, n, bits)


@needs_ref
@pytest.mark.parametrize("scheme,n,bits,seeded", [("ckks", 1024, [40, 30, 40], True), ("bfv", 1024, [36, 36, 37], False),
                                                  ("bgv", 2048, [40, 40, 45], True), ("ckks", 8192, [50, 40, 60], True),
                                                  ("ckks", 8192, [60, 59, 60], True)])
def test_key_streams(emu, scheme, n, bits, seeded):
    import serial_cases as SC
    SC.case_key_streams(scheme, n, bits, seeded)


@needs_ref
@pytest.mark.parametrize("scheme,n,bits", SMALL)
def test_malformed_streams_fail_like_the_reference(emu, scheme, n, bits):
    import serial_cases as SC
    SC.case_malformed_streams(scheme, n, bits)


@needs_ref
@pytest.mark.parametrize("scheme,n,bits", SMALL)
def test_plaintext_streams(emu, scheme, n, bits):
    import serial_cases as SC
    SC.case_plaintext_streams(scheme, n, bits)


@needs_ref
def test_malformed_key_streams(emu):
    import serial_cases as SC
    SC.case_malformed_key_streams("ckks", 1024, [40, 30, 40])


def test_golden_streams(emu):
    """committed streams written by the real reference: loaded words (SHA-256) and re-saved bytes match what the reference
    itself produced when the fixture was made"""
    import seal_amd as S
    from harness import DeviceSide
    meta = json.load(open(os.path.join(GOLD, "serial_golden.json")))
    for g in meta["ciphertexts"]:
        d = DeviceSide(g["scheme"], g["n"], g["primes"], g["plain_modulus"])
        for ci, pid in enumerate(g["parms_ids"]):
            assert list(d.ctx.parms_id_at(ci)) == pid
        data = open(os.path.join(GOLD, g["file"]), "rb").read()
        ct = S.Ciphertext(d.ctx)
        assert ct.load_bytes(data) == len(data)
        words = ct.to_numpy()[:, 0]
        assert hashlib.sha256(np.ascontiguousarray(words).tobytes()).hexdigest() == g["sha256_words"], g["file"]
        assert (ct.size(), ct.is_ntt_form(), ct.scale(), ct.correction_factor()) == (g["size"], g["is_ntt_form"], g["scale"], g["correction_factor"])
        assert hashlib.sha256(ct.save_bytes()).hexdigest() == g["sha256_full_stream"], g["file"]
    for g in meta["keys"]:
        d = DeviceSide(g["scheme"], g["n"], g["primes"], g["plain_modulus"])
        data = open(os.path.join(GOLD, g["file"]), "rb").read()
        rlk = S.RelinKeys(d.ctx)
        assert rlk.load_bytes(data) == len(data)
        K = len(g["primes"]) - 1
        x3 = np.stack([np.stack([np.random.default_rng(100 + p * 16 + i).integers(0, g["primes"][i], g["n"], dtype=np.uint64) for i in range(K)]) for p in range(3)])
        cx = d.ct(x3, scale=2.0 ** 10)
        d.ev.relinearize_inplace(cx, rlk)
        assert hashlib.sha256(np.ascontiguousarray(cx.to_numpy()[:, 0]).tobytes()).hexdigest() == g["sha256_relinearized"], g["file"]


# ---- decryption on the device and the whole server flow on byte streams
@needs_ref
@pytest.mark.parametrize("scheme,n,bits", [("ckks", 1024, [40, 30, 30, 40]), ("bfv", 1024, [36, 36, 37]), ("bgv", 2048, [40, 40, 45]), ("ckks", 8192, [50, 40, 60])])
def test_decrypt(emu, scheme, n, bits):
    import decrypt_cases as DC
    DC.case_decrypt(scheme, n, bits)


@needs_ref
@pytest.mark.parametrize("scheme,n,bits", [("ckks", 1024, [40, 30, 30, 40]), ("bfv", 2048, [36, 36, 37]), ("bgv", 2048, [40, 40, 45])])
def test_end_to_end_streams(emu, scheme, n, bits):
    import decrypt_cases as DC
    DC.case_end_to_end_streams(scheme, n, bits)


@needs_ref
@pytest.mark.parametrize("scheme,n,bits", [("ckks", 1024, [40, 30, 30, 40]), ("bfv", 1024, [36, 36, 37]), ("bgv", 2048, [40, 40, 45]),
                                           ("ckks", 32768, [50, 55]),    # large enough for the threaded bulk PRNG draws
                                           ("ckks", 8192, [60, 60, 60]), ("bfv", 4096, [50, 59])])   # ~3 % of a's words rejected and redrawn
def test_encrypt_symmetric(emu, scheme, n, bits):
    import decrypt_cases as DC
    DC.case_encrypt_symmetric(scheme, n, bits)


@needs_ref
@pytest.mark.parametrize("scheme,n,bits", [("ckks", 1024, [40, 30, 30, 40]), ("bfv", 2048, [36, 36, 37]), ("bgv", 4096, [50, 50, 58])])
def test_keygen(emu, scheme, n, bits):
    import decrypt_cases as DC
    DC.case_keygen(scheme, n, bits)


@needs_ref
def test_encrypt_with_host_sampling(emu, monkeypatch):
    """u, e are normally drawn on the device from the bootstrap stream; the host branch (taken when a ternary draw is redrawn) must
    produce the same bytes."""
    import decrypt_cases as DC
    monkeypatch.setenv("SEALHIP_ENCRYPT_HOST_SAMPLING", "1")
    DC.case_encrypt_symmetric("ckks", 1024, [40, 30, 30, 40])
    DC.case_encrypt_asymmetric("bgv", 2048, [40, 40, 45])


@needs_ref
@pytest.mark.parametrize("scheme,n,bits", [("bfv", 1024, [36, 36, 37]), ("bgv", 2048, [40, 40, 45]), ("bfv", 8192, [50, 55, 56])])
def test_batch_encoder(emu, scheme, n, bits):
    import decrypt_cases as DC
    DC.case_batch_encoder(scheme, n, bits)


def test_batch_encoder_needs_batching(emu):
    import seal_amd as S
    from harness import DeviceSide
    from oracle import coeff_modulus_create
    with pytest.raises(S.InvalidArgument):
        S.BatchEncoder(DeviceSide("ckks", 1024, coeff_modulus_create(1024, [40, 40])).ctx)
    with pytest.raises(S.InvalidArgument):
        S.BatchEncoder(DeviceSide("bfv", 1024, coeff_modulus_create(1024, [40, 40]), 1 << 10).ctx)   # t not prime


@needs_ref
@pytest.mark.parametrize("scheme,n,bits", [("bfv", 4096, [50, 50, 50, 58]), ("bgv", 4096, [50, 50, 50, 58])])
def test_slot_semantics(emu, scheme, n, bits):
    import decrypt_cases as DC
    DC.case_slot_semantics(scheme, n, bits)


@needs_ref
@pytest.mark.parametrize("scheme,n,bits", SMALL)
def test_compressed_streams(emu, scheme, n, bits):
    import serial_cases as SC
    SC.case_compressed_streams(scheme, n, bits)


@needs_ref
@pytest.mark.parametrize("scheme,n,bits", [("ckks", 1024, [40, 30, 30, 40]), ("bfv", 1024, [36, 36, 37]), ("bgv", 2048, [40, 40, 45]), ("bfv", 1024, [40]),
                                           ("bfv", 32768, [50, 55])])   # the last: large enough for the threaded bulk PRNG draws
def test_encrypt_asymmetric(emu, scheme, n, bits):
    import decrypt_cases as DC
    DC.case_encrypt_asymmetric(scheme, n, bits)


@needs_ref
@pytest.mark.parametrize("n,bits", [(1024, [40, 30, 30, 40]), (4096, [50, 40, 40, 50]), (8, [30, 30]), (1024, [60, 50, 50, 50, 50, 60])])
def test_ckks_encoder(emu, n, bits):
    import decrypt_cases as DC
    DC.case_ckks_encoder(n, bits)


# ---- the reference's example programs on the device API
@needs_ref
def test_example_ckks_basics(emu):
    import example_cases as EC
    EC.example_ckks_basics(4096, (60, 40, 40, 60))


@needs_ref
def test_example_batching_rotation(emu):
    import example_cases as EC
    EC.example_batching_rotation(4096, (36, 36, 37))


@pytest.mark.parametrize("scheme,n,bits", [("ckks", 1024, [40, 30, 30, 40]), ("bfv", 2048, [36, 36, 37]), ("ckks", 8192, [60, 59, 60])])
def test_shake256_seeded_streams(emu, scheme, n, bits):
    import serial_cases as SC
    SC.case_shake256_seeded_streams(scheme, n, bits)
