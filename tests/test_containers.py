"""The container surface of the C ABI (sealhip.h section 1d) against the real reference: every level's ContextData constants and
qualifiers, EncryptionParameters / SecretKey / PublicKey streams byte for byte, Ciphertext reserve / resize bookkeeping, KSwitchKeys
copies and key lists (tests/container_cases.py).  CPU: small degrees on the fiber emulator; GPU: the same cases at real sizes."""
import pytest

import container_cases as K
import sealref

needs_ref = pytest.mark.skipif(not sealref.available(), reason="needs the real reference (oracle/_ref)")

SMALL = [("ckks", 64, [40, 30, 30, 41]), ("bfv", 128, [36, 36, 37]), ("bgv", 64, [30, 31, 32, 33]), ("bfv", 64, [25, 25, 26], 27)]
LARGE = [("ckks", 8192, [60, 40, 40, 60]), ("bfv", 8192, [50, 55, 56]), ("bgv", 16384, [50, 50, 50, 55]), ("ckks", 65536, [60] + [50] * 14 + [60]),
         ("bfv", 4096, [25, 25, 26], 27)]


@needs_ref
@pytest.mark.parametrize("cfg", SMALL, ids=lambda c: "%s-%d-%d" % (c[0], c[1], len(c[2])))
def test_context_data_emulated(emu, cfg):
    K.case_context_data(*cfg)


@needs_ref
def test_security_level_emulated(emu):
    K.case_security_level()


@needs_ref
@pytest.mark.parametrize("cfg", SMALL[:3], ids=lambda c: "%s-%d" % (c[0], c[1]))
def test_ciphertext_container_emulated(emu, cfg):
    K.case_ciphertext_container(*cfg)


@needs_ref
@pytest.mark.parametrize("cfg", SMALL[:3], ids=lambda c: "%s-%d" % (c[0], c[1]))
def test_keys_container_emulated(emu, cfg):
    K.case_keys_container(*cfg)


@needs_ref
@pytest.mark.gpu
@pytest.mark.parametrize("cfg", LARGE, ids=lambda c: "%s-%d-%d" % (c[0], c[1], len(c[2])))
def test_context_data_gpu(gpu, cfg):
    K.case_context_data(*cfg)


@needs_ref
@pytest.mark.gpu
def test_security_level_gpu(gpu):
    K.case_security_level()


@needs_ref
@pytest.mark.gpu
@pytest.mark.parametrize("cfg", LARGE[:3], ids=lambda c: "%s-%d" % (c[0], c[1]))
def test_ciphertext_container_gpu(gpu, cfg):
    K.case_ciphertext_container(*cfg)


@needs_ref
@pytest.mark.gpu
@pytest.mark.parametrize("cfg", LARGE[:3], ids=lambda c: "%s-%d" % (c[0], c[1]))
def test_keys_container_gpu(gpu, cfg):
    K.case_keys_container(*cfg)
