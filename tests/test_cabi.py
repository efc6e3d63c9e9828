"""CPU: the C-ABI shared library (gfx950 build) loads, exports every symbol include/sealhip.h declares,
and refuses to work without a device (no CPU fallback).  No compute is launched here."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GPU_LIB = os.path.join(ROOT, "seal_amd", "lib", "libsealhip.so")
HEADER = os.path.join(ROOT, "include", "sealhip.h")


def declared_symbols():
    text = open(HEADER).read()
    return sorted(set(re.findall(r"SHL_FUNC\s+(\w+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(GPU_LIB):
        import subprocess
        subprocess.check_call(["make", "-s", "-j8", "gpu"], cwd=os.path.join(ROOT, "seal_amd", "csrc"))
    return C.CDLL(GPU_LIB)


def test_header_declares_the_hot_path():
    syms = declared_symbols()
    for must in ["Evaluator_Multiply", "Evaluator_Relinearize", "Evaluator_RescaleToNext", "Evaluator_RotateVector",
                 "Evaluator_ModSwitchToNext1", "Evaluator_ApplyGalois", "shl_ntt_forward", "shl_ntt_inverse",
                 "shl_dyadic_product", "shl_rns_stage"]:
        assert must in syms


def test_library_exports_every_declared_symbol(lib):
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, "declared in include/sealhip.h but not exported: %s" % missing


def test_python_binding_lists_every_declared_symbol():
    from seal_amd import _native
    assert sorted(_native.SYMBOLS) == declared_symbols()


def test_null_handles_return_e_pointer(lib):
    lib.Evaluator_Multiply.restype = C.c_long
    assert lib.Evaluator_Multiply(None, None, None, None, None) & 0xFFFFFFFF == 0x80004003  # E_POINTER
    lib.SEALContext_Create.restype = C.c_long
    assert lib.SEALContext_Create(None, True, 0, None) & 0xFFFFFFFF == 0x80004003


def test_no_cpu_fallback_without_a_device(lib):
    """On a machine without a HIP device the context constructor must fail loudly (COR_E_IO), never
    fall back to a CPU path.  (On a GPU box this test is vacuous and passes.)"""
    n = C.c_int(0)
    hip = C.CDLL("libamdhip64.so")
    have_gpu = hip.hipGetDeviceCount(C.byref(n)) == 0 and n.value > 0
    lib.EncParams_Create1.restype = C.c_long
    lib.SEALContext_Create.restype = C.c_long
    p = C.c_void_p()
    assert lib.EncParams_Create1(C.c_uint8(2), C.byref(p)) == 0
    lib.EncParams_SetPolyModulusDegree(p, C.c_uint64(8))
    arr = (C.c_uint64 * 1)(17)
    lib.EncParams_SetCoeffModulus(p, C.c_uint64(1), arr)
    ctx = C.c_void_p()
    hr = lib.SEALContext_Create(p, True, 0, C.byref(ctx)) & 0xFFFFFFFF
    if have_gpu:
        assert hr == 0
        lib.SEALContext_Destroy(ctx)
    else:
        assert hr == 0x80131620  # COR_E_IO: "no HIP device visible: libsealhip has no CPU fallback"
    lib.EncParams_Destroy(p)
