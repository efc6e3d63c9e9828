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


# ---- the container headers of sealc against include/sealhip.h
REF_C = "/root/reference/native/src/seal/c"
CONTAINER_HEADERS = ["ciphertext", "kswitchkeys", "sealcontext", "contextdata", "encryptionparameters", "secretkey", "publickey"]


def _signatures(text, marker):
    """{name: [argument types]} of every `marker name(args);` declaration; parameter names and const are dropped"""
    out = {}
    for name, args in re.findall(marker + r"\s+(\w+)\s*\(([^)]*)\)\s*;", text):
        types = []
        for a in [x.strip() for x in args.split(",")]:
            if a in ("", "void"):
                continue
            a = re.sub(r"\bconst\b", "", a)
            a = re.sub(r"\w+$", "", a.strip()) if not a.strip().endswith("*") else a   # drop the parameter name
            types.append(re.sub(r"\s+", "", a))
        out[name] = types
    return out


def _listed(text, heading):
    """the function names in the comment block of sealhip.h that starts with `heading` (up to the next block)"""
    start = text.index(heading)
    end = min(i for i in (text.find("SAME NAME, DIFFERENT ARGUMENTS", start + 1), text.find("Everything else of those headers", start)) if i > 0)
    return set(re.findall(r"\b(?:Ciphertext|KSwitchKeys|SEALContext|ContextData|EncParams|SecretKey|PublicKey)_\w+", text[start:end]))


@pytest.mark.skipif(not os.path.isdir(REF_C), reason="the reference tree is only in the build container")
def test_container_headers_are_covered_or_listed():
    text = open(HEADER).read()
    ours = _signatures(text, "SHL_FUNC")
    absent = _listed(text, "DELIBERATELY ABSENT")
    differ = _listed(text, "SAME NAME, DIFFERENT ARGUMENTS")
    # the explanations may mention functions that DO exist here (what to use instead): only undeclared names count as absent
    absent = {n for n in absent if n not in ours}
    problems = []
    for h in CONTAINER_HEADERS:
        ref = _signatures(open(os.path.join(REF_C, h + ".h")).read(), "SEAL_C_FUNC")
        assert ref, h
        for name, types in ref.items():
            if name in absent:
                continue
            if name not in ours:
                problems.append("%s.h: %s is neither declared nor listed as deliberately absent" % (h, name))
            elif name not in differ and ours[name] != types:
                problems.append("%s.h: %s%r is declared here as %r" % (h, name, types, ours[name]))
    assert not problems, "\n".join(problems)
    # and the lists do not rot: every listed name is a sealc function of these headers
    all_ref = set()
    for h in CONTAINER_HEADERS:
        all_ref |= set(_signatures(open(os.path.join(REF_C, h + ".h")).read(), "SEAL_C_FUNC"))
    assert absent <= all_ref and {n for n in differ if n in ours} <= all_ref | {"SecretKey_Assign", "PublicKey_Assign"}, (absent - all_ref, differ - all_ref)
