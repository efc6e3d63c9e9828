"""ctypes binding of libsealhip.so (include/sealhip.h).

The only library this package ever loads is the gfx950 build at seal_amd/lib/libsealhip.so.
There is no CPU fallback: if the library is missing or no HIP device is visible the import of a
context fails loudly.  (tests/ may point `load()` at the fiber-emulated build of the same sources
to debug index arithmetic without a GPU; nothing in seal_amd/ does.)
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "lib", "libsealhip.so")

S_OK = 0
E_POINTER = 0x80004003
E_INVALIDARG = 0x80070057
E_OUTOFMEMORY = 0x8007000E
E_UNEXPECTED = 0x8000FFFF
COR_E_IO = 0x80131620
COR_E_INVALIDOPERATION = 0x80131509
E_INVALID_INDEX = 0x80070585


class SealHipError(Exception):
    """Raised for a failing HRESULT; subclasses mirror the C++ exception classes of the reference."""

    def __init__(self, hr, message):
        super().__init__("%s (HRESULT 0x%08X)" % (message, hr & 0xFFFFFFFF))
        self.hresult = hr & 0xFFFFFFFF
        self.message = message


class InvalidArgument(SealHipError, ValueError):  # std::invalid_argument -> E_INVALIDARG
    pass


class LogicError(SealHipError, RuntimeError):  # std::logic_error -> COR_E_INVALIDOPERATION
    pass


class OutOfRange(SealHipError, IndexError):  # std::out_of_range -> ERROR_INVALID_INDEX
    pass


class DeviceError(SealHipError, OSError):  # std::runtime_error (HIP failure) -> COR_E_IO
    pass


_ERR = {E_INVALIDARG: InvalidArgument, COR_E_INVALIDOPERATION: LogicError, E_INVALID_INDEX: OutOfRange,
        COR_E_IO: DeviceError}

# every symbol include/sealhip.h declares (the CPU test-suite checks the .so exports all of them)
SYMBOLS = """
SealHip_Version SealHip_DeviceInfo SealHip_LastError CoeffModulus_Create1 PlainModulus_Batching
EncParams_Create1 EncParams_Destroy EncParams_SetPolyModulusDegree EncParams_GetPolyModulusDegree
EncParams_SetCoeffModulus EncParams_GetCoeffModulus EncParams_SetPlainModulus2 EncParams_GetScheme
SEALContext_Create SEALContext_Destroy SEALContext_KeyParmsId SEALContext_FirstParmsId SEALContext_LastParmsId
SEALContext_UsingKeyswitching SEALContext_ChainIndex SEALContext_ParmsIdAt SEALContext_CoeffModulusAt
SEALContext_TotalCoeffModulusBitCount SEALContext_SetParmsId SEALContext_NTTRoot SEALContext_BaseBsk
Ciphertext_Create3 Ciphertext_CreateBatch Ciphertext_Create2 Ciphertext_Set Ciphertext_Destroy Ciphertext_Resize1
Ciphertext_Size Ciphertext_BatchCount Ciphertext_PolyModulusDegree Ciphertext_CoeffModulusSize Ciphertext_ParmsId
Ciphertext_IsNTTForm Ciphertext_SetIsNTTForm Ciphertext_Scale Ciphertext_SetScale Ciphertext_CorrectionFactor
Ciphertext_SetCorrectionFactor Ciphertext_IsTransparent Ciphertext_DevicePtr Ciphertext_CopyFromHost
Ciphertext_CopyToHost Ciphertext_CopyWordsToHost Ciphertext_CopyFromDevice
Ciphertext_SaveSize Ciphertext_Save Ciphertext_UnsafeLoad Ciphertext_Load Ciphertext_LoadItem Ciphertext_SaveItem
KSwitchKeys_UnsafeLoad KSwitchKeys_Load KSwitchKeys_SaveSize KSwitchKeys_Save
KeyGenerator_Create1 KeyGenerator_Create2 KeyGenerator_Destroy KeyGenerator_SecretKey KeyGenerator_CreatePublicKey
KeyGenerator_CreateRelinKeys KeyGenerator_CreateGaloisKeysFromSteps KeyGenerator_CreateGaloisKeysAll KeyGenerator_CreateGaloisKeysFromElts KeyGenerator_KeyToHost KeyGenerator_SeededSaveSize KeyGenerator_CreateRelinKeysSave KeyGenerator_CreateGaloisKeysFromEltsSave SecretKey_Get PublicKey_Get
SecretKey_Create SecretKey_Destroy SecretKey_Set SecretKey_UnsafeLoad SecretKey_Load Decryptor_Create Decryptor_Destroy
Decryptor_Decrypt Decryptor_InvariantNoiseBudget Decryptor_DecryptBatchWords Decryptor_DecryptBatch
CKKSEncoder_Create CKKSEncoder_Destroy CKKSEncoder_SlotCount CKKSEncoder_Encode1 CKKSEncoder_Encode2 CKKSEncoder_Encode3 CKKSEncoder_Encode4 CKKSEncoder_Encode5 CKKSEncoder_Decode1 CKKSEncoder_Decode2
BatchEncoder_Create BatchEncoder_Destroy BatchEncoder_GetSlotCount BatchEncoder_Encode1 BatchEncoder_Encode2 BatchEncoder_Decode1
BatchEncoder_Decode2 BatchEncoder_EncodeDevice BatchEncoder_DecodeDevice
PublicKey_Create PublicKey_Destroy PublicKey_Set PublicKey_UnsafeLoad PublicKey_Load Encryptor_Encrypt Encryptor_EncryptZero1 Encryptor_EncryptZero2 Encryptor_EncryptZeroSymmetric2
Encryptor_Create Encryptor_Destroy Encryptor_SetSeed Encryptor_EncryptZeroSymmetric1 Encryptor_EncryptSymmetric
Encryptor_SymmetricSaveSize Encryptor_EncryptZeroSymmetricSave Encryptor_EncryptSymmetricSave
Plaintext_Create1 Plaintext_Create5 Plaintext_Destroy Plaintext_Set4 Plaintext_SetFromDevice Plaintext_CoeffCount
Plaintext_IsNTTForm Plaintext_GetParmsId Plaintext_SetParmsId Plaintext_Scale Plaintext_SetScale Plaintext_CopyToHost
Plaintext_SaveSize Plaintext_Save Plaintext_UnsafeLoad Plaintext_Load
Evaluator_AddMany Evaluator_AddPlain Evaluator_SubPlain Evaluator_MultiplyMany Evaluator_MultiplyPlain Evaluator_Exponentiate
Evaluator_TransformToNTT1 Evaluator_ModSwitchToNext2 Evaluator_ModSwitchTo2
KSwitchKeys_Create1 KSwitchKeys_Destroy KSwitchKeys_Size KSwitchKeys_SetKey KSwitchKeys_SetKeyFromDevice
KSwitchKeys_SetKeyDigits KSwitchKeys_HasKey KSwitchKeys_DeviceBytes RelinKeys_GetIndex GaloisKeys_GetIndex GaloisTool_GetEltFromStep
Evaluator_Create Evaluator_Destroy Evaluator_SetStream Evaluator_Synchronize Evaluator_CopyTo Evaluator_SetTransparentCheck
Evaluator_BeginCapture Evaluator_EndCapture Evaluator_LaunchGraph Graph_Destroy
Evaluator_Negate Evaluator_Add Evaluator_Sub Evaluator_Multiply Evaluator_Square Evaluator_Relinearize
Evaluator_ModSwitchToNext1 Evaluator_ModSwitchTo1 Evaluator_RescaleToNext Evaluator_RescaleTo
Evaluator_ModReduceToNext Evaluator_ModReduceTo Evaluator_TransformToNTT2 Evaluator_TransformFromNTT Evaluator_ApplyGalois
Evaluator_RotateRows Evaluator_RotateColumns Evaluator_RotateVector Evaluator_ComplexConjugate
Evaluator_ContextUsingKeyswitching
Evaluator_SwitchKeyAccWords Evaluator_RelinearizePartial Evaluator_RelinearizeFinish Evaluator_ApplyGaloisPartial
Evaluator_ApplyGaloisFinish
Comm_GetUniqueId Comm_RcclAvailable Comm_Create Comm_Destroy Comm_Info Comm_DigitRange Comm_AllReduceWords Comm_BroadcastWords
Evaluator_RelinearizeDigitParallel Evaluator_ApplyGaloisDigitParallel Evaluator_RotateVectorDigitParallel
Evaluator_BroadcastKeyDigits Evaluator_SwitchKeySlots Evaluator_SwitchKeyPackTargets Evaluator_SwitchKeyFinishOwned
Evaluator_SwitchKeyAddGathered
SealHip_ReleasePool SealHip_PoolStats SealHip_TailStats SealHip_ProductStats SealHip_GaloisStats SealHip_KsChunkStats SealHip_SetStagedHostCopies SealHip_InstallAbortTrace shl_stream_create shl_stream_destroy shl_device_count shl_set_device
shl_ntt_forward shl_ntt_inverse shl_dyadic_product shl_apply_galois shl_rns_stage shl_malloc shl_free
shl_memcpy_h2d shl_memcpy_d2h shl_device_synchronize shl_timer_create shl_timer_destroy shl_timer_start
shl_timer_stop
Ciphertext_Create4 Ciphertext_Create5 Ciphertext_GetDataAt1 Ciphertext_GetDataAt2 Ciphertext_Release Ciphertext_Reserve1
Ciphertext_Reserve2 Ciphertext_Reserve3 Ciphertext_Resize2 Ciphertext_Resize3 Ciphertext_Resize4 Ciphertext_SetDataAt
Ciphertext_SetParmsId Ciphertext_SizeCapacity ContextData_ChainIndex ContextData_CoeffDivPlainModulus ContextData_Destroy ContextData_NextContextData
ContextData_Parms ContextData_ParmsId ContextData_PlainUpperHalfIncrement ContextData_PlainUpperHalfThreshold ContextData_PrevContextData ContextData_Qualifiers
ContextData_TotalCoeffModulus ContextData_TotalCoeffModulusBitCount ContextData_UpperHalfIncrement ContextData_UpperHalfThreshold EPQ_Destroy EPQ_ParametersSet
EPQ_SecLevel EPQ_UsingBatching EPQ_UsingDescendingModulusChain EPQ_UsingFFT EPQ_UsingFastPlainLift EPQ_UsingNTT
EncParams_Create2 EncParams_Equals EncParams_GetParmsId EncParams_GetPlainModulus EncParams_Load EncParams_Save
EncParams_SaveSize EncParams_Set KSwitchKeys_AddKeyList KSwitchKeys_ClearDataAndReserve KSwitchKeys_Create2 KSwitchKeys_GetKeyList
KSwitchKeys_GetParmsId KSwitchKeys_RawSize KSwitchKeys_Set KSwitchKeys_SetParmsId PublicKey_Assign PublicKey_Create2
PublicKey_ParmsId PublicKey_Save PublicKey_SaveSize SEALContext_FirstContextData SEALContext_GetContextData SEALContext_KeyContextData
SEALContext_LastContextData SEALContext_ParameterErrorMessage SEALContext_ParameterErrorName SEALContext_ParametersSet SecretKey_Assign SecretKey_Create2
SecretKey_ParmsId SecretKey_Save SecretKey_SaveSize
""".split()

_lib = None
_lib_path = None


def load(path=None):
    """Load (once) and return the C-ABI library.  `path` is for tests only."""
    global _lib, _lib_path
    want = os.path.abspath(path or DEFAULT_LIB)
    if _lib is not None and _lib_path == want:
        return _lib
    if not os.path.exists(want):
        raise ImportError(
            "seal_amd: %s not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % want)
    lib = C.CDLL(want)
    for name in SYMBOLS:
        getattr(lib, name).restype = C.c_long
    _lib, _lib_path = lib, want
    return lib


def lib():
    return _lib if _lib is not None else load()


def check(hr):
    hr &= 0xFFFFFFFF
    if hr == S_OK:
        return
    n = C.c_uint64(0)
    lib().SealHip_LastError(None, C.byref(n))
    buf = C.create_string_buffer(max(int(n.value), 1))
    lib().SealHip_LastError(buf, C.byref(n))
    cls = _ERR.get(hr, SealHipError)
    if hr == E_OUTOFMEMORY:
        raise MemoryError(buf.value.decode())
    raise cls(hr, buf.value.decode(errors="replace"))
