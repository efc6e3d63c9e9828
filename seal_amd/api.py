"""Host-side mirror of the reference's object model for the hot path, over the C ABI.

Class and method names follow seal::EncryptionParameters / SEALContext / Ciphertext / RelinKeys /
GaloisKeys / Evaluator (native/src/seal/*.h) so that the parity tests read like the reference's own
(native/tests/seal/evaluator.cpp).  A `Ciphertext` here is a device-resident *batch* of ciphertexts
with shared metadata; numpy arrays cross the boundary as [size][batch][K][N] uint64 slabs
(for batch == 1 that is Ciphertext::data(), ciphertext.h:337-349).
"""
import ctypes as C
import os

import numpy as np

from . import _native as N

SCHEME = {"none": 0, "bfv": 1, "ckks": 2, "bgv": 3}


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class CoeffModulus:
    @staticmethod
    def Create(poly_modulus_degree, bit_sizes):
        """CoeffModulus::Create (modulus.cpp:143-184)."""
        bits = (C.c_int * len(bit_sizes))(*bit_sizes)
        out = np.zeros(len(bit_sizes), dtype=np.uint64)
        N.check(N.lib().CoeffModulus_Create1(C.c_uint64(poly_modulus_degree), C.c_uint64(len(bit_sizes)), bits, _p(out)))
        return [int(x) for x in out]


class PlainModulus:
    @staticmethod
    def Batching(poly_modulus_degree, bit_size):
        v = C.c_uint64()
        N.check(N.lib().PlainModulus_Batching(C.c_uint64(poly_modulus_degree), C.c_int(bit_size), C.byref(v)))
        return v.value


class EncryptionParameters:
    def __init__(self, scheme):
        self.scheme = scheme
        self._h = C.c_void_p()
        N.check(N.lib().EncParams_Create1(C.c_uint8(SCHEME[scheme]), C.byref(self._h)))

    def __del__(self):
        if getattr(self, "_h", None):
            N.lib().EncParams_Destroy(self._h)
            self._h = None

    def set_poly_modulus_degree(self, n):
        N.check(N.lib().EncParams_SetPolyModulusDegree(self._h, C.c_uint64(n)))

    def set_coeff_modulus(self, primes):
        a = np.array(list(primes), dtype=np.uint64)
        N.check(N.lib().EncParams_SetCoeffModulus(self._h, C.c_uint64(len(a)), _p(a)))

    def set_plain_modulus(self, t):
        N.check(N.lib().EncParams_SetPlainModulus2(self._h, C.c_uint64(t)))

    def poly_modulus_degree(self):
        v = C.c_uint64()
        N.check(N.lib().EncParams_GetPolyModulusDegree(self._h, C.byref(v)))
        return v.value

    def coeff_modulus(self):
        n = C.c_uint64()
        N.check(N.lib().EncParams_GetCoeffModulus(self._h, C.byref(n), None))
        out = np.zeros(n.value, dtype=np.uint64)
        N.check(N.lib().EncParams_GetCoeffModulus(self._h, C.byref(n), _p(out)))
        return [int(x) for x in out]

    def plain_modulus(self):
        v = C.c_uint64()
        N.check(N.lib().EncParams_GetPlainModulus(self._h, C.byref(v)))
        return v.value

    def scheme_id(self):
        v = C.c_uint8()
        N.check(N.lib().EncParams_GetScheme(self._h, C.byref(v)))
        return v.value

    def parms_id(self):
        out = (C.c_uint64 * 4)()
        N.check(N.lib().EncParams_GetParmsId(self._h, out))
        return tuple(out)

    def copy(self):
        h = C.c_void_p()
        N.check(N.lib().EncParams_Create2(self._h, C.byref(h)))
        return EncryptionParameters._wrap(h)

    def assign(self, other):
        N.check(N.lib().EncParams_Set(self._h, other._h))
        self.scheme = {v: k for k, v in SCHEME.items()}[self.scheme_id()]

    def equals(self, other):
        b = C.c_bool()
        N.check(N.lib().EncParams_Equals(self._h, other._h, C.byref(b)))
        return b.value

    @staticmethod
    def _wrap(handle):
        p = EncryptionParameters.__new__(EncryptionParameters)
        p._h = handle
        p.scheme = {v: k for k, v in SCHEME.items()}[p.scheme_id()]
        return p

    def save_bytes(self, compr_mode=0):
        """EncryptionParameters::save (compr_mode 0 none, 1 zlib, 2 zstd)"""
        cap = C.c_int64()
        N.check(N.lib().EncParams_SaveSize(self._h, C.c_uint8(compr_mode), C.byref(cap)))
        buf = (C.c_uint8 * cap.value)()
        n = C.c_int64()
        N.check(N.lib().EncParams_Save(self._h, buf, C.c_uint64(cap.value), C.c_uint8(compr_mode), C.byref(n)))
        return C.string_at(buf, n.value)

    def load_bytes(self, data):
        """EncryptionParameters::load; returns the bytes read"""
        data = bytes(data)
        n = C.c_int64()
        N.check(N.lib().EncParams_Load(self._h, C.cast(C.c_char_p(data), C.c_void_p), C.c_uint64(len(data)), C.byref(n)))
        self.scheme = {v: k for k, v in SCHEME.items()}[self.scheme_id()]
        return n.value


class SEALContext:
    """SEALContext(parms, expand_mod_chain, sec_level) — builds and uploads every table."""

    def __init__(self, parms, expand_mod_chain=True, sec_level=0):
        self.parms = parms
        self._h = C.c_void_p()
        N.check(N.lib().SEALContext_Create(parms._h, C.c_bool(expand_mod_chain), C.c_int(sec_level), C.byref(self._h)))
        self.n = parms.poly_modulus_degree()
        self.scheme = parms.scheme

    def __del__(self):
        if getattr(self, "_h", None):
            N.lib().SEALContext_Destroy(self._h)
            self._h = None

    def _pid(self, fn):
        out = (C.c_uint64 * 4)()
        N.check(getattr(N.lib(), fn)(self._h, out))
        return tuple(out)

    def key_parms_id(self):
        return self._pid("SEALContext_KeyParmsId")

    def first_parms_id(self):
        return self._pid("SEALContext_FirstParmsId")

    def last_parms_id(self):
        return self._pid("SEALContext_LastParmsId")

    def using_keyswitching(self):
        b = C.c_bool()
        N.check(N.lib().SEALContext_UsingKeyswitching(self._h, C.byref(b)))
        return b.value

    def chain_index(self, parms_id):
        pid = (C.c_uint64 * 4)(*parms_id)
        v = C.c_uint64()
        N.check(N.lib().SEALContext_ChainIndex(self._h, pid, C.byref(v)))
        return v.value

    def parms_id_at(self, chain_index):
        out = (C.c_uint64 * 4)()
        N.check(N.lib().SEALContext_ParmsIdAt(self._h, C.c_uint64(chain_index), out))
        return tuple(out)

    def set_parms_id(self, chain_index, parms_id):
        pid = (C.c_uint64 * 4)(*parms_id)
        N.check(N.lib().SEALContext_SetParmsId(self._h, C.c_uint64(chain_index), pid))

    def coeff_modulus_at(self, chain_index):
        n = C.c_uint64()
        N.check(N.lib().SEALContext_CoeffModulusAt(self._h, C.c_uint64(chain_index), C.byref(n), None))
        out = np.zeros(n.value, dtype=np.uint64)
        N.check(N.lib().SEALContext_CoeffModulusAt(self._h, C.c_uint64(chain_index), C.byref(n), _p(out)))
        return [int(x) for x in out]

    def total_coeff_modulus_bit_count(self, chain_index):
        v = C.c_int()
        N.check(N.lib().SEALContext_TotalCoeffModulusBitCount(self._h, C.c_uint64(chain_index), C.byref(v)))
        return v.value

    def ntt_root(self, prime_index):
        v = C.c_uint64()
        N.check(N.lib().SEALContext_NTTRoot(self._h, C.c_uint64(prime_index), C.byref(v)))
        return v.value

    def base_bsk(self, chain_index):
        n = C.c_uint64()
        N.check(N.lib().SEALContext_BaseBsk(self._h, C.c_uint64(chain_index), C.byref(n), None))
        out = np.zeros(max(n.value, 1), dtype=np.uint64)
        N.check(N.lib().SEALContext_BaseBsk(self._h, C.c_uint64(chain_index), C.byref(n), _p(out)))
        return [int(x) for x in out[: n.value]]

    def galois_elt_from_step(self, step):
        v = C.c_uint32()
        N.check(N.lib().GaloisTool_GetEltFromStep(self._h, C.c_int(step), C.byref(v)))
        return v.value

    # -- SEALContext::key_context_data / first_context_data / last_context_data / get_context_data (context.h:322-346)
    def _cd(self, fn, *args):
        h = C.c_void_p()
        N.check(getattr(N.lib(), fn)(self._h, *args, C.byref(h)))
        return ContextData(self, h) if h.value else None

    def key_context_data(self):
        return self._cd("SEALContext_KeyContextData")

    def first_context_data(self):
        return self._cd("SEALContext_FirstContextData")

    def last_context_data(self):
        return self._cd("SEALContext_LastContextData")

    def get_context_data(self, parms_id):
        return self._cd("SEALContext_GetContextData", (C.c_uint64 * 4)(*parms_id))

    def parameters_set(self):
        b = C.c_bool()
        N.check(N.lib().SEALContext_ParametersSet(self._h, C.byref(b)))
        return b.value

    def parameter_error(self):
        out = []
        for fn in ("SEALContext_ParameterErrorName", "SEALContext_ParameterErrorMessage"):
            n = C.c_uint64()
            N.check(getattr(N.lib(), fn)(self._h, None, C.byref(n)))
            buf = C.create_string_buffer(n.value + 1)
            N.check(getattr(N.lib(), fn)(self._h, buf, C.byref(n)))
            out.append(buf.value.decode())
        return tuple(out)


class ContextData:
    """SEALContext::ContextData (context.h:181-319): one level of the modulus-switching chain; the handle belongs to the context"""

    def __init__(self, context, handle):
        self.context, self._h = context, handle

    def _words(self, fn):
        n = C.c_uint64(0)
        N.check(getattr(N.lib(), fn)(self._h, C.byref(n), None))
        out = np.zeros(n.value, dtype=np.uint64)
        if n.value:
            N.check(getattr(N.lib(), fn)(self._h, C.byref(n), _p(out)))
        return [int(x) for x in out]

    def _link(self, fn):
        h = C.c_void_p()
        N.check(getattr(N.lib(), fn)(self._h, C.byref(h)))
        return ContextData(self.context, h) if h.value else None

    def chain_index(self):
        v = C.c_uint64()
        N.check(N.lib().ContextData_ChainIndex(self._h, C.byref(v)))
        return v.value

    def parms_id(self):
        out = (C.c_uint64 * 4)()
        N.check(N.lib().ContextData_ParmsId(self._h, out))
        return tuple(out)

    def parms(self):
        h = C.c_void_p()
        N.check(N.lib().ContextData_Parms(self._h, C.byref(h)))
        return EncryptionParameters._wrap(h)

    def qualifiers(self):
        h = C.c_void_p()
        N.check(N.lib().ContextData_Qualifiers(self._h, C.byref(h)))
        out = {}
        try:
            for key, fn, ctype in (("parameters_set", "EPQ_ParametersSet", C.c_bool), ("using_fft", "EPQ_UsingFFT", C.c_bool),
                                   ("using_ntt", "EPQ_UsingNTT", C.c_bool), ("using_batching", "EPQ_UsingBatching", C.c_bool),
                                   ("using_fast_plain_lift", "EPQ_UsingFastPlainLift", C.c_bool),
                                   ("using_descending_modulus_chain", "EPQ_UsingDescendingModulusChain", C.c_bool),
                                   ("sec_level", "EPQ_SecLevel", C.c_int)):
                v = ctype()
                N.check(getattr(N.lib(), fn)(h, C.byref(v)))
                out[key] = int(v.value)
        finally:
            N.lib().EPQ_Destroy(h)
        return out

    def total_coeff_modulus(self):
        return self._words("ContextData_TotalCoeffModulus")

    def total_coeff_modulus_bit_count(self):
        v = C.c_int()
        N.check(N.lib().ContextData_TotalCoeffModulusBitCount(self._h, C.byref(v)))
        return v.value

    def coeff_div_plain_modulus(self):
        return self._words("ContextData_CoeffDivPlainModulus")

    def plain_upper_half_threshold(self):
        v = C.c_uint64()
        N.check(N.lib().ContextData_PlainUpperHalfThreshold(self._h, C.byref(v)))
        return v.value

    def plain_upper_half_increment(self):
        return self._words("ContextData_PlainUpperHalfIncrement")

    def upper_half_threshold(self):
        return self._words("ContextData_UpperHalfThreshold")

    def upper_half_increment(self):
        return self._words("ContextData_UpperHalfIncrement")

    def prev_context_data(self):
        return self._link("ContextData_PrevContextData")

    def next_context_data(self):
        return self._link("ContextData_NextContextData")


class Ciphertext:
    def __init__(self, context, batch=1, _copy_of=None):
        self.context = context
        self._h = C.c_void_p()
        if _copy_of is not None:
            N.check(N.lib().Ciphertext_Create2(_copy_of._h, C.byref(self._h)))
        else:
            N.check(N.lib().Ciphertext_CreateBatch(context._h, C.c_uint64(batch), C.byref(self._h)))

    def __del__(self):
        if getattr(self, "_h", None):
            N.lib().Ciphertext_Destroy(self._h)
            self._h = None

    def copy(self):
        return Ciphertext(self.context, _copy_of=self)

    # -- metadata
    def _get(self, fn, ctype):
        v = ctype()
        N.check(getattr(N.lib(), fn)(self._h, C.byref(v)))
        return v.value

    def size(self):
        return self._get("Ciphertext_Size", C.c_uint64)

    def batch(self):
        return self._get("Ciphertext_BatchCount", C.c_uint64)

    def coeff_modulus_size(self):
        return self._get("Ciphertext_CoeffModulusSize", C.c_uint64)

    def poly_modulus_degree(self):
        return self._get("Ciphertext_PolyModulusDegree", C.c_uint64)

    def is_ntt_form(self):
        return self._get("Ciphertext_IsNTTForm", C.c_bool)

    def scale(self):
        return self._get("Ciphertext_Scale", C.c_double)

    def correction_factor(self):
        return self._get("Ciphertext_CorrectionFactor", C.c_uint64)

    def parms_id(self):
        out = (C.c_uint64 * 4)()
        N.check(N.lib().Ciphertext_ParmsId(self._h, out))
        return tuple(out)

    def chain_index(self):
        return self.context.chain_index(self.parms_id())

    def set_is_ntt_form(self, v):
        N.check(N.lib().Ciphertext_SetIsNTTForm(self._h, C.c_bool(v)))

    def set_scale(self, v):
        N.check(N.lib().Ciphertext_SetScale(self._h, C.c_double(v)))

    def set_correction_factor(self, v):
        N.check(N.lib().Ciphertext_SetCorrectionFactor(self._h, C.c_uint64(v)))

    def is_transparent(self):
        return self._get("Ciphertext_IsTransparent", C.c_bool)

    def resize(self, parms_id, size):
        pid = (C.c_uint64 * 4)(*parms_id)
        N.check(N.lib().Ciphertext_Resize1(self._h, self.context._h, pid, C.c_uint64(size)))

    # -- Ciphertext::reserve / resize / size_capacity / operator[] / release (ciphertext.h:154-592)
    def reserve(self, size_capacity, parms_id=None, with_context=True):
        if parms_id is not None:
            N.check(N.lib().Ciphertext_Reserve1(self._h, self.context._h, (C.c_uint64 * 4)(*parms_id), C.c_uint64(size_capacity)))
        elif with_context:
            N.check(N.lib().Ciphertext_Reserve2(self._h, self.context._h, C.c_uint64(size_capacity)))
        else:
            N.check(N.lib().Ciphertext_Reserve3(self._h, C.c_uint64(size_capacity)))

    def resize_same_level(self, size, with_context=True):
        if with_context:
            N.check(N.lib().Ciphertext_Resize2(self._h, self.context._h, C.c_uint64(size)))
        else:
            N.check(N.lib().Ciphertext_Resize3(self._h, C.c_uint64(size)))

    def resize_geometry(self, size, poly_modulus_degree, coeff_modulus_size):
        N.check(N.lib().Ciphertext_Resize4(self._h, C.c_uint64(size), C.c_uint64(poly_modulus_degree), C.c_uint64(coeff_modulus_size)))

    def size_capacity(self):
        return self._get("Ciphertext_SizeCapacity", C.c_uint64)

    def set_parms_id(self, parms_id):
        N.check(N.lib().Ciphertext_SetParmsId(self._h, (C.c_uint64 * 4)(*parms_id)))

    def get_data_at(self, index, coeff_index=None):
        v = C.c_uint64()
        if coeff_index is None:
            N.check(N.lib().Ciphertext_GetDataAt1(self._h, C.c_uint64(index), C.byref(v)))
        else:
            N.check(N.lib().Ciphertext_GetDataAt2(self._h, C.c_uint64(index), C.c_uint64(coeff_index), C.byref(v)))
        return v.value

    def set_data_at(self, index, value):
        N.check(N.lib().Ciphertext_SetDataAt(self._h, C.c_uint64(index), C.c_uint64(value)))

    def release(self):
        N.check(N.lib().Ciphertext_Release(self._h))

    @staticmethod
    def with_parms_id(context, parms_id, capacity=None):
        """Ciphertext(context, parms_id[, size_capacity]) (ciphertext.h:128-149)"""
        ct = Ciphertext.__new__(Ciphertext)
        ct.context, ct._h = context, C.c_void_p()
        pid = (C.c_uint64 * 4)(*parms_id)
        if capacity is None:
            N.check(N.lib().Ciphertext_Create4(context._h, pid, None, C.byref(ct._h)))
        else:
            N.check(N.lib().Ciphertext_Create5(context._h, pid, C.c_uint64(capacity), None, C.byref(ct._h)))
        return ct

    # -- data
    def shape(self):
        return (self.size(), self.batch(), self.coeff_modulus_size(), self.poly_modulus_degree())

    def device_ptr(self):
        ptr = C.c_void_p()
        words = C.c_uint64()
        N.check(N.lib().Ciphertext_DevicePtr(self._h, C.byref(ptr), C.byref(words)))
        return ptr.value, words.value

    def load(self, array):
        """Host -> device.  array: uint64 [size][batch][K][N] (or [size][K][N] for batch == 1)."""
        a = np.ascontiguousarray(array, dtype=np.uint64)
        N.check(N.lib().Ciphertext_CopyFromHost(self._h, _p(a), C.c_uint64(a.size)))

    def load_device(self, device_ptr, words, stream=None):
        N.check(N.lib().Ciphertext_CopyFromDevice(self._h, C.c_void_p(device_ptr), C.c_uint64(words), C.c_void_p(stream or 0)))

    def to_numpy(self):
        out = np.zeros(self.shape(), dtype=np.uint64)
        if out.size:
            N.check(N.lib().Ciphertext_CopyToHost(self._h, _p(out), C.c_uint64(out.size)))
        return out

    def item_to_numpy(self, item):
        """device -> host copy of ONE batch item: uint64 [size][K][N] (the evaluator's stream is drained first)"""
        size, batch, K, n = self.shape()
        if not 0 <= item < batch:
            raise IndexError("batch item")
        out = np.zeros((size, K, n), dtype=np.uint64)
        ptr, _ = self.device_ptr()
        N.check(N.lib().shl_device_synchronize())
        for p in range(size):
            src = C.c_void_p(ptr + ((p * batch + item) * K * n) * 8)
            N.check(N.lib().shl_memcpy_d2h(_p(out[p]), src, C.c_uint64(K * n * 8)))
        return out

    # -- the reference's wire format (Ciphertext::save / load / unsafe_load, ciphertext.cpp:153-403)
    def load_bytes(self, data, unsafe=False, item=None):
        """seal::Ciphertext::load (unsafe=True: unsafe_load) of a serialized stream; item = slot of a batch.  Returns bytes read."""
        data = bytes(data)
        buf = C.cast(C.c_char_p(data), C.c_void_p)  # the stream is parsed in place
        n = C.c_int64()
        if item is not None:
            N.check(N.lib().Ciphertext_LoadItem(self._h, self.context._h, C.c_uint64(item), buf, C.c_uint64(len(data)), C.byref(n)))
        else:
            fn = N.lib().Ciphertext_UnsafeLoad if unsafe else N.lib().Ciphertext_Load
            N.check(fn(self._h, self.context._h, buf, C.c_uint64(len(data)), C.byref(n)))
        return n.value

    def save_size(self, compr_mode=0):
        return self._get2("Ciphertext_SaveSize", C.c_uint8(compr_mode), C.c_int64)

    def _get2(self, fn, arg, ctype):
        v = ctype()
        N.check(getattr(N.lib(), fn)(self._h, arg, C.byref(v)))
        return v.value

    def save_bytes(self, item=None, compr_mode=0):
        """seal::Ciphertext::save(compr_mode_type::none) of the ciphertext (or of slot `item` of a batch)"""
        cap = self.save_size(compr_mode)
        buf = (C.c_uint8 * cap)()
        n = C.c_int64()
        if item is not None:
            N.check(N.lib().Ciphertext_SaveItem(self._h, C.c_uint64(item), buf, C.c_uint64(cap), C.c_uint8(compr_mode), C.byref(n)))
        else:
            N.check(N.lib().Ciphertext_Save(self._h, buf, C.c_uint64(cap), C.c_uint8(compr_mode), C.byref(n)))
        return C.string_at(buf, n.value)

    @staticmethod
    def from_numpy(context, array, parms_id, is_ntt_form, scale=1.0, correction_factor=1):
        a = np.ascontiguousarray(array, dtype=np.uint64)
        if a.ndim == 3:
            a = a[:, None, :, :]
        ct = Ciphertext(context, batch=a.shape[1])
        ct.resize(parms_id, a.shape[0])
        ct.set_is_ntt_form(is_ntt_form)
        ct.set_scale(scale)
        ct.set_correction_factor(correction_factor)
        ct.load(a)
        return ct


class Plaintext:
    """seal::Plaintext resident in HBM (sealhip.h): coefficient form (<= N coefficients mod t) or NTT form (K*N words)."""

    def __init__(self, context, _copy_of=None):
        self.context = context
        self._h = C.c_void_p()
        if _copy_of is not None:
            N.check(N.lib().Plaintext_Create5(_copy_of._h, C.byref(self._h)))
        else:
            N.check(N.lib().Plaintext_Create1(context._h, C.byref(self._h)))

    def __del__(self):
        if getattr(self, "_h", None):
            N.lib().Plaintext_Destroy(self._h)
            self._h = None

    def copy(self):
        return Plaintext(self.context, _copy_of=self)

    @staticmethod
    def from_numpy(context, array, parms_id=None, scale=1.0):
        """coefficients (1-D, coefficient form) or [K][N] residues with the level's parms_id (NTT form)"""
        p = Plaintext(context)
        a = np.ascontiguousarray(array, dtype=np.uint64).reshape(-1)
        N.check(N.lib().Plaintext_Set4(p._h, C.c_uint64(a.size), _p(a)))
        if parms_id is not None:
            p.set_parms_id(parms_id)
        p.set_scale(scale)
        return p

    def coeff_count(self):
        v = C.c_uint64()
        N.check(N.lib().Plaintext_CoeffCount(self._h, C.byref(v)))
        return v.value

    def is_ntt_form(self):
        v = C.c_bool()
        N.check(N.lib().Plaintext_IsNTTForm(self._h, C.byref(v)))
        return v.value

    def parms_id(self):
        pid = (C.c_uint64 * 4)()
        N.check(N.lib().Plaintext_GetParmsId(self._h, pid))
        return tuple(pid)

    def set_parms_id(self, parms_id):
        pid = (C.c_uint64 * 4)(*parms_id)
        N.check(N.lib().Plaintext_SetParmsId(self._h, pid))

    def scale(self):
        v = C.c_double()
        N.check(N.lib().Plaintext_Scale(self._h, C.byref(v)))
        return v.value

    def set_scale(self, v):
        N.check(N.lib().Plaintext_SetScale(self._h, C.c_double(v)))

    def to_numpy(self):
        out = np.zeros(self.coeff_count(), dtype=np.uint64)
        N.check(N.lib().Plaintext_CopyToHost(self._h, _p(out), C.c_uint64(out.size)))
        return out

    # -- the reference's wire format (Plaintext::save / load / unsafe_load)
    def load_bytes(self, data, unsafe=False):
        data = bytes(data)
        buf = C.cast(C.c_char_p(data), C.c_void_p)
        n = C.c_int64()
        fn = N.lib().Plaintext_UnsafeLoad if unsafe else N.lib().Plaintext_Load
        N.check(fn(self._h, self.context._h, buf, C.c_uint64(len(data)), C.byref(n)))
        return n.value

    def save_bytes(self, compr_mode=0):
        """Plaintext::save; compr_mode 0 none, 1 zlib, 2 zstd"""
        cap = C.c_int64()
        N.check(N.lib().Plaintext_SaveSize(self._h, C.c_uint8(compr_mode), C.byref(cap)))
        buf = (C.c_uint8 * cap.value)()
        n = C.c_int64()
        N.check(N.lib().Plaintext_Save(self._h, buf, C.c_uint64(cap.value), C.c_uint8(compr_mode), C.byref(n)))
        return C.string_at(buf, n.value)


class KSwitchKeys:
    """Device-resident key-switching keys; slab per index: [digits][2][L][N] (kswitchkeys.h:340)."""

    def __init__(self, context):
        self.context = context
        self._h = C.c_void_p()
        N.check(N.lib().KSwitchKeys_Create1(C.byref(self._h)))

    def __del__(self):
        if getattr(self, "_h", None):
            N.lib().KSwitchKeys_Destroy(self._h)
            self._h = None

    def set_key(self, index, array):
        a = np.ascontiguousarray(array, dtype=np.uint64)
        N.check(N.lib().KSwitchKeys_SetKey(self._h, self.context._h, C.c_uint64(index), C.c_uint64(a.shape[0]), _p(a)))

    def set_key_digits(self, index, digit_first, array):
        """upload only the digits [digit_first, digit_first + len(array)) of key `index` (digit-parallel key switching)"""
        a = np.ascontiguousarray(array, dtype=np.uint64)
        N.check(N.lib().KSwitchKeys_SetKeyDigits(self._h, self.context._h, C.c_uint64(index), C.c_uint64(digit_first),
                                                 C.c_uint64(a.shape[0]), _p(a)))

    def set_key_device(self, index, digits, device_ptr):
        N.check(N.lib().KSwitchKeys_SetKeyFromDevice(self._h, self.context._h, C.c_uint64(index), C.c_uint64(digits),
                                                     C.c_void_p(device_ptr)))

    def load_bytes(self, data, unsafe=False):
        """KSwitchKeys::load / unsafe_load of a serialized RelinKeys / GaloisKeys stream (seeded or full).  Returns bytes read."""
        data = bytes(data)
        buf = C.cast(C.c_char_p(data), C.c_void_p)  # the stream is parsed in place
        n = C.c_int64()
        fn = N.lib().KSwitchKeys_UnsafeLoad if unsafe else N.lib().KSwitchKeys_Load
        N.check(fn(self._h, self.context._h, buf, C.c_uint64(len(data)), C.byref(n)))
        return n.value

    def save_bytes(self, compr_mode=0):
        """KSwitchKeys::save of the full keys (compr_mode 0 none, 1 zlib, 2 zstd)"""
        cap = C.c_int64()
        N.check(N.lib().KSwitchKeys_SaveSize(self._h, C.c_uint8(compr_mode), C.byref(cap)))
        buf = (C.c_uint8 * cap.value)()
        n = C.c_int64()
        N.check(N.lib().KSwitchKeys_Save(self._h, buf, C.c_uint64(cap.value), C.c_uint8(compr_mode), C.byref(n)))
        return C.string_at(buf, n.value)

    def has_index(self, index):
        b = C.c_bool()
        N.check(N.lib().KSwitchKeys_HasKey(self._h, C.c_uint64(index), C.byref(b)))
        return b.value

    def copy(self):
        """KSwitchKeys(copy): every key slab duplicated in HBM"""
        k = type(self).__new__(type(self))
        k.context, k._h = self.context, C.c_void_p()
        N.check(N.lib().KSwitchKeys_Create2(self._h, C.byref(k._h)))
        return k

    def assign(self, other):
        N.check(N.lib().KSwitchKeys_Set(self._h, other._h))

    def device_bytes(self):
        """HBM held by the keys of this object (sealhip.h: KSwitchKeys_DeviceBytes)"""
        v = C.c_uint64()
        N.check(N.lib().KSwitchKeys_DeviceBytes(self._h, C.byref(v)))
        return v.value

    def raw_size(self):
        v = C.c_uint64()
        N.check(N.lib().KSwitchKeys_RawSize(self._h, C.byref(v)))
        return v.value

    def key_list(self, index):
        """the digits of key `index` as PublicKey objects (KSwitchKeys::data()[index]); copies, owned by the caller"""
        n = C.c_uint64()
        N.check(N.lib().KSwitchKeys_GetKeyList(self._h, C.c_uint64(index), C.byref(n), None))
        handles = (C.c_void_p * max(1, n.value))()
        N.check(N.lib().KSwitchKeys_GetKeyList(self._h, C.c_uint64(index), C.byref(n), handles))
        out = []
        for i in range(n.value):
            pk = PublicKey.__new__(PublicKey)
            pk.context, pk._h = self.context, C.c_void_p(handles[i])
            out.append(pk)
        return out

    def add_key_list(self, public_keys):
        handles = (C.c_void_p * max(1, len(public_keys)))(*[pk._h.value for pk in public_keys])
        N.check(N.lib().KSwitchKeys_AddKeyList(self._h, C.c_uint64(len(public_keys)), handles))

    def clear_data_and_reserve(self, size):
        N.check(N.lib().KSwitchKeys_ClearDataAndReserve(self._h, C.c_uint64(size)))

    def parms_id(self):
        out = (C.c_uint64 * 4)()
        N.check(N.lib().KSwitchKeys_GetParmsId(self._h, out))
        return tuple(out)

    def set_parms_id(self, parms_id):
        N.check(N.lib().KSwitchKeys_SetParmsId(self._h, (C.c_uint64 * 4)(*parms_id)))

    def size(self):
        v = C.c_uint64()
        N.check(N.lib().KSwitchKeys_Size(self._h, C.byref(v)))
        return v.value


class RelinKeys(KSwitchKeys):
    @staticmethod
    def get_index(key_power):
        v = C.c_uint64()
        N.check(N.lib().RelinKeys_GetIndex(C.c_uint64(key_power), C.byref(v)))
        return v.value

    def has_key(self, key_power):
        return self.has_index(self.get_index(key_power))


class GaloisKeys(KSwitchKeys):
    @staticmethod
    def get_index(galois_elt):
        v = C.c_uint64()
        N.check(N.lib().GaloisKeys_GetIndex(C.c_uint32(galois_elt), C.byref(v)))
        return v.value

    def has_key(self, galois_elt):
        return self.has_index(self.get_index(galois_elt))


class SecretKey:
    """seal::SecretKey resident in HBM: [L][N] words, key level, NTT form"""

    def __init__(self, context, words=None):
        self.context = context
        self._h = C.c_void_p()
        N.check(N.lib().SecretKey_Create(context._h, C.byref(self._h)))
        if words is not None:
            self.set(words)

    def __del__(self):
        if getattr(self, "_h", None):
            N.lib().SecretKey_Destroy(self._h)
            self._h = None

    def set(self, words):
        a = np.ascontiguousarray(words, dtype=np.uint64)
        N.check(N.lib().SecretKey_Set(self._h, _p(a), C.c_uint64(a.size)))

    def words(self, L, n):
        out = np.empty((L, n), dtype=np.uint64)
        N.check(N.lib().SecretKey_Get(self._h, _p(out)))
        return out

    def save_bytes(self, compr_mode=0):
        """SecretKey::save (compr_mode 0 none, 1 zlib, 2 zstd)"""
        cap = C.c_int64()
        N.check(N.lib().SecretKey_SaveSize(self._h, C.c_uint8(compr_mode), C.byref(cap)))
        buf = (C.c_uint8 * cap.value)()
        n = C.c_int64()
        N.check(N.lib().SecretKey_Save(self._h, buf, C.c_uint64(cap.value), C.c_uint8(compr_mode), C.byref(n)))
        return C.string_at(buf, n.value)

    def parms_id(self):
        out = (C.c_uint64 * 4)()
        N.check(N.lib().SecretKey_ParmsId(self._h, out))
        return tuple(out)

    def copy(self):
        k = SecretKey.__new__(SecretKey)
        k.context, k._h = self.context, C.c_void_p()
        N.check(N.lib().SecretKey_Create2(self._h, C.byref(k._h)))
        return k

    def assign(self, other):
        N.check(N.lib().SecretKey_Assign(self._h, other._h))

    def load_bytes(self, data, unsafe=False):
        data = bytes(data)
        n = C.c_int64()
        fn = N.lib().SecretKey_UnsafeLoad if unsafe else N.lib().SecretKey_Load
        N.check(fn(self._h, self.context._h, C.cast(C.c_char_p(data), C.c_void_p), C.c_uint64(len(data)), C.byref(n)))
        return n.value


class Decryptor:
    """seal::Decryptor on the device (sealhip.h): decrypt (batch of one -> Plaintext) and decrypt_batch (raw device words)"""

    def __init__(self, context, secret_key):
        self.context = context
        self._h = C.c_void_p()
        N.check(N.lib().Decryptor_Create(context._h, secret_key._h, C.byref(self._h)))

    def __del__(self):
        if getattr(self, "_h", None):
            N.lib().Decryptor_Destroy(self._h)
            self._h = None

    def decrypt(self, encrypted, destination=None):
        destination = destination if destination is not None else Plaintext(self.context)
        N.check(N.lib().Decryptor_Decrypt(self._h, encrypted._h, destination._h))
        return destination

    def invariant_noise_budget(self, encrypted):
        v = C.c_int()
        N.check(N.lib().Decryptor_InvariantNoiseBudget(self._h, encrypted._h, C.byref(v)))
        return v.value

    def decrypt_batch(self, encrypted, out=None):
        """-> DeviceBuffer of [batch][K][N] (CKKS) or [batch][N] (BFV / BGV) words and its word count; `out` reuses a buffer"""
        w = C.c_uint64()
        N.check(N.lib().Decryptor_DecryptBatchWords(self._h, encrypted._h, C.byref(w)))
        buf = out if out is not None else DeviceBuffer(w.value)
        N.check(N.lib().Decryptor_DecryptBatch(self._h, encrypted._h, C.c_void_p(buf.ptr), w))
        return buf, w.value


class CKKSEncoder:
    """seal::CKKSEncoder on the device (sealhip.h)"""

    def __init__(self, context):
        self.context = context
        self._h = C.c_void_p()
        N.check(N.lib().CKKSEncoder_Create(context._h, C.byref(self._h)))

    def __del__(self):
        if getattr(self, "_h", None):
            N.lib().CKKSEncoder_Destroy(self._h)
            self._h = None

    def slot_count(self):
        v = C.c_uint64()
        N.check(N.lib().CKKSEncoder_SlotCount(self._h, C.byref(v)))
        return v.value

    def encode(self, values, parms_id, scale, destination=None):
        destination = destination if destination is not None else Plaintext(self.context)
        pid = (C.c_uint64 * 4)(*parms_id)
        if isinstance(values, (int, np.integer)) and scale is None:
            N.check(N.lib().CKKSEncoder_Encode5(self._h, C.c_int64(int(values)), pid, destination._h))
            return destination
        if isinstance(values, (float, int, np.floating, np.integer)):
            N.check(N.lib().CKKSEncoder_Encode3(self._h, C.c_double(float(values)), pid, C.c_double(scale), destination._h, None))
            return destination
        if isinstance(values, (complex, np.complexfloating)):   # one complex value in every slot (c/ckksencoder.h:35)
            N.check(N.lib().CKKSEncoder_Encode4(self._h, C.c_double(values.real), C.c_double(values.imag), pid, C.c_double(scale), destination._h, None))
            return destination
        a = np.asarray(values)
        if np.iscomplexobj(a):
            a = np.ascontiguousarray(a, dtype=np.complex128)
            N.check(N.lib().CKKSEncoder_Encode2(self._h, C.c_uint64(a.size), _p(a), pid, C.c_double(scale), destination._h, None))
        else:
            a = np.ascontiguousarray(a, dtype=np.float64)
            N.check(N.lib().CKKSEncoder_Encode1(self._h, C.c_uint64(a.size), _p(a), pid, C.c_double(scale), destination._h, None))
        return destination

    def decode(self, plain, complex_values=False):
        n = C.c_uint64()
        out = np.zeros(self.slot_count(), dtype=np.complex128 if complex_values else np.float64)
        fn = N.lib().CKKSEncoder_Decode2 if complex_values else N.lib().CKKSEncoder_Decode1
        N.check(fn(self._h, plain._h, C.byref(n), _p(out), None))
        return out


class PublicKey:
    """seal::PublicKey resident in HBM: [2][L][N] words, key level, NTT form"""

    def __init__(self, context, words=None):
        self.context = context
        self._h = C.c_void_p()
        N.check(N.lib().PublicKey_Create(context._h, C.byref(self._h)))
        if words is not None:
            a = np.ascontiguousarray(words, dtype=np.uint64)
            N.check(N.lib().PublicKey_Set(self._h, _p(a), C.c_uint64(a.size)))

    def __del__(self):
        if getattr(self, "_h", None):
            N.lib().PublicKey_Destroy(self._h)
            self._h = None

    def words(self, L, n):
        out = np.empty((2, L, n), dtype=np.uint64)
        N.check(N.lib().PublicKey_Get(self._h, _p(out)))
        return out

    def save_bytes(self, compr_mode=0):
        """PublicKey::save (compr_mode 0 none, 1 zlib, 2 zstd)"""
        cap = C.c_int64()
        N.check(N.lib().PublicKey_SaveSize(self._h, C.c_uint8(compr_mode), C.byref(cap)))
        buf = (C.c_uint8 * cap.value)()
        n = C.c_int64()
        N.check(N.lib().PublicKey_Save(self._h, buf, C.c_uint64(cap.value), C.c_uint8(compr_mode), C.byref(n)))
        return C.string_at(buf, n.value)

    def parms_id(self):
        out = (C.c_uint64 * 4)()
        N.check(N.lib().PublicKey_ParmsId(self._h, out))
        return tuple(out)

    def copy(self):
        k = PublicKey.__new__(PublicKey)
        k.context, k._h = self.context, C.c_void_p()
        N.check(N.lib().PublicKey_Create2(self._h, C.byref(k._h)))
        return k

    def assign(self, other):
        N.check(N.lib().PublicKey_Assign(self._h, other._h))

    def load_bytes(self, data, unsafe=False):
        data = bytes(data)
        n = C.c_int64()
        fn = N.lib().PublicKey_UnsafeLoad if unsafe else N.lib().PublicKey_Load
        N.check(fn(self._h, self.context._h, C.cast(C.c_char_p(data), C.c_void_p), C.c_uint64(len(data)), C.byref(n)))
        return n.value


class KeyGenerator:
    """seal::KeyGenerator on the device (sealhip.h): seed = None -> operating-system entropy (the only secure choice).
    seed = 8 words installs the reference's seeded Blake2xbPRNGFactory, INSECURE and for parity tests only: every sampling
    call restarts from the same seed, so all key digits share (a, e) and the saved public seed is the head of the stream
    that sampled the secret key."""

    def __init__(self, context, secret_key=None, seed=None):
        self.context = context
        self._h = C.c_void_p()
        self._seed = None if seed is None else np.ascontiguousarray(seed, dtype=np.uint64)
        sp = None if self._seed is None else _p(self._seed)
        if secret_key is None:
            N.check(N.lib().KeyGenerator_Create1(context._h, sp, C.byref(self._h)))
        else:
            N.check(N.lib().KeyGenerator_Create2(context._h, secret_key._h, sp, C.byref(self._h)))

    def __del__(self):
        if getattr(self, "_h", None):
            N.lib().KeyGenerator_Destroy(self._h)
            self._h = None

    def secret_key(self):
        sk = SecretKey(self.context)
        N.check(N.lib().KeyGenerator_SecretKey(self._h, sk._h))
        return sk

    def create_public_key(self):
        pk = PublicKey(self.context)
        N.check(N.lib().KeyGenerator_CreatePublicKey(self._h, pk._h))
        return pk

    def create_relin_keys(self):
        rlk = RelinKeys(self.context)
        N.check(N.lib().KeyGenerator_CreateRelinKeys(self._h, rlk._h))
        return rlk

    def create_galois_keys(self, galois_elts=None, steps=None):
        """keys for the given Galois elements, or for the given rotation steps, or (neither) for all the elements
        GaloisTool::get_elts_all lists"""
        glk = GaloisKeys(self.context)
        if galois_elts is not None:
            e = np.ascontiguousarray(galois_elts, dtype=np.uint32)
            N.check(N.lib().KeyGenerator_CreateGaloisKeysFromElts(self._h, C.c_uint64(e.size), e.ctypes.data_as(C.c_void_p), glk._h))
        elif steps is not None:
            st = np.ascontiguousarray(steps, dtype=np.int32)
            N.check(N.lib().KeyGenerator_CreateGaloisKeysFromSteps(self._h, C.c_uint64(st.size), st.ctypes.data_as(C.c_void_p), glk._h))
        else:
            N.check(N.lib().KeyGenerator_CreateGaloisKeysAll(self._h, glk._h))
        return glk

    def save_seeded(self, galois_elts=None):
        """the seeded stream of fresh RelinKeys (galois_elts None) or GaloisKeys for the elements: Serializable<...>::save"""
        galois = galois_elts is not None
        e = np.ascontiguousarray(galois_elts if galois else [], dtype=np.uint32)
        cap = C.c_int64()
        N.check(N.lib().KeyGenerator_SeededSaveSize(self._h, C.c_bool(galois), C.c_uint64(len(set(e.tolist())) if galois else 1), C.byref(cap)))
        buf = C.create_string_buffer(cap.value)
        n = C.c_int64()
        if galois:
            N.check(N.lib().KeyGenerator_CreateGaloisKeysFromEltsSave(self._h, C.c_uint64(e.size), e.ctypes.data_as(C.c_void_p),
                                                                      C.cast(buf, C.c_void_p), C.c_uint64(cap.value), C.byref(n)))
        else:
            N.check(N.lib().KeyGenerator_CreateRelinKeysSave(self._h, C.cast(buf, C.c_void_p), C.c_uint64(cap.value), C.byref(n)))
        return C.string_at(buf, n.value)

    def key_words(self, galois_elt, digits, L, n):
        """one key as the reference lays it out: [digits][2][L][N] (galois_elt 0 = the relinearization key)"""
        out = np.empty((digits, 2, L, n), dtype=np.uint64)
        N.check(N.lib().KeyGenerator_KeyToHost(self._h, C.c_uint32(galois_elt), _p(out), C.c_uint64(out.size)))
        return out


class BatchEncoder:
    """seal::BatchEncoder on the device (sealhip.h): N integers modulo t <-> one plaintext polynomial"""

    def __init__(self, context):
        self.context = context
        self._h = C.c_void_p()
        N.check(N.lib().BatchEncoder_Create(context._h, C.byref(self._h)))

    def __del__(self):
        if getattr(self, "_h", None):
            N.lib().BatchEncoder_Destroy(self._h)
            self._h = None

    def slot_count(self):
        v = C.c_uint64()
        N.check(N.lib().BatchEncoder_GetSlotCount(self._h, C.byref(v)))
        return v.value

    def encode(self, values, destination=None, signed=False):
        destination = destination if destination is not None else Plaintext(self.context)
        a = np.ascontiguousarray(values, dtype=np.int64 if signed else np.uint64)
        fn = N.lib().BatchEncoder_Encode2 if signed else N.lib().BatchEncoder_Encode1
        N.check(fn(self._h, C.c_uint64(a.size), _p(a), destination._h))
        return destination

    def decode(self, plain, signed=False):
        out = np.zeros(self.slot_count(), dtype=np.int64 if signed else np.uint64)
        n = C.c_uint64()
        fn = N.lib().BatchEncoder_Decode2 if signed else N.lib().BatchEncoder_Decode1
        N.check(fn(self._h, plain._h, C.byref(n), _p(out), None))
        return out

    def decode_device(self, coefficients, batch, signed=False, out=None):
        """coefficients: DeviceBuffer [batch][N] (e.g. from Decryptor.decrypt_batch) -> DeviceBuffer of slot values"""
        out = out if out is not None else DeviceBuffer(batch * self.slot_count())
        N.check(N.lib().BatchEncoder_DecodeDevice(self._h, C.c_void_p(coefficients.ptr), C.c_uint64(batch), C.c_bool(signed), C.c_void_p(out.ptr)))
        return out

    def encode_device(self, values, batch, signed=False, out=None):
        out = out if out is not None else DeviceBuffer(batch * self.slot_count())
        N.check(N.lib().BatchEncoder_EncodeDevice(self._h, C.c_void_p(values.ptr), C.c_uint64(batch), C.c_bool(signed), C.c_void_p(out.ptr)))
        return out


class Encryptor:
    """seal::Encryptor, secret-key half (sealhip.h): encrypt_symmetric / encrypt_zero_symmetric and their seeded streams"""

    def __init__(self, context, secret_key=None, seed=None, public_key=None):
        self.context = context
        self._h = C.c_void_p()
        N.check(N.lib().Encryptor_Create(context._h, public_key._h if public_key is not None else None,
                                         secret_key._h if secret_key is not None else None, C.byref(self._h)))
        if seed is not None:
            self.set_seed(seed)

    def __del__(self):
        if getattr(self, "_h", None):
            N.lib().Encryptor_Destroy(self._h)
            self._h = None

    def set_seed(self, seed):
        """the reference's seeded Blake2xbPRNGFactory: 8 words (or one int = first word); None -> operating-system entropy.
        INSECURE, parity tests only: every encryption restarts from the same seed (identical (a, e) for every call)."""
        if seed is None:
            N.check(N.lib().Encryptor_SetSeed(self._h, None))
            return
        words = [seed] + [0] * 7 if isinstance(seed, int) else list(seed)
        N.check(N.lib().Encryptor_SetSeed(self._h, (C.c_uint64 * 8)(*words)))

    def encrypt(self, plain, destination=None):
        """Encryptor::encrypt (public key)"""
        destination = destination if destination is not None else Ciphertext(self.context)
        N.check(N.lib().Encryptor_Encrypt(self._h, plain._h, destination._h, None))
        return destination

    def encrypt_zero(self, parms_id=None, destination=None):
        """parms_id None: at the first data level (Encryptor_EncryptZero2, c/encryptor.h:26)"""
        destination = destination if destination is not None else Ciphertext(self.context)
        if parms_id is None:
            N.check(N.lib().Encryptor_EncryptZero2(self._h, destination._h, None))
        else:
            N.check(N.lib().Encryptor_EncryptZero1(self._h, (C.c_uint64 * 4)(*parms_id), destination._h, None))
        return destination

    def encrypt_zero_symmetric(self, parms_id=None, destination=None):
        destination = destination if destination is not None else Ciphertext(self.context)
        if parms_id is None:
            N.check(N.lib().Encryptor_EncryptZeroSymmetric2(self._h, C.c_bool(False), destination._h, None))
        else:
            N.check(N.lib().Encryptor_EncryptZeroSymmetric1(self._h, (C.c_uint64 * 4)(*parms_id), C.c_bool(False), destination._h, None))
        return destination

    def encrypt_symmetric(self, plain, destination=None):
        destination = destination if destination is not None else Ciphertext(self.context)
        N.check(N.lib().Encryptor_EncryptSymmetric(self._h, plain._h, C.c_bool(False), destination._h, None))
        return destination

    def _save(self, parms_id, call):
        cap = C.c_int64()
        N.check(N.lib().Encryptor_SymmetricSaveSize(self._h, (C.c_uint64 * 4)(*parms_id), C.byref(cap)))
        buf = (C.c_uint8 * cap.value)()
        n = C.c_int64()
        N.check(call(buf, C.c_uint64(cap.value), C.byref(n)))
        return C.string_at(buf, n.value)

    def encrypt_zero_symmetric_save(self, parms_id):
        pid = (C.c_uint64 * 4)(*parms_id)
        return self._save(parms_id, lambda b, c, n: N.lib().Encryptor_EncryptZeroSymmetricSave(self._h, pid, b, c, n))

    def encrypt_symmetric_save(self, plain):
        pid = plain.parms_id() if plain.is_ntt_form() else self.context.first_parms_id()
        return self._save(pid, lambda b, c, n: N.lib().Encryptor_EncryptSymmetricSave(self._h, plain._h, b, c, n))


class Graph:
    """an executable hipGraph recorded by Evaluator.capture(); launch() is stream-ordered on the evaluator's stream"""

    def __init__(self, evaluator, handle):
        self.evaluator, self._h = evaluator, handle

    def launch(self):
        N.check(N.lib().Evaluator_LaunchGraph(self.evaluator._h, self._h))

    def __del__(self):
        if getattr(self, "_h", None):
            N.lib().Graph_Destroy(self._h)
            self._h = None


class Evaluator:
    """seal::Evaluator's hot-path surface (evaluator.h:79-1387), in-place and destination forms."""

    def __init__(self, context):
        self.context = context
        self._h = C.c_void_p()
        N.check(N.lib().Evaluator_Create(context._h, C.byref(self._h)))

    def __del__(self):
        if getattr(self, "_h", None):
            N.lib().Evaluator_Destroy(self._h)
            self._h = None

    def set_stream(self, stream):
        N.check(N.lib().Evaluator_SetStream(self._h, C.c_void_p(stream or 0)))

    def set_transparent_check(self, on):
        N.check(N.lib().Evaluator_SetTransparentCheck(self._h, C.c_bool(on)))

    def synchronize(self):
        N.check(N.lib().Evaluator_Synchronize(self._h))

    # -- hipGraph capture of a fixed operation sequence (sealhip.h: Evaluator_BeginCapture ...)
    def capture(self, fn):
        """record the evaluator operations issued by fn() into a graph; returns a Graph whose launch() replays them"""
        N.check(N.lib().Evaluator_BeginCapture(self._h))
        try:
            fn()
        finally:
            g = C.c_void_p()
            hr = N.lib().Evaluator_EndCapture(self._h, C.byref(g))
        N.check(hr)
        return Graph(self, g)

    def _u(self, fn, a, dest, *extra, pool=False):
        d = a if dest is None else dest
        args = [self._h, a._h] + list(extra) + [d._h]
        if pool:
            args.append(None)
        N.check(getattr(N.lib(), fn)(*args))
        return d

    def copy_to(self, a, destination):
        """destination := a on this evaluator's stream (library extension: Evaluator_CopyTo)"""
        N.check(N.lib().Evaluator_CopyTo(self._h, a._h, destination._h))
        return destination

    def negate_inplace(self, a):
        return self._u("Evaluator_Negate", a, None)

    def negate(self, a, destination):
        return self._u("Evaluator_Negate", a, destination)

    def add_inplace(self, a, b):
        N.check(N.lib().Evaluator_Add(self._h, a._h, b._h, a._h))
        return a

    def add(self, a, b, destination):
        N.check(N.lib().Evaluator_Add(self._h, a._h, b._h, destination._h))
        return destination

    def sub_inplace(self, a, b):
        N.check(N.lib().Evaluator_Sub(self._h, a._h, b._h, a._h))
        return a

    def sub(self, a, b, destination):
        N.check(N.lib().Evaluator_Sub(self._h, a._h, b._h, destination._h))
        return destination

    def multiply_inplace(self, a, b):
        N.check(N.lib().Evaluator_Multiply(self._h, a._h, b._h, a._h, None))
        return a

    def multiply(self, a, b, destination):
        N.check(N.lib().Evaluator_Multiply(self._h, a._h, b._h, destination._h, None))
        return destination

    def square_inplace(self, a):
        return self._u("Evaluator_Square", a, None, pool=True)

    def square(self, a, destination):
        return self._u("Evaluator_Square", a, destination, pool=True)

    def relinearize_inplace(self, a, relin_keys):
        N.check(N.lib().Evaluator_Relinearize(self._h, a._h, relin_keys._h, a._h, None))
        return a

    def relinearize(self, a, relin_keys, destination):
        N.check(N.lib().Evaluator_Relinearize(self._h, a._h, relin_keys._h, destination._h, None))
        return destination

    def mod_switch_to_next_inplace(self, a):
        return self._u("Evaluator_ModSwitchToNext1", a, None, pool=True)

    def mod_switch_to_next(self, a, destination):
        return self._u("Evaluator_ModSwitchToNext1", a, destination, pool=True)

    def mod_switch_to_inplace(self, a, parms_id):
        if isinstance(a, Plaintext):  # the reference overloads the name for plaintexts (evaluator.h:520-560)
            return self.mod_switch_plain_to_inplace(a, parms_id)
        pid = (C.c_uint64 * 4)(*parms_id)
        N.check(N.lib().Evaluator_ModSwitchTo1(self._h, a._h, pid, a._h, None))
        return a

    def mod_switch_to(self, a, parms_id, destination):
        pid = (C.c_uint64 * 4)(*parms_id)
        N.check(N.lib().Evaluator_ModSwitchTo1(self._h, a._h, pid, destination._h, None))
        return destination

    def rescale_to_next_inplace(self, a):
        return self._u("Evaluator_RescaleToNext", a, None, pool=True)

    def rescale_to_next(self, a, destination):
        return self._u("Evaluator_RescaleToNext", a, destination, pool=True)

    def rescale_to_inplace(self, a, parms_id):
        pid = (C.c_uint64 * 4)(*parms_id)
        N.check(N.lib().Evaluator_RescaleTo(self._h, a._h, pid, a._h, None))
        return a

    def rescale_to(self, a, parms_id, destination):
        pid = (C.c_uint64 * 4)(*parms_id)
        N.check(N.lib().Evaluator_RescaleTo(self._h, a._h, pid, destination._h, None))
        return destination

    def mod_reduce_to_next_inplace(self, a):
        return self._u("Evaluator_ModReduceToNext", a, None, pool=True)

    def mod_reduce_to_inplace(self, a, parms_id):
        pid = (C.c_uint64 * 4)(*parms_id)
        N.check(N.lib().Evaluator_ModReduceTo(self._h, a._h, pid, a._h, None))
        return a

    def transform_to_ntt_inplace(self, a):
        return self._u("Evaluator_TransformToNTT2", a, None)

    def transform_from_ntt_inplace(self, a):
        return self._u("Evaluator_TransformFromNTT", a, None)

    def apply_galois_inplace(self, a, galois_elt, galois_keys):
        N.check(N.lib().Evaluator_ApplyGalois(self._h, a._h, C.c_uint32(galois_elt), galois_keys._h, a._h, None))
        return a

    def rotate_rows_inplace(self, a, steps, galois_keys):
        N.check(N.lib().Evaluator_RotateRows(self._h, a._h, C.c_int(steps), galois_keys._h, a._h, None))
        return a

    def rotate_columns_inplace(self, a, galois_keys):
        N.check(N.lib().Evaluator_RotateColumns(self._h, a._h, galois_keys._h, a._h, None))
        return a

    def rotate_vector_inplace(self, a, steps, galois_keys):
        N.check(N.lib().Evaluator_RotateVector(self._h, a._h, C.c_int(steps), galois_keys._h, a._h, None))
        return a

    # out-of-place forms: with the exact Galois key present the operand is read where it lies (no copy into the destination)
    def apply_galois(self, a, galois_elt, galois_keys, destination):
        N.check(N.lib().Evaluator_ApplyGalois(self._h, a._h, C.c_uint32(galois_elt), galois_keys._h, destination._h, None))
        return destination

    def rotate_rows(self, a, steps, galois_keys, destination):
        N.check(N.lib().Evaluator_RotateRows(self._h, a._h, C.c_int(steps), galois_keys._h, destination._h, None))
        return destination

    def rotate_columns(self, a, galois_keys, destination):
        N.check(N.lib().Evaluator_RotateColumns(self._h, a._h, galois_keys._h, destination._h, None))
        return destination

    def rotate_vector(self, a, steps, galois_keys, destination):
        N.check(N.lib().Evaluator_RotateVector(self._h, a._h, C.c_int(steps), galois_keys._h, destination._h, None))
        return destination

    def complex_conjugate(self, a, galois_keys, destination):
        N.check(N.lib().Evaluator_ComplexConjugate(self._h, a._h, galois_keys._h, destination._h, None))
        return destination

    # ---- plaintext operands and many-operand forms
    def _pl(self, fn, a, plain, dest, pool=False):
        d = a if dest is None else dest
        args = [self._h, a._h, plain._h, d._h] + ([None] if pool else [])
        N.check(getattr(N.lib(), fn)(*args))
        return d

    def add_plain_inplace(self, a, plain):
        return self._pl("Evaluator_AddPlain", a, plain, None)

    def sub_plain_inplace(self, a, plain):
        return self._pl("Evaluator_SubPlain", a, plain, None)

    def multiply_plain_inplace(self, a, plain):
        return self._pl("Evaluator_MultiplyPlain", a, plain, None, pool=True)

    def multiply_plain(self, a, plain, destination):
        return self._pl("Evaluator_MultiplyPlain", a, plain, destination, pool=True)

    def transform_plain_to_ntt_inplace(self, plain, parms_id):
        pid = (C.c_uint64 * 4)(*parms_id)
        N.check(N.lib().Evaluator_TransformToNTT1(self._h, plain._h, pid, plain._h, None))
        return plain

    def mod_switch_plain_to_next_inplace(self, plain):
        N.check(N.lib().Evaluator_ModSwitchToNext2(self._h, plain._h, plain._h))
        return plain

    def mod_switch_plain_to_inplace(self, plain, parms_id):
        pid = (C.c_uint64 * 4)(*parms_id)
        N.check(N.lib().Evaluator_ModSwitchTo2(self._h, plain._h, pid, plain._h))
        return plain

    def add_many(self, cts, destination):
        arr = (C.c_void_p * len(cts))(*[c._h for c in cts])
        N.check(N.lib().Evaluator_AddMany(self._h, C.c_uint64(len(cts)), arr, destination._h))
        return destination

    def multiply_many(self, cts, relin_keys, destination):
        arr = (C.c_void_p * len(cts))(*[c._h for c in cts])
        N.check(N.lib().Evaluator_MultiplyMany(self._h, C.c_uint64(len(cts)), arr, relin_keys._h, destination._h, None))
        return destination

    def exponentiate_inplace(self, a, exponent, relin_keys):
        N.check(N.lib().Evaluator_Exponentiate(self._h, a._h, C.c_uint64(exponent), relin_keys._h, a._h, None))
        return a

    # ---- digit-parallel key switching (sealhip.h section 1b); acc_ptr = device pointer of switch_key_acc_words words
    def switch_key_acc_words(self, a):
        v = C.c_uint64()
        N.check(N.lib().Evaluator_SwitchKeyAccWords(self._h, a._h, C.byref(v)))
        return v.value

    def relinearize_partial(self, a, relin_keys, digit_first, digit_count, acc_ptr):
        N.check(N.lib().Evaluator_RelinearizePartial(self._h, a._h, relin_keys._h, C.c_uint64(digit_first), C.c_uint64(digit_count),
                                                     C.c_void_p(acc_ptr)))

    def relinearize_finish(self, a, acc_ptr, parts):
        N.check(N.lib().Evaluator_RelinearizeFinish(self._h, a._h, C.c_void_p(acc_ptr), C.c_uint64(parts)))

    def apply_galois_partial(self, a, galois_elt, galois_keys, digit_first, digit_count, acc_ptr):
        N.check(N.lib().Evaluator_ApplyGaloisPartial(self._h, a._h, C.c_uint32(galois_elt), galois_keys._h, C.c_uint64(digit_first),
                                                     C.c_uint64(digit_count), C.c_void_p(acc_ptr)))

    def apply_galois_finish(self, a, acc_ptr, parts):
        N.check(N.lib().Evaluator_ApplyGaloisFinish(self._h, a._h, C.c_void_p(acc_ptr), C.c_uint64(parts)))

    # -- digit-parallel forms with the exchange inside the library (sealhip.h section 1c)
    def relinearize_inplace_dp(self, a, relin_keys, comm, exchange=0):
        N.check(N.lib().Evaluator_RelinearizeDigitParallel(self._h, a._h, relin_keys._h, comm._h, C.c_int(exchange), a._h))
        return a

    def apply_galois_inplace_dp(self, a, galois_elt, galois_keys, comm, exchange=0):
        N.check(N.lib().Evaluator_ApplyGaloisDigitParallel(self._h, a._h, C.c_uint32(galois_elt), galois_keys._h, comm._h, C.c_int(exchange), a._h))
        return a

    def rotate_vector_inplace_dp(self, a, steps, galois_keys, comm, exchange=0):
        N.check(N.lib().Evaluator_RotateVectorDigitParallel(self._h, a._h, C.c_int(steps), galois_keys._h, comm._h, C.c_int(exchange), a._h))
        return a

    def broadcast_key_digits(self, keys, index, staging_ptr, comm, root=0):
        N.check(N.lib().Evaluator_BroadcastKeyDigits(self._h, keys._h, C.c_uint64(index), C.c_void_p(staging_ptr), comm._h, C.c_int(root)))

    def switch_key_slots(self, a, nranks):
        v = C.c_uint64()
        N.check(N.lib().Evaluator_SwitchKeySlots(self._h, a._h, C.c_uint64(nranks), C.byref(v)))
        return v.value

    def switch_key_pack_targets(self, a, acc_ptr, nranks, send_ptr, special_ptr):
        N.check(N.lib().Evaluator_SwitchKeyPackTargets(self._h, a._h, C.c_void_p(acc_ptr), C.c_uint64(nranks), C.c_void_p(send_ptr), C.c_void_p(special_ptr)))

    def switch_key_finish_owned(self, a, recv_ptr, special_ptr, nranks, rank, own_ptr):
        N.check(N.lib().Evaluator_SwitchKeyFinishOwned(self._h, a._h, C.c_void_p(recv_ptr), C.c_void_p(special_ptr), C.c_uint64(nranks),
                                                       C.c_uint64(rank), C.c_void_p(own_ptr)))

    def switch_key_add_gathered(self, a, all_ptr, nranks):
        N.check(N.lib().Evaluator_SwitchKeyAddGathered(self._h, a._h, C.c_void_p(all_ptr), C.c_uint64(nranks)))

    def complex_conjugate_inplace(self, a, galois_keys):
        N.check(N.lib().Evaluator_ComplexConjugate(self._h, a._h, galois_keys._h, a._h, None))
        return a


# ---- per-kernel seam on raw device slabs -------------------------------------------------------
class Comm:
    """One rank of a communicator over the GPUs of a node (sealhip.h section 1c): RCCL, collectives on the evaluator's stream."""
    ALL_REDUCE, REDUCE_SCATTER = 0, 1

    @staticmethod
    def unique_id():
        buf = (C.c_uint8 * 128)()
        N.check(N.lib().Comm_GetUniqueId(buf))
        return bytes(buf)

    @staticmethod
    def rccl_available():
        v = C.c_bool()
        N.check(N.lib().Comm_RcclAvailable(C.byref(v)))
        return v.value

    def __init__(self, unique_id, nranks, rank):
        self._h = C.c_void_p()
        buf = (C.c_uint8 * 128)(*bytes(unique_id))
        N.check(N.lib().Comm_Create(buf, C.c_int(nranks), C.c_int(rank), C.byref(self._h)))
        self.nranks, self.rank = nranks, rank

    def __del__(self):
        if getattr(self, "_h", None):
            N.lib().Comm_Destroy(self._h)
            self._h = None

    def loopback(self):
        v = C.c_bool()
        N.check(N.lib().Comm_Info(self._h, None, None, C.byref(v)))
        return v.value

    def digit_range(self, digits):
        a, b = C.c_uint64(), C.c_uint64()
        N.check(N.lib().Comm_DigitRange(self._h, C.c_uint64(digits), C.byref(a), C.byref(b)))
        return a.value, b.value

    def all_reduce(self, device_ptr, words, stream=None):
        N.check(N.lib().Comm_AllReduceWords(self._h, C.c_void_p(device_ptr), C.c_uint64(words), C.c_void_p(stream or 0)))

    def broadcast(self, device_ptr, words, root=0, stream=None):
        N.check(N.lib().Comm_BroadcastWords(self._h, C.c_void_p(device_ptr), C.c_uint64(words), C.c_int(root), C.c_void_p(stream or 0)))


class DeviceBuffer:
    """A raw uint64 slab in HBM (shl_malloc) for the per-kernel entry points."""

    def __init__(self, words):
        self.words = int(words)
        self._ptr = C.c_void_p()
        N.check(N.lib().shl_malloc(C.c_uint64(self.words * 8), C.byref(self._ptr)))

    def __del__(self):
        if getattr(self, "_ptr", None) and self._ptr.value:
            N.lib().shl_free(self._ptr)
            self._ptr = None

    @property
    def ptr(self):
        return self._ptr.value

    @staticmethod
    def from_numpy(a):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        b = DeviceBuffer(a.size)
        N.check(N.lib().shl_memcpy_h2d(b._ptr, _p(a), C.c_uint64(a.size * 8)))
        return b

    def to_numpy(self, shape):
        out = np.zeros(shape, dtype=np.uint64)
        N.check(N.lib().shl_memcpy_d2h(_p(out), self._ptr, C.c_uint64(out.size * 8)))
        return out


def ntt_forward(context, buf, polys, comps, first_prime=0, lazy=False, stream=None):
    N.check(N.lib().shl_ntt_forward(context._h, C.c_void_p(buf.ptr), C.c_uint64(polys), C.c_uint64(comps),
                                    C.c_uint64(first_prime), C.c_int(int(lazy)), C.c_void_p(stream or 0)))


def ntt_inverse(context, buf, polys, comps, first_prime=0, lazy=False, stream=None):
    N.check(N.lib().shl_ntt_inverse(context._h, C.c_void_p(buf.ptr), C.c_uint64(polys), C.c_uint64(comps),
                                    C.c_uint64(first_prime), C.c_int(int(lazy)), C.c_void_p(stream or 0)))


def dyadic_product(context, a, b, r, polys, comps, first_prime=0, stream=None):
    N.check(N.lib().shl_dyadic_product(context._h, C.c_void_p(a.ptr), C.c_void_p(b.ptr), C.c_void_p(r.ptr),
                                       C.c_uint64(polys), C.c_uint64(comps), C.c_uint64(first_prime),
                                       C.c_void_p(stream or 0)))


def apply_galois(context, chain_index, ntt_form, galois_elt, src, dst, polys, stream=None):
    N.check(N.lib().shl_apply_galois(context._h, C.c_uint64(chain_index), C.c_int(int(ntt_form)), C.c_uint32(galois_elt),
                                     C.c_void_p(src.ptr), C.c_void_p(dst.ptr), C.c_uint64(polys), C.c_void_p(stream or 0)))


RNS_STAGE = {"fastbconv_m_tilde": 0, "sm_mrq": 1, "fast_floor": 2, "fastbconv_sk": 3,
             "divide_and_round_q_last": 4, "divide_and_round_q_last_ntt": 5}


def rns_stage(context, chain_index, which, src, dst, polys, stream=None):
    N.check(N.lib().shl_rns_stage(context._h, C.c_uint64(chain_index), C.c_int(RNS_STAGE[which]), C.c_void_p(src.ptr),
                                  C.c_void_p(dst.ptr), C.c_uint64(polys), C.c_void_p(stream or 0)))


class Stream:
    """a HIP stream owned by the library's caller (shl_stream_create); .handle goes to Evaluator.set_stream"""

    def __init__(self, non_blocking=True):
        self._h = C.c_void_p()
        N.check(N.lib().shl_stream_create(C.c_bool(non_blocking), C.byref(self._h)))

    @property
    def handle(self):
        return self._h.value

    def __del__(self):
        if getattr(self, "_h", None) and self._h.value:
            N.lib().shl_stream_destroy(self._h)
            self._h = None


def device_count():
    n = C.c_int()
    N.check(N.lib().shl_device_count(C.byref(n)))
    return n.value


def set_device(index):
    """select this thread's GPU (one process per GPU) before any context is created"""
    N.check(N.lib().shl_set_device(C.c_int(index)))


def release_pool():
    """return the library's cached HBM blocks to the driver"""
    N.check(N.lib().SealHip_ReleasePool())


def set_staged_host_copies(enabled):
    """host <-> device copies of ciphertext words through the library's pinned bounce buffers (sealhip.h) instead of a direct
    hipMemcpy on the caller's pageable buffer; process-wide"""
    N.check(N.lib().SealHip_SetStagedHostCopies(C.c_bool(enabled)))


def install_abort_trace(path):
    """append the call stack of an aborting thread to `path` before the process dies (sealhip.h: SealHip_InstallAbortTrace)"""
    N.check(N.lib().SealHip_InstallAbortTrace(C.c_char_p(os.fsencode(path))))


def pool_stats():
    """(bytes held by the pool, number of cross-stream hand-outs ordered by an event)"""
    a, b = C.c_uint64(), C.c_uint64()
    N.check(N.lib().SealHip_PoolStats(C.byref(a), C.byref(b)))
    return a.value, b.value


def tail_stats():
    """deferred key-switch tails (sealhip.h: SealHip_TailStats): (folded into a rescale, completed on their own, discarded)"""
    a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
    N.check(N.lib().SealHip_TailStats(C.byref(a), C.byref(b), C.byref(c)))
    return a.value, b.value, c.value


def product_stats():
    """deferred tensor products (sealhip.h: SealHip_ProductStats): (consumed by a fused relinearize, formed on their own, discarded)"""
    a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
    N.check(N.lib().SealHip_ProductStats(C.byref(a), C.byref(b), C.byref(c)))
    return a.value, b.value, c.value


def galois_stats():
    """rotations (sealhip.h: SealHip_GaloisStats): (operand read through the index map inside the key switch, permutation kernels run)"""
    a, b = C.c_uint64(), C.c_uint64()
    N.check(N.lib().SealHip_GaloisStats(C.byref(a), C.byref(b)))
    return a.value, b.value


def ks_chunk_stats():
    """chunked key switching (sealhip.h: SealHip_KsChunkStats): (calls that ran in chunks, chunks issued, largest intermediate in bytes)"""
    a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
    N.check(N.lib().SealHip_KsChunkStats(C.byref(a), C.byref(b), C.byref(c)))
    return a.value, b.value, c.value


def device_synchronize():
    N.check(N.lib().shl_device_synchronize())


def device_info():
    name = C.create_string_buffer(256)
    cus = C.c_int()
    mem = C.c_uint64()
    N.check(N.lib().SealHip_DeviceInfo(name, C.c_uint64(256), C.byref(cus), C.byref(mem)))
    return name.value.decode(), cus.value, mem.value


class HipTimer:
    """HIP-event timer on a given stream (shl_timer_*)."""

    def __init__(self):
        self._h = C.c_void_p()
        N.check(N.lib().shl_timer_create(C.byref(self._h)))

    def __del__(self):
        if getattr(self, "_h", None):
            N.lib().shl_timer_destroy(self._h)
            self._h = None

    def start(self, stream=None):
        N.check(N.lib().shl_timer_start(self._h, C.c_void_p(stream or 0)))

    def stop(self, stream=None):
        ms = C.c_float()
        N.check(N.lib().shl_timer_stop(self._h, C.c_void_p(stream or 0), C.byref(ms)))
        return ms.value
