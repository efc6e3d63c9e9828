"""ctypes wrapper over oracle/_ref/libsealref.so — the REAL reference (Microsoft SEAL 4.4.3,
HEXL off) compiled from /root/reference by oracle/Makefile.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
`available()` is False when the prebuilt library is absent (then tests fall back to the
plain-C restatement in oracle/seal_oracle.c and to tests/golden/ fixtures).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SEALREF_LIB: another flavour of the same reference (bench.py's cpu_baseline times oracle/_ref/libsealref_clang.so in a child process)
LIB_PATH = os.environ.get("SEALREF_LIB") or os.path.join(_HERE, "..", "oracle", "_ref", "libsealref.so")
CLANG_LIB_PATH = os.path.join(_HERE, "..", "oracle", "_ref", "libsealref_clang.so")

_lib = None


def available():
    return os.path.exists(LIB_PATH)


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(LIB_PATH)
        _lib.ref_galois_elt_from_step.restype = C.c_uint64
    return _lib


class RefError(Exception):
    NAMES = {1: "invalid_argument", 2: "logic_error", 3: "out_of_range", 4: "other"}

    def __init__(self, code):
        super().__init__(self.NAMES.get(code, str(code)))
        self.code = code


def _ck(rc):
    if rc != 0:
        raise RefError(rc)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


SCHEME = {"bfv": 1, "ckks": 2, "bgv": 3}


def coeff_modulus_create(n, bit_sizes):
    bits = (C.c_int * len(bit_sizes))(*bit_sizes)
    out = np.zeros(len(bit_sizes), dtype=np.uint64)
    _ck(lib().ref_coeff_modulus_create(C.c_uint64(n), bits, C.c_uint64(len(bit_sizes)), _p(out)))
    return [int(x) for x in out]


def plain_modulus_batching(n, bits):
    out = C.c_uint64()
    _ck(lib().ref_plain_modulus_batching(C.c_uint64(n), C.c_int(bits), C.byref(out)))
    return out.value


def parms_load(data):
    """EncryptionParameters::load -> (scheme, poly_modulus_degree, primes, plain_modulus); raises RefError with the reference's class"""
    buf = (C.c_uint8 * len(data)).from_buffer_copy(bytes(data))
    scheme = C.c_int()
    deg, plain, cnt = C.c_uint64(), C.c_uint64(), C.c_uint64(256)
    primes = np.zeros(256, dtype=np.uint64)
    _ck(lib().ref_parms_load(buf, C.c_uint64(len(data)), C.byref(scheme), C.byref(deg), C.byref(plain), _p(primes), C.byref(cnt)))
    return scheme.value, deg.value, [int(x) for x in primes[: cnt.value]], plain.value


def bfv_default(n):
    out = np.zeros(64, dtype=np.uint64)
    cnt = C.c_uint64()
    _ck(lib().ref_bfv_default(C.c_uint64(n), _p(out), C.byref(cnt)))
    return [int(x) for x in out[: cnt.value]]


class RefCiphertext:
    def __init__(self, ctx, handle):
        self.ctx = ctx
        self.h = handle

    def __del__(self):
        if self.h:
            lib().ref_ct_destroy(self.h)
            self.h = None

    def info(self):
        ci, size, k = C.c_uint64(), C.c_uint64(), C.c_uint64()
        ntt = C.c_int()
        scale = C.c_double()
        cf = C.c_uint64()
        lib().ref_ct_info(self.ctx.h, self.h, C.byref(ci), C.byref(size), C.byref(k), C.byref(ntt),
                          C.byref(scale), C.byref(cf))
        return dict(chain_index=ci.value, size=size.value, coeff_modulus_size=k.value,
                    is_ntt_form=bool(ntt.value), scale=scale.value, correction_factor=cf.value)

    def data(self):
        i = self.info()
        out = np.zeros((i["size"], i["coeff_modulus_size"], self.ctx.n), dtype=np.uint64)
        lib().ref_ct_data(self.h, _p(out))
        return out

    def copy(self):
        h = C.c_void_p()
        _ck(lib().ref_ct_copy(self.h, C.byref(h)))
        return RefCiphertext(self.ctx, h)


class RefPlaintext:
    def __init__(self, ctx, handle):
        self.ctx = ctx
        self.h = handle

    def __del__(self):
        if self.h:
            lib().ref_pt_destroy(self.h)
            self.h = None

    def info(self):
        ci, cnt = C.c_uint64(), C.c_uint64()
        ntt = C.c_int()
        scale = C.c_double()
        lib().ref_pt_info(self.ctx.h, self.h, C.byref(ci), C.byref(cnt), C.byref(ntt), C.byref(scale))
        return dict(chain_index=None if ci.value == 2 ** 64 - 1 else ci.value, coeff_count=cnt.value,
                    is_ntt_form=bool(ntt.value), scale=scale.value)

    def data(self):
        out = np.zeros(self.info()["coeff_count"], dtype=np.uint64)
        lib().ref_pt_data(self.h, _p(out))
        return out


class RefContext:
    """SEALContext(parms, expand_mod_chain=True, sec_level_type::none) + KeyGenerator + Evaluator."""

    def __init__(self, scheme, n, primes, plain_modulus=0, seed=0x5EA1):
        self.scheme = scheme
        self.n = n
        self.primes = list(primes)
        self.plain_modulus = plain_modulus
        arr = np.array(primes, dtype=np.uint64)
        h = C.c_void_p()
        _ck(lib().ref_ctx_create(C.c_int(SCHEME[scheme]), C.c_uint64(n), _p(arr), C.c_uint64(len(primes)),
                                 C.c_uint64(plain_modulus), C.c_uint64(seed), C.byref(h)))
        self.h = h
        a, b, u = C.c_uint64(), C.c_uint64(), C.c_int()
        lib().ref_ctx_info(self.h, C.byref(a), C.byref(b), C.byref(u))
        self.key_chain_index, self.first_chain_index, self.using_keyswitching = a.value, b.value, bool(u.value)

    def __del__(self):
        if getattr(self, "h", None):
            lib().ref_ctx_destroy(self.h)
            self.h = None

    # -- introspection
    def level_primes(self, chain_index):
        out = np.zeros(256, dtype=np.uint64)
        cnt = C.c_uint64()
        _ck(lib().ref_ctx_level_primes(self.h, C.c_uint64(chain_index), _p(out), C.byref(cnt)))
        return [int(x) for x in out[: cnt.value]]

    def ntt_tables(self, chain_index, idx, want_tables=True):
        root, invn = C.c_uint64(), C.c_uint64()
        fwd = np.zeros(self.n, dtype=np.uint64) if want_tables else None
        inv = np.zeros(self.n, dtype=np.uint64) if want_tables else None
        _ck(lib().ref_ctx_ntt_tables(self.h, C.c_uint64(chain_index), C.c_uint64(idx), C.byref(root), C.byref(invn),
                                     _p(fwd) if want_tables else None, _p(inv) if want_tables else None))
        return root.value, invn.value, fwd, inv

    def behz_bases(self, chain_index):
        bsk = np.zeros(260, dtype=np.uint64)
        cnt, mt, g = C.c_uint64(), C.c_uint64(), C.c_uint64()
        _ck(lib().ref_ctx_behz_bases(self.h, C.c_uint64(chain_index), _p(bsk), C.byref(cnt), C.byref(mt), C.byref(g)))
        return [int(x) for x in bsk[: cnt.value]], mt.value, g.value

    # -- keys
    def keygen_relin(self):
        _ck(lib().ref_keygen_relin(self.h))

    def keygen_galois_elts(self, elts):
        a = (C.c_uint32 * len(elts))(*elts)
        _ck(lib().ref_keygen_galois_elts(self.h, a, C.c_uint64(len(elts))))

    def keygen_galois_steps(self, steps):
        a = (C.c_int * len(steps))(*steps)
        _ck(lib().ref_keygen_galois_steps(self.h, a, C.c_uint64(len(steps))))

    def key(self, kind, index):
        """-> uint64 array [digits][2][L][N]; kind 'relin' | 'galois'."""
        k = 0 if kind == "relin" else 1
        nd = C.c_uint64()
        _ck(lib().ref_key_digits(self.h, C.c_int(k), C.c_uint64(index), C.byref(nd)))
        L = len(self.primes)
        out = np.zeros((nd.value, 2, L, self.n), dtype=np.uint64)
        _ck(lib().ref_key_copy(self.h, C.c_int(k), C.c_uint64(index), _p(out)))
        return out

    def set_key(self, kind, index, words):
        """overwrite an existing key with caller-supplied residues [digits][2][L][N] (synthetic seeded keys)"""
        k = 0 if kind == "relin" else 1
        w = np.ascontiguousarray(words, dtype=np.uint64)
        _ck(lib().ref_key_set(self.h, C.c_int(k), C.c_uint64(index), _p(w)))

    # -- wire format
    # -- container surface (tests/container_cases.py)
    def data_words(self, chain_index, which):
        """0 total_coeff_modulus, 1 coeff_div_plain_modulus, 2 plain_upper_half_increment, 3 upper_half_threshold, 4 upper_half_increment;
        [] when the reference did not compute it for these parameters"""
        out = np.zeros(256, dtype=np.uint64)
        cnt = C.c_uint64()
        _ck(lib().ref_ctx_data_words(self.h, C.c_uint64(chain_index), C.c_int(which), _p(out), C.byref(cnt)))
        return [int(x) for x in out[: cnt.value]]

    def qualifiers(self, chain_index):
        out = (C.c_int * 8)()
        puht = C.c_uint64()
        _ck(lib().ref_ctx_qualifiers(self.h, C.c_uint64(chain_index), out, C.byref(puht)))
        keys = ("total_coeff_modulus_bit_count", "using_fft", "using_ntt", "using_batching", "using_fast_plain_lift",
                "using_descending_modulus_chain", "sec_level", "parameters_set")
        d = {k: int(v) for k, v in zip(keys, out)}
        d["plain_upper_half_threshold"] = puht.value
        return d

    def parms_save(self, chain_index, mode=0):
        buf = (C.c_uint8 * 8192)()
        n = C.c_uint64()
        _ck(lib().ref_parms_save(self.h, C.c_uint64(chain_index), C.c_int(mode), buf, C.c_uint64(8192), C.byref(n)))
        return bytes(buf[: n.value])

    def ct_container_op(self, ct, op, chain_index, n):
        """op: 0 reserve(context, parms_id, n)  1 reserve(n)  2 resize(context, parms_id, n)  3 resize(n)  4 release()
        -> (size, size_capacity, coeff_modulus_size, poly_modulus_degree)"""
        out = (C.c_uint64 * 4)()
        _ck(lib().ref_ct_container_op(self.h, ct.h, C.c_int(op), C.c_uint64(chain_index), C.c_uint64(n), out))
        return tuple(int(v) for v in out)

    def parms_id(self, chain_index):
        out = (C.c_uint64 * 4)()
        _ck(lib().ref_ctx_parms_id(self.h, C.c_uint64(chain_index), out))
        return tuple(out)

    def ct_save(self, ct):
        """Ciphertext::save(compr_mode_type::none) -> bytes"""
        i = ct.info()
        cap = 4096 + 8 * i["size"] * i["coeff_modulus_size"] * self.n
        buf = (C.c_uint8 * cap)()
        n = C.c_uint64()
        _ck(lib().ref_ct_save(ct.h, buf, C.c_uint64(cap), C.byref(n)))
        return bytes(buf[:n.value])

    def ct_save_mode(self, ct, mode):
        """Ciphertext::save with compr_mode `mode` (1 = zlib) -> bytes"""
        i = ct.info()
        cap = 65536 + 9 * i["size"] * i["coeff_modulus_size"] * self.n
        buf = (C.c_uint8 * cap)()
        n = C.c_uint64()
        _ck(lib().ref_ct_save_mode(ct.h, C.c_int(mode), buf, C.c_uint64(cap), C.byref(n)))
        return bytes(buf[:n.value])

    def pt_save_mode(self, pt, mode):
        cap = 65536 + 9 * pt.info()["coeff_count"]
        buf = (C.c_uint8 * cap)()
        n = C.c_uint64()
        _ck(lib().ref_pt_save_mode(pt.h, C.c_int(mode), buf, C.c_uint64(cap), C.byref(n)))
        return bytes(buf[:n.value])

    def keys_save_mode(self, kind, mode):
        """the context's current RelinKeys / GaloisKeys object saved with compr_mode `mode`"""
        k = 0 if kind == "relin" else 1
        L = len(self.primes)
        cap = 1 << 20
        cap += 9 * self.n * (2 + self.key_slots(kind) ) + 10 * (L - 1) * 2 * L * self.n * max(1, sum(1 for _ in range(1)))
        cap *= 4
        buf = (C.c_uint8 * cap)()
        n = C.c_uint64()
        _ck(lib().ref_keys_save_mode(self.h, C.c_int(k), C.c_int(mode), buf, C.c_uint64(cap), C.byref(n)))
        return bytes(buf[:n.value])

    def ct_load(self, data, unsafe=False):
        """Ciphertext::load / unsafe_load -> (RefCiphertext, bytes consumed)"""
        buf = (C.c_uint8 * max(1, len(data))).from_buffer_copy(bytes(data) or b"\x00")
        h, n = C.c_void_p(), C.c_uint64()
        _ck(lib().ref_ct_load(self.h, buf, C.c_uint64(len(data)), C.c_int(1 if unsafe else 0), C.byref(h), C.byref(n)))
        return RefCiphertext(self, h), n.value

    def encrypt_zero_symmetric_save(self, chain_index, seeded=True):
        """Encryptor::encrypt_zero_symmetric(parms_id) saved (seeded: the Serializable<> form with c_1 as a seed)"""
        cap = 4096 + 8 * 2 * len(self.primes) * self.n
        buf = (C.c_uint8 * cap)()
        n = C.c_uint64()
        _ck(lib().ref_encrypt_zero_symmetric_save(self.h, C.c_uint64(chain_index), C.c_int(1 if seeded else 0), buf, C.c_uint64(cap), C.byref(n)))
        return bytes(buf[:n.value])

    def keys_save(self, kind, seeded=True, elts=()):
        """RelinKeys / GaloisKeys saved (seeded: Serializable<> form); the context's key object is reloaded from the stream"""
        k = 0 if kind == "relin" else 1
        L = len(self.primes)
        nkeys = 1 if k == 0 else max(1, len(elts))
        cap = 65536 + 16 * self.n + nkeys * (L - 1) * (4096 + 8 * 2 * L * self.n)  # 8 bytes per (mostly empty) key slot
        buf = (C.c_uint8 * cap)()
        n = C.c_uint64()
        a = (C.c_uint32 * max(1, len(elts)))(*elts)
        _ck(lib().ref_keys_save(self.h, C.c_int(k), C.c_int(1 if seeded else 0), a, C.c_uint64(len(elts)), buf, C.c_uint64(cap), C.byref(n)))
        return bytes(buf[:n.value])

    def pt_save(self, pt):
        cap = 4096 + 8 * pt.info()["coeff_count"]
        buf = (C.c_uint8 * cap)()
        n = C.c_uint64()
        _ck(lib().ref_pt_save(pt.h, buf, C.c_uint64(cap), C.byref(n)))
        return bytes(buf[:n.value])

    def pt_load(self, data, unsafe=False):
        buf = (C.c_uint8 * max(1, len(data))).from_buffer_copy(bytes(data) or b"\x00")
        h, n = C.c_void_p(), C.c_uint64()
        _ck(lib().ref_pt_load(self.h, buf, C.c_uint64(len(data)), C.c_int(1 if unsafe else 0), C.byref(h), C.byref(n)))
        return RefPlaintext(self, h), n.value

    def ckks_encode(self, values, chain_index, scale):
        v = np.ascontiguousarray(values, dtype=np.float64)
        h = C.c_void_p()
        _ck(lib().ref_ckks_encode(self.h, _p(v), C.c_uint64(v.size), C.c_uint64(chain_index), C.c_double(scale), C.byref(h)))
        return RefPlaintext(self, h)

    def batch_encode(self, values):
        v = np.ascontiguousarray(values, dtype=np.uint64)
        h = C.c_void_p()
        _ck(lib().ref_batch_encode(self.h, _p(v), C.c_uint64(v.size), C.byref(h)))
        return RefPlaintext(self, h)

    def secret_key(self):
        """SecretKey::data(): [L][N] words, key level, NTT form"""
        out = np.zeros((len(self.primes), self.n), dtype=np.uint64)
        _ck(lib().ref_secret_key_copy(self.h, _p(out)))
        return out

    def secret_key_save(self):
        cap = 4096 + 8 * len(self.primes) * self.n
        buf = (C.c_uint8 * cap)()
        n = C.c_uint64()
        _ck(lib().ref_secret_key_save(self.h, buf, C.c_uint64(cap), C.byref(n)))
        return bytes(buf[:n.value])

    def decrypt(self, ct):
        """Decryptor::decrypt -> RefPlaintext"""
        h = C.c_void_p()
        _ck(lib().ref_decrypt(self.h, ct.h, C.byref(h)))
        return RefPlaintext(self, h)

    def encrypt_symmetric_save(self, pt, seeded=True):
        cap = 4096 + 8 * 2 * len(self.primes) * self.n
        buf = (C.c_uint8 * cap)()
        n = C.c_uint64()
        _ck(lib().ref_encrypt_symmetric_save(self.h, pt.h, C.c_int(1 if seeded else 0), buf, C.c_uint64(cap), C.byref(n)))
        return bytes(buf[:n.value])

    def batch_decode(self, pt, signed=False):
        out = np.zeros(self.n, dtype=np.int64 if signed else np.uint64)
        _ck(lib().ref_batch_decode(self.h, pt.h, C.c_int(1 if signed else 0), _p(out)))
        return out

    def batch_encode_signed(self, values):
        v = np.ascontiguousarray(values, dtype=np.int64)
        h = C.c_void_p()
        _ck(lib().ref_batch_encode_signed(self.h, _p(v), C.c_uint64(v.size), C.byref(h)))
        return RefPlaintext(self, h)

    def public_key(self):
        out = np.zeros((2, len(self.primes), self.n), dtype=np.uint64)
        _ck(lib().ref_public_key_copy(self.h, _p(out)))
        return out

    def encrypt_asymmetric_save(self, pt=None, chain_index=0):
        """Encryptor(context, public_key).encrypt(pt) (or encrypt_zero at chain_index) saved in full"""
        cap = 4096 + 8 * 2 * len(self.primes) * self.n
        buf = (C.c_uint8 * cap)()
        n = C.c_uint64()
        _ck(lib().ref_encrypt_asymmetric_save(self.h, pt.h if pt is not None else None, C.c_uint64(chain_index), buf, C.c_uint64(cap), C.byref(n)))
        return bytes(buf[:n.value])

    def ckks_encode_complex(self, values, chain_index, scale):
        v = np.ascontiguousarray(values, dtype=np.complex128)
        h = C.c_void_p()
        _ck(lib().ref_ckks_encode_complex(self.h, _p(v), C.c_uint64(v.size), C.c_uint64(chain_index), C.c_double(scale), C.byref(h)))
        return RefPlaintext(self, h)

    def ckks_encode_value(self, value, chain_index, scale=None):
        """encode(double value, parms_id, scale) or, with scale None, encode(int64 value, parms_id)"""
        h = C.c_void_p()
        integer = scale is None
        _ck(lib().ref_ckks_encode_value(self.h, C.c_double(0.0 if integer else float(value)), C.c_int(1 if integer else 0),
                                        C.c_int64(int(value) if integer else 0), C.c_uint64(chain_index), C.c_double(scale or 1.0), C.byref(h)))
        return RefPlaintext(self, h)

    def ckks_decode(self, pt, complex_values=False):
        out = np.zeros(self.n // 2, dtype=np.complex128 if complex_values else np.float64)
        _ck(lib().ref_ckks_decode(self.h, pt.h, C.c_int(1 if complex_values else 0), _p(out)))
        return out

    def noise_budget(self, ct):
        v = C.c_int()
        _ck(lib().ref_noise_budget(self.h, ct.h, C.byref(v)))
        return v.value

    def keys_load(self, data, unsafe=False):
        buf = (C.c_uint8 * max(1, len(data))).from_buffer_copy(bytes(data) or b"\x00")
        n = C.c_uint64()
        _ck(lib().ref_keys_load(self.h, buf if len(data) else None, C.c_uint64(len(data)), C.c_int(1 if unsafe else 0), C.byref(n)))
        return n.value

    def public_key_save_seeded(self):
        """Serializable<PublicKey>::save of a fresh public key (the seeded form)"""
        cap = 2 * len(self.primes) * self.n * 8 + 4096
        buf = (C.c_uint8 * cap)()
        n = C.c_uint64()
        _ck(lib().ref_public_key_save_seeded(self.h, buf, C.c_uint64(cap), C.byref(n)))
        return bytes(buf[: n.value])

    def public_key_load_words(self, data):
        """PublicKey::load(stream).data() as uint64 [2][L][N]"""
        out = np.zeros((2, len(self.primes), self.n), dtype=np.uint64)
        buf = (C.c_uint8 * len(data)).from_buffer_copy(bytes(data))
        _ck(lib().ref_public_key_load_words(self.h, buf, C.c_uint64(len(data)), _p(out)))
        return out

    def keys_install(self, kind, data):
        """the context's RelinKeys ('relin') / GaloisKeys ('galois') object := the serialized stream"""
        buf = (C.c_uint8 * len(data)).from_buffer_copy(bytes(data))
        n = C.c_uint64()
        _ck(lib().ref_keys_install(self.h, C.c_int(0 if kind == "relin" else 1), buf, C.c_uint64(len(data)), C.byref(n)))
        return n.value

    def public_key_save(self):
        cap = 4096 + 8 * 2 * len(self.primes) * self.n
        buf = (C.c_uint8 * cap)()
        n = C.c_uint64()
        _ck(lib().ref_public_key_save(self.h, buf, C.c_uint64(cap), C.byref(n)))
        return bytes(buf[:n.value])

    def key_slots(self, kind):
        v = C.c_uint64()
        _ck(lib().ref_key_slots(self.h, C.c_int(0 if kind == "relin" else 1), C.byref(v)))
        return v.value

    def galois_elts_all(self):
        out = (C.c_uint32 * 64)()
        n = C.c_uint64()
        _ck(lib().ref_galois_elts_all(self.h, out, C.c_uint64(64), C.byref(n)))
        return [int(out[i]) for i in range(n.value)]

    def galois_elt_from_step(self, step):
        return int(lib().ref_galois_elt_from_step(self.h, C.c_int(step)))

    # -- ciphertexts
    def ct(self, chain_index, data, is_ntt, scale=1.0, correction_factor=1):
        data = np.ascontiguousarray(data, dtype=np.uint64)
        h = C.c_void_p()
        _ck(lib().ref_ct_create(self.h, C.c_uint64(chain_index), C.c_uint64(data.shape[0]), C.c_int(int(is_ntt)),
                                C.c_double(scale), C.c_uint64(correction_factor), _p(data), C.byref(h)))
        return RefCiphertext(self, h)

    def ct_assign(self, dst, src):
        """dst = src (seal::Ciphertext::operator=)"""
        _ck(lib().ref_ct_assign(dst.h, src.h))

    def _op1(self, name, ct, *args):
        _ck(getattr(lib(), name)(self.h, ct.h, *args))
        return ct

    def multiply_inplace(self, a, b):
        _ck(lib().ref_multiply_inplace(self.h, a.h, b.h))
        return a

    def add_inplace(self, a, b):
        _ck(lib().ref_add_inplace(self.h, a.h, b.h))
        return a

    def sub_inplace(self, a, b):
        _ck(lib().ref_sub_inplace(self.h, a.h, b.h))
        return a

    def square_inplace(self, a):
        return self._op1("ref_square_inplace", a)

    def negate_inplace(self, a):
        return self._op1("ref_negate_inplace", a)

    def relinearize_inplace(self, a):
        return self._op1("ref_relinearize_inplace", a)

    def rescale_to_next_inplace(self, a):
        return self._op1("ref_rescale_to_next_inplace", a)

    def mod_switch_to_next_inplace(self, a):
        return self._op1("ref_mod_switch_to_next_inplace", a)

    def mod_reduce_to_next_inplace(self, a):
        return self._op1("ref_mod_reduce_to_next_inplace", a)

    # the multi-level forms of the reference itself (evaluator.cpp:1451-1473, 1543-1595, 1625-1647)
    def rescale_to_inplace(self, a, chain_index):
        return self._op1("ref_rescale_to_inplace", a, C.c_uint64(chain_index))

    def mod_switch_to_inplace(self, a, chain_index):
        return self._op1("ref_mod_switch_to_inplace", a, C.c_uint64(chain_index))

    def mod_reduce_to_inplace(self, a, chain_index):
        return self._op1("ref_mod_reduce_to_inplace", a, C.c_uint64(chain_index))

    def rotate_vector_inplace(self, a, steps):
        return self._op1("ref_rotate_vector_inplace", a, C.c_int(steps))

    def rotate_rows_inplace(self, a, steps):
        return self._op1("ref_rotate_rows_inplace", a, C.c_int(steps))

    def rotate_columns_inplace(self, a):
        return self._op1("ref_rotate_columns_inplace", a)

    def complex_conjugate_inplace(self, a):
        return self._op1("ref_complex_conjugate_inplace", a)

    def apply_galois_inplace(self, a, elt):
        return self._op1("ref_apply_galois_inplace", a, C.c_uint32(elt))

    def transform_to_ntt_inplace(self, a):
        return self._op1("ref_transform_to_ntt_inplace", a)

    def transform_from_ntt_inplace(self, a):
        return self._op1("ref_transform_from_ntt_inplace", a)

    # -- plaintext operands and many-operand forms
    def pt(self, data, chain_index=None, scale=1.0):
        """coefficient form (chain_index None) or NTT form at chain_index ([K][N] residues)"""
        d = np.ascontiguousarray(data, dtype=np.uint64).reshape(-1)
        h = C.c_void_p()
        ci = 2 ** 64 - 1 if chain_index is None else chain_index
        _ck(lib().ref_pt_create(self.h, C.c_uint64(ci), C.c_uint64(d.size), C.c_double(scale), _p(d), C.byref(h)))
        return RefPlaintext(self, h)

    def add_plain_inplace(self, a, p):
        _ck(lib().ref_add_plain_inplace(self.h, a.h, p.h))
        return a

    def sub_plain_inplace(self, a, p):
        _ck(lib().ref_sub_plain_inplace(self.h, a.h, p.h))
        return a

    def multiply_plain_inplace(self, a, p):
        _ck(lib().ref_multiply_plain_inplace(self.h, a.h, p.h))
        return a

    def pt_transform_to_ntt_inplace(self, p, chain_index):
        _ck(lib().ref_pt_transform_to_ntt_inplace(self.h, p.h, C.c_uint64(chain_index)))
        return p

    def pt_mod_switch_to_next_inplace(self, p):
        _ck(lib().ref_pt_mod_switch_to_next_inplace(self.h, p.h))
        return p

    def add_many(self, cts):
        arr = (C.c_void_p * len(cts))(*[c.h for c in cts])
        _ck(lib().ref_add_many(self.h, arr, C.c_uint64(len(cts))))
        return cts[0]

    def multiply_many(self, cts):
        arr = (C.c_void_p * len(cts))(*[c.h for c in cts])
        _ck(lib().ref_multiply_many(self.h, arr, C.c_uint64(len(cts))))
        return cts[0]

    def exponentiate_inplace(self, a, exponent):
        _ck(lib().ref_exponentiate_inplace(self.h, a.h, C.c_uint64(exponent)))
        return a

    # -- L1 kernels
    def ntt(self, chain_index, first, data, mode):
        """data [count][N] consecutive comps starting at prime `first`; mode fwd|fwd_lazy|inv|inv_lazy."""
        m = {"fwd": 0, "fwd_lazy": 1, "inv": 2, "inv_lazy": 3}[mode]
        d = np.ascontiguousarray(data, dtype=np.uint64).copy()
        _ck(lib().ref_ntt(self.h, C.c_uint64(chain_index), C.c_uint64(first), C.c_uint64(d.shape[0]), C.c_int(m), _p(d)))
        return d

    def dyadic_product(self, chain_index, idx, a, b):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        b = np.ascontiguousarray(b, dtype=np.uint64)
        r = np.zeros_like(a)
        _ck(lib().ref_dyadic_product(self.h, C.c_uint64(chain_index), C.c_uint64(idx), _p(a), _p(b), _p(r)))
        return r

    def apply_galois_raw(self, chain_index, ntt_form, elt, data):
        d = np.ascontiguousarray(data, dtype=np.uint64)
        out = np.zeros_like(d)
        _ck(lib().ref_apply_galois_raw(self.h, C.c_uint64(chain_index), C.c_int(int(ntt_form)), C.c_uint32(elt),
                                       _p(d), _p(out)))
        return out

    def rns_stage(self, chain_index, which, data, out_comps):
        w = {"fastbconv_m_tilde": 0, "sm_mrq": 1, "fast_floor": 2, "fastbconv_sk": 3,
             "divide_and_round_q_last": 4, "divide_and_round_q_last_ntt": 5}[which]
        d = np.ascontiguousarray(data, dtype=np.uint64)
        out = np.zeros((out_comps, self.n), dtype=np.uint64)
        _ck(lib().ref_rns_stage(self.h, C.c_uint64(chain_index), C.c_int(w), _p(d), _p(out)))
        return out

    # -- encode/encrypt helpers
    def ckks_encrypt(self, values, scale):
        v = np.ascontiguousarray(values, dtype=np.float64)
        h = C.c_void_p()
        _ck(lib().ref_ckks_encrypt(self.h, _p(v), C.c_uint64(len(v)), C.c_double(scale), C.byref(h)))
        return RefCiphertext(self, h)

    def ckks_decrypt(self, ct, count):
        out = np.zeros(count, dtype=np.float64)
        _ck(lib().ref_ckks_decrypt(self.h, ct.h, _p(out), C.c_uint64(count)))
        return out

    def batch_encrypt(self, values):
        v = np.ascontiguousarray(values, dtype=np.uint64)
        h = C.c_void_p()
        _ck(lib().ref_batch_encrypt(self.h, _p(v), C.c_uint64(len(v)), C.byref(h)))
        return RefCiphertext(self, h)

    def batch_decrypt(self, ct, count):
        out = np.zeros(count, dtype=np.uint64)
        _ck(lib().ref_batch_decrypt(self.h, ct.h, _p(out), C.c_uint64(count)))
        return out

    # -- CPU baseline
    def time_pipeline(self, pipeline, threads, reps):
        p = {"ckks_mul_relin_rescale": 0, "bfv_mul_relin_modswitch": 1, "rotate": 2, "ntt": 3, "ckks_chain": 4}[pipeline]
        s = C.c_double()
        _ck(lib().ref_time_pipeline(self.h, C.c_int(p), C.c_int(threads), C.c_int(reps), C.byref(s)))
        return s.value
