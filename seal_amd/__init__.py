"""seal_amd — MI355X (gfx950) implementation of Microsoft SEAL's RNS polynomial-arithmetic hot path
behind the reference's Evaluator / Ciphertext API surface.  The arithmetic lives in
seal_amd/lib/libsealhip.so (hand-written HIP, C ABI in include/sealhip.h); this package is the thin
Python host side used by the parity tests and the benchmark.  No CPU fallback exists."""
from .api import (CoeffModulus, PlainModulus, EncryptionParameters, SEALContext, ContextData, Ciphertext, Plaintext, KSwitchKeys,
                  RelinKeys, GaloisKeys, Evaluator, SecretKey, PublicKey, KeyGenerator, Decryptor, Encryptor, BatchEncoder, CKKSEncoder, Graph, DeviceBuffer, HipTimer, ntt_forward, ntt_inverse,
                  dyadic_product, apply_galois, rns_stage, device_synchronize, device_info, release_pool, pool_stats, tail_stats, product_stats, galois_stats, ks_chunk_stats, Stream, Comm, device_count, set_device,
                  set_staged_host_copies, install_abort_trace)
from ._native import SealHipError, InvalidArgument, LogicError, OutOfRange, DeviceError, load

__all__ = [
    "CoeffModulus", "PlainModulus", "EncryptionParameters", "SEALContext", "ContextData", "Ciphertext", "Plaintext", "KSwitchKeys",
    "RelinKeys", "GaloisKeys", "Evaluator", "SecretKey", "PublicKey", "KeyGenerator", "Decryptor", "Encryptor", "BatchEncoder", "CKKSEncoder", "Graph", "DeviceBuffer", "HipTimer", "ntt_forward", "ntt_inverse",
    "dyadic_product", "apply_galois", "rns_stage", "device_synchronize", "device_info", "release_pool", "pool_stats", "tail_stats", "product_stats", "galois_stats", "ks_chunk_stats", "Stream", "Comm", "device_count", "set_device", "set_staged_host_copies", "install_abort_trace", "SealHipError",
    "InvalidArgument", "LogicError", "OutOfRange", "DeviceError", "load",
]
