#!/usr/bin/env python
"""Writes tests/golden/serial_*.bin + serial_golden.json from the REAL reference (oracle/_ref, built from /root/reference):
seeded Serializable<Ciphertext> streams (CKKS / BFV / BGV), a SHAKE256-seeded variant, and a seeded RelinKeys stream, with
the SHA-256 of what the reference itself loads / re-saves / computes from them.  Run here (needs /root/reference)."""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import sealref  # noqa: E402


def sha(a):
    return hashlib.sha256(a if isinstance(a, bytes) else np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    out = dict(ciphertexts=[], keys=[])
    for scheme, n, bits, tb in (("ckks", 256, [40, 30, 40], 0), ("bfv", 256, [36, 36, 37], 20), ("bgv", 256, [40, 40, 45], 20)):
        primes = sealref.coeff_modulus_create(n, bits)
        t = sealref.plain_modulus_batching(n, tb) if tb else 0
        ref = sealref.RefContext(scheme, n, primes, t)
        pids = [list(ref.parms_id(ci)) for ci in range(ref.key_chain_index + 1)]
        for variant in ("blake2xb", "shake256"):
            data = bytearray(ref.encrypt_zero_symmetric_save(ref.first_chain_index, True))
            if variant == "shake256":
                if scheme != "ckks":
                    continue
                data[len(data) - 65] = 2
            data = bytes(data)
            rct, _ = ref.ct_load(data)
            i = rct.info()
            name = "serial_%s_seeded_%s.bin" % (scheme, variant)
            open(os.path.join(HERE, name), "wb").write(data)
            out["ciphertexts"].append(dict(file=name, scheme=scheme, n=n, primes=primes, plain_modulus=t, parms_ids=pids,
                                           size=i["size"], is_ntt_form=i["is_ntt_form"], scale=i["scale"],
                                           correction_factor=i["correction_factor"], sha256_words=sha(rct.data()),
                                           sha256_full_stream=sha(ref.ct_save(rct))))
        if scheme == "ckks":
            stream = ref.keys_save("relin", True)
            name = "serial_ckks_relinkeys_seeded.bin"
            open(os.path.join(HERE, name), "wb").write(stream)
            K = len(primes) - 1
            x3 = np.stack([np.stack([np.random.default_rng(100 + p * 16 + i).integers(0, primes[i], n, dtype=np.uint64) for i in range(K)])
                           for p in range(3)])
            rx = ref.ct(ref.first_chain_index, x3, True, 2.0 ** 10)
            ref.relinearize_inplace(rx)
            out["keys"].append(dict(file=name, scheme=scheme, n=n, primes=primes, plain_modulus=t, sha256_relinearized=sha(rx.data())))
    json.dump(out, open(os.path.join(HERE, "serial_golden.json"), "w"), indent=1)
    print("wrote", [c["file"] for c in out["ciphertexts"]] + [k["file"] for k in out["keys"]])


if __name__ == "__main__":
    main()
