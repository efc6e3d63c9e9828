"""A chained C++ program against the reference's unmodified headers (seal::Evaluator = the drop-in of integration/) next to
the same chain on the device-resident C ABI at batch 1 and on the reference's own CPU Evaluator.

chain (CKKS N=65536, {60,14x50,60}): from the first data level down to two primes: multiply_inplace, relinearize_inplace,
rescale_to_next_inplace, rotate_vector_inplace(1) on one object, the second operand following with mod_switch_to_next;
the result's words are read on the host once at the end.  Each variant runs in its own process (the drop-in's page-aligned
pool needs the process to itself, tests/test_dropin.py)."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
N, BITS = 65536, [60] + [50] * 14 + [60]


def shim_variant(libname, reps):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_dropin as T
    lib = T._bind(os.path.join(ROOT, libname)) if "dropin" in libname else __import__("sealref")
    primes = lib.coeff_modulus_create(N, BITS)
    ctx = lib.RefContext("ckks", N, primes)
    ctx.keygen_relin()
    ctx.keygen_galois_steps([1])
    ctx.time_pipeline("ckks_chain", 1, 1)  # warm-up: tables, key upload
    if os.environ.get("SEALHIP_DROPIN_TRACE"):
        reps = 1
    s = ctx.time_pipeline("ckks_chain", 1, reps)
    out = dict(ms_per_chain=1e3 * s / reps)
    if "dropin" in libname:
        out["transfers"] = T._dropin_stats(lib)
    return out


def device_variant(reps, transparent_check=False):
    sys.path.insert(0, ROOT)
    import numpy as np
    import seal_amd as S
    primes = S.CoeffModulus.Create(N, BITS)
    p = S.EncryptionParameters("ckks")
    p.set_poly_modulus_degree(N)
    p.set_coeff_modulus(primes)
    ctx = S.SEALContext(p, True, 0)
    ev = S.Evaluator(ctx)
    ev.set_transparent_check(transparent_check)  # SEAL_THROW_ON_TRANSPARENT_CIPHERTEXT: a device -> host flag read per operation
    kg = S.KeyGenerator(ctx)
    rlk = kg.create_relin_keys()
    glk = kg.create_galois_keys(steps=[1])
    K = len(primes) - 1
    rng = np.random.default_rng(3)
    mk = lambda: np.stack([np.stack([rng.integers(0, primes[i], N, dtype=np.uint64) for i in range(K)]) for _ in range(2)])  # noqa: E731
    a = S.Ciphertext.from_numpy(ctx, mk(), ctx.first_parms_id(), True, 2.0 ** 24)
    b = S.Ciphertext.from_numpy(ctx, mk(), ctx.first_parms_id(), True, 2.0 ** 24)

    def chain():
        w, bb = a.copy(), b.copy()
        while w.coeff_modulus_size() > 2:
            ev.multiply_inplace(w, bb)
            ev.relinearize_inplace(w, rlk)
            ev.rescale_to_next_inplace(w)
            ev.rotate_vector_inplace(w, 1, glk)
            ev.mod_switch_to_next_inplace(bb)
            w.set_scale(2.0 ** 24)
            bb.set_scale(2.0 ** 24)
        return w.item_to_numpy(0)[0, 0, 0]
    chain()
    t0 = time.perf_counter()
    for _ in range(reps):
        chain()
    return dict(ms_per_chain=1e3 * (time.perf_counter() - t0) / reps)


def main():
    if len(sys.argv) > 1:
        kind = sys.argv[1]
        if kind == "device":
            print("RESULT " + json.dumps(device_variant(8)))
        elif kind == "device_checked":
            print("RESULT " + json.dumps(device_variant(8, True)))
        elif kind == "reference":
            print("RESULT " + json.dumps(shim_variant("oracle/_ref/libsealref.so", 1)))
        else:
            print("RESULT " + json.dumps(shim_variant("integration/_build/libsealdropin.so", 8)))
        return
    rows = []
    for name, kind, env in (("device-resident C ABI, batch 1 (Python host), transparent check on as in the drop-in", "device_checked", {}),
                            ("device-resident C ABI, batch 1 (Python host), no per-operation check", "device", {}),
                            ("drop-in behind seal::Evaluator, device-resident mirrors (SEALHIP_DROPIN_RESIDENT=1)", "dropin", {"SEALHIP_DROPIN_RESIDENT": "1"}),
                            ("drop-in, upload / download per call (the default)", "dropin", {}),
                            ("reference seal::Evaluator, 1 CPU thread", "reference", {})):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), kind], env=dict(os.environ, **env), capture_output=True, text=True, timeout=1500)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
        if r.returncode != 0 or not line:
            print("%-92s FAILED: %s" % (name, (r.stdout + r.stderr)[-400:]))
            continue
        res = json.loads(line[0][7:])
        rows.append((name, res))
        print("%-92s %9.2f ms per chain (13 levels x multiply+relinearize+rescale+rotate)%s" % (
            name, res["ms_per_chain"], "  transfers %s" % res["transfers"] if "transfers" in res else ""), flush=True)
    if len(rows) >= 3:
        print("drop-in / device-resident (same per-operation check) = %.2fx" % (rows[2][1]["ms_per_chain"] / rows[0][1]["ms_per_chain"]))


if __name__ == "__main__":
    main()
