"""PCIe-inclusive rate of the header-compatible drop-in (host-resident seal::Ciphertext in, host-resident out) at C5."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_dropin as T, sealref as R
D = T._bind(os.path.join(ROOT, "integration", "_build", "libsealdropin.so"))
n, bits = 65536, [60] + [50] * 14 + [60]
primes = R.coeff_modulus_create(n, bits)
for name, lib in (("drop-in (GPU, 2 PCIe copies per call)", D), ("reference (1 CPU thread)", R)):
    ctx = lib.RefContext("ckks", n, primes)
    ctx.keygen_relin()
    ctx.time_pipeline("ckks_mul_relin_rescale", 1, 1)   # warm-up: tables, key upload
    reps = 8 if lib is D else 2
    s = ctx.time_pipeline("ckks_mul_relin_rescale", 1, reps)
    print("%-40s %.2f ms per ciphertext (%.1f ct/s)" % (name, 1e3 * s / reps, reps / s), flush=True)
