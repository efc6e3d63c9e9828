"""Wire-format rates at the headline parameters (CKKS N = 65536, {60, 14x50, 60}) on the GPU box: Ciphertext_Load / _Save of one
ciphertext (15 MiB of coefficients) and KSwitchKeys_Load of a RelinKeys stream, seeded (126 MB stored, 126 MB expanded on the host
cores with BLAKE2Xb) and full (252 MB), next to the reference's own load of the same bytes (oracle/_ref, checker/baseline only)."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import seal_amd as S
import sealref
from harness import DeviceSide

n, bits = 65536, [60] + [50] * 14 + [60]
primes = sealref.coeff_modulus_create(n, bits)
ref = sealref.RefContext("ckks", n, primes, 0)
d = DeviceSide("ckks", n, primes, 0)


def timed(fn, reps=3):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    S.device_synchronize()
    return (time.perf_counter() - t0) / reps


for seeded in (True, False):
    data = ref.encrypt_zero_symmetric_save(ref.first_chain_index, seeded)
    ct = S.Ciphertext(d.ctx)
    t = timed(lambda: ct.load_bytes(data))
    tu = timed(lambda: ct.load_bytes(data, unsafe=True))
    tr = timed(lambda: ref.ct_load(data))
    print("ciphertext %-6s stream %6.1f MB: Ciphertext_Load %7.2f ms (UnsafeLoad %7.2f ms)   reference load %7.2f ms" % (
        "seeded" if seeded else "full", len(data) / 1e6, t * 1e3, tu * 1e3, tr * 1e3), flush=True)
ts = timed(lambda: ct.save_bytes())
print("ciphertext save (device -> %5.1f MB stream): %7.2f ms" % (ct.save_size() / 1e6, ts * 1e3), flush=True)
for seeded in (True, False):
    data = ref.keys_save("relin", seeded)
    rlk = S.RelinKeys(d.ctx)
    t = timed(lambda: rlk.load_bytes(data), reps=2)
    tu = timed(lambda: rlk.load_bytes(data, unsafe=True), reps=2)
    tr = timed(lambda: ref.keys_load(data), reps=2)
    print("RelinKeys  %-6s stream %6.1f MB: KSwitchKeys_Load %7.1f ms (UnsafeLoad %7.1f ms)   reference load %7.1f ms   [%d host threads]" % (
        "seeded" if seeded else "full", len(data) / 1e6, t * 1e3, tu * 1e3, tr * 1e3, os.cpu_count()), flush=True)
