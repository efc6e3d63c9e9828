"""Time-boxed random CKKS / BFV sequences with deferred key-switch tails on the device, compared with seal::Evaluator every few
operations (tests/fuzz_cases.py); every other sequence runs its key switches as one digit group (SEALHIP_KS_SPLIT=1).
usage (GPU box): FUZZ_SECONDS=180 FUZZ_SEED0=300 python tools/quick/fuzz_deferred.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.getcwd(), "tests")); sys.path.insert(0, os.getcwd())
import seal_amd as S, sealref
import fuzz_cases as F, test_fuzz as T
S.load()
t0, budget = time.time(), float(os.environ.get("FUZZ_SECONDS", "180"))
seed = int(os.environ.get("FUZZ_SEED0", "300"))
ok = rej = 0
fails = []
f0 = S.tail_stats()
p0 = S.product_stats()
g0 = S.galois_stats()
while time.time() - t0 < budget:
    cfgs = T._ckks_configs(seed, 4, [8192, 16384, 32768, 65536]) + T._bfv_configs(seed, 2, [8192, 16384])
    for i, cfg in enumerate(cfgs):
        if time.time() - t0 > budget:
            break
        if i % 2:
            os.environ["SEALHIP_KS_SPLIT"] = "1"
            os.environ["SEALHIP_LAZY_PRODUCT_MIN_WGS"] = "0"   # three-object products stay pending at these small batches (round 6)
        else:
            os.environ.pop("SEALHIP_KS_SPLIT", None)
            os.environ.pop("SEALHIP_LAZY_PRODUCT_MIN_WGS", None)
        try:
            F.run_sequence(*cfg, check_prob=0.25, scale0=2.0 ** 30 if cfg[0] == "ckks" else None, three_object_prob=0.6); ok += 1
        except sealref.RefError:
            rej += 1
        except Exception as e:
            fails.append((cfg, repr(e)[:300]))
    seed += 1
f1 = S.tail_stats()
print("sequences ok", ok, "rejected-by-reference", rej, "FAIL", len(fails), "seeds", int(os.environ.get("FUZZ_SEED0", "300")), "..", seed - 1,
      "tails folded / plain / dropped:", [b - a for a, b in zip(f0, f1)],
      "products fused / formed / dropped:", [b - a for a, b in zip(p0, S.product_stats())],
      "rotations gathered / permuted:", [b - a for a, b in zip(g0, S.galois_stats())], "seconds %.0f" % (time.time() - t0))
for f in fails[:10]:
    print(f)
