import sys, os, traceback
sys.path.insert(0, os.path.join(os.getcwd(), "tests")); sys.path.insert(0, os.getcwd())
import seal_amd as S, sealref
import fuzz_cases as F, test_fuzz as T
S.load()
ok = rej = 0
fails = []
for seed in range(int(os.environ.get("FUZZ_SEED0", "100")), int(os.environ.get("FUZZ_SEED0", "100")) + int(os.environ.get("FUZZ_SEEDS", "12"))):
    for cfg in T._configs(seed, 30, [16, 128, 1024, 4096, 8192, 16384, 32768, 65536]):
        if cfg[1] >= 32768 and len(cfg[2]) > 4:
            cfg = cfg[:2] + (cfg[2][:4],) + cfg[3:]
        try:
            F.run_sequence(*cfg, wild_prob=float(os.environ.get("FUZZ_WILD", "0"))); ok += 1
        except sealref.RefError:
            rej += 1
        except Exception as e:
            fails.append((cfg, repr(e)[:300]))
print("ok", ok, "rejected-by-reference", rej, "FAIL", len(fails))
for f in fails[:10]:
    print(f)
