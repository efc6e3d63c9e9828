"""random sequences at N = 4096 / 2048 / 8192 only (the size class of the one fuzz failure of round 3), new context per sequence"""
import os, sys
sys.path.insert(0, os.path.join(os.getcwd(), "tests")); sys.path.insert(0, os.getcwd())
import numpy as np
import seal_amd as S, sealref
import fuzz_cases as F
S.load()
rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
ok = 0
fails = []
for it in range(int(os.environ.get("SEQS", "300"))):
    n = int(rng.choice([2048, 4096, 4096, 4096, 8192]))
    L = int(rng.integers(2, 5))
    bits = [int(b) for b in rng.integers(36, 59, L)]
    while sum(bits) > {2048: 54 * 2, 4096: 109, 8192: 218}[n] * 2:   # stay well inside what the parameter check accepts
        bits = bits[:-1]
    if len(bits) < 2:
        bits = [40, 41]
    cfg = ("ckks", n, bits, 20, int(rng.integers(1, 5)), int(rng.integers(1, 5)), 900000 + it)
    try:
        F.run_sequence(*cfg)
        ok += 1
    except sealref.RefError:
        pass
    except (S.InvalidArgument, S.LogicError):
        pass
    except AssertionError as e:
        fails.append((cfg, str(e)[:700]))
print("ok", ok, "FAIL", len(fails))
for f in fails[:5]:
    print(f)
