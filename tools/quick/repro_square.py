"""repeat one fuzz sequence and a bare square_inplace on the device, looking for run-to-run differences"""
import os, sys
sys.path.insert(0, os.path.join(os.getcwd(), "tests")); sys.path.insert(0, os.getcwd())
import numpy as np
import seal_amd as S, sealref
import fuzz_cases as F
from harness import DeviceSide
from oracle import coeff_modulus_create, rand_ct
S.load()
cfg = ('ckks', 4096, [54, 42, 55], 20, 3, 7, 700004)
bad = 0
for rep in range(int(os.environ.get("REPS", "30"))):
    try:
        F.run_sequence(*cfg)
    except AssertionError as e:
        bad += 1
        print("rep", rep, "FAIL", str(e)[:200])
print("sequence repeats failed:", bad)
# bare square: same input, many times; every result must equal the first
primes = coeff_modulus_create(4096, [54, 42, 55])
d = DeviceSide("ckks", 4096, primes, 0)
rng = np.random.default_rng(5)
slabs = [rand_ct(rng, primes, 2, 4096, size=2) for _ in range(3)]
first = None
diffs = 0
for rep in range(int(os.environ.get("REPS2", "300"))):
    c = d.ct(slabs, scale=2.0 ** 8, is_ntt=True)
    d.ev.square_inplace(c)
    out = c.to_numpy()
    if first is None:
        first = out
    elif not np.array_equal(out, first):
        diffs += 1
        idx = np.argwhere(out != first)
        print("rep", rep, "differs in", len(idx), "words; first", idx[0], "last", idx[-1])
print("bare square runs that differ from the first:", diffs)
