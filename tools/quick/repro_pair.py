"""the first configurations of fuzz_stress seed 700 in order, repeated: does the failure of 700004 depend on what ran before it?"""
import os, sys
sys.path.insert(0, os.path.join(os.getcwd(), "tests")); sys.path.insert(0, os.getcwd())
import numpy as np
import seal_amd as S, sealref
import fuzz_cases as F, test_fuzz as T
S.load()
cfgs = list(T._configs(700, 30, [16, 128, 1024, 4096, 8192, 16384, 32768, 65536]))[:6]
for c in cfgs:
    print(c)
for rep in range(int(os.environ.get("REPS", "12"))):
    for i, cfg in enumerate(cfgs):
        if cfg[1] >= 32768 and len(cfg[2]) > 4:
            cfg = cfg[:2] + (cfg[2][:4],) + cfg[3:]
        try:
            log = F.run_sequence(*cfg)
        except AssertionError as e:
            print("rep", rep, "cfg", i, "FAIL", str(e)[:300])
        except sealref.RefError as e:
            print("rep", rep, "cfg", i, "ref rejected", e)
        else:
            if rep == 0:
                print("cfg", i, "ops:", " > ".join(log))
print("done")
