#!/usr/bin/env python3
"""Static instruction mix per kernel of a device assembly file (hipcc --cuda-device-only -S).
usage: asm_stats.py file.s [substring of the demangled kernel name]"""
import collections
import re
import subprocess
import sys

txt = open(sys.argv[1]).read().split("\n")
flt = sys.argv[2] if len(sys.argv) > 2 else ""
kernels = {}
cur = None
for ln in txt:
    m = re.match(r"^(_Z\w+):", ln)
    if m:
        cur = m.group(1)
        kernels[cur] = []
        continue
    if ln.startswith("\t.end_amdhsa_kernel") or ln.startswith(".Lfunc_end"):
        cur = None
    if cur and re.match(r"^\t[a-z]", ln):
        kernels[cur].append(ln.strip().split()[0])
names = list(kernels)
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
for nm, d in zip(names, dem):
    d = re.sub(r"^void sealhip::\(anonymous namespace\)::", "", d)
    d = re.sub(r"\(.*$", "", d)
    if flt not in d:
        continue
    ops = kernels[nm]
    c = collections.Counter()
    for o in ops:
        if o.startswith("v_"):
            c["valu"] += 1
        elif o.startswith("s_"):
            c["salu"] += 1
        elif o.startswith("scratch_"):
            c["scratch"] += 1
        elif o.startswith("global_") or o.startswith("buffer_") or o.startswith("flat_"):
            c["vmem"] += 1
        elif o.startswith("ds_"):
            c["lds"] += 1
        else:
            c["other"] += 1
    top = collections.Counter(o for o in ops if o.startswith("v_")).most_common(8)
    print("%-28s total %6d valu %6d salu %5d vmem %4d lds %4d scratch %4d | %s" % (
        d, len(ops), c["valu"], c["salu"], c["vmem"], c["lds"], c["scratch"], " ".join("%s:%d" % (k[2:], v) for k, v in top)))
