#!/usr/bin/env python3
"""Table of hipcc's kernel-resource-usage remarks: VGPRs, AGPRs, scratch, waves/SIMD, LDS per kernel.
usage: hipcc ... -Rpass-analysis=kernel-resource-usage -c x.hip 2> remarks.txt; tools/quick/resources.py remarks.txt [filter]"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
names = [b.split("\n")[0].strip() for b in blocks]
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
KEYS = ["VGPRs", "AGPRs", r"ScratchSize \[bytes/lane\]", r"Occupancy \[waves/SIMD\]", r"LDS Size \[bytes/block\]"]
print("vgpr agpr scratch occ     lds  kernel")
for b, d in zip(blocks, dem):
    d = re.sub(r"^void sealhip::\(anonymous namespace\)::", "", d)
    d = re.sub(r"\(.*$", "", d)
    if flt and flt not in d:
        continue
    v = []
    for k in KEYS:
        m = re.search(k + r": (\d+)", b)
        v.append(m.group(1) if m else "?")
    print("%4s %4s %7s %3s %7s  %s" % (v[0], v[1], v[2], v[3], v[4], d))
