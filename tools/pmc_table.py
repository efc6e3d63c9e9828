#!/usr/bin/env python
"""Counter table of the NTT roofline launch (bench.py's roofline leg: ntt_forward over 2*batch*K transforms of 2^16):
where every byte goes at the L2 <-> fabric boundary and what the waves do meanwhile.

One rocprofv3 pass per counter group (the PMC slots do not hold more, MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE alone
need separate passes) over `bench.py --pmc-child` (no torch: the launch only).  Values are per launch (sum over the
dispatches of a kernel / number of ntt_forward calls).  FETCH_SIZE is doubled (gfx950 tallies 128-B requests at 64 B).
There is no Infinity-Cache (MALL) counter in this rocprofv3 (`rocprofv3 -L`, kept in profiles/r02_counters_available.txt):
requests that hit it are counted as fabric requests like those that go to HBM; what the MALL absorbs is measured by time
(tools/microbench/flow_handoff.hip: two kernels over chunks whose intermediate fits it).

usage: python tools/pmc_table.py [--batch 256] > profiles/r02_ntt_counters.txt"""
import argparse
import glob
import os
import shutil
import sqlite3
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GROUPS = [
    ["FETCH_SIZE"],
    ["WRITE_SIZE"],
    ["TCC_HIT_sum", "TCC_MISS_sum"],
    ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum"],
    ["TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum"],
    ["TCC_REQ_sum", "TCC_READ_sum", "TCC_WRITE_sum"],
    ["TCP_TCC_READ_REQ_sum", "TCP_TCC_WRITE_REQ_sum"],
    ["SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU"],
    ["SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"],
    ["GRBM_GUI_ACTIVE"],
    ["SQ_WAVE_CYCLES", "SQ_WAIT_INST_LDS", "SQ_INST_LEVEL_LDS", "SQ_INST_LEVEL_VMEM", "SQ_IFETCH", "SQ_IFETCH_LEVEL", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC"],
    ["SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_FLAT", "SQ_LDS_CMD_FIFO_FULL", "SQ_LDS_DATA_FIFO_FULL", "SQ_LDS_ADDR_CONFLICT", "SQ_LDS_UNALIGNED_STALL", "SQ_INST_CYCLES_VMEM_RD"],
]
CALLS = 3  # bench.py: PMC_CALLS


def short(name):
    name = name.replace("sealhip::(anonymous namespace)::", "").replace("void ", "")
    cut = name.find("(")
    return name[:cut] if cut > 0 else name


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--bench-args", default="", help="profile `python bench.py <args>` instead of the NTT launch, e.g. "
                    "'--batch 64 --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --no-verify'")
    ap.add_argument("--filter", default="ntt", help="substring of the kernel names to tabulate")
    ap.add_argument("--groups", default="", help="comma-separated indices of the counter groups to run (default: all)")
    args = ap.parse_args()
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    table = {}   # kernel -> counter -> per-launch value
    durs = {}
    tmp = tempfile.mkdtemp(prefix="pmc_table_", dir="/tmp")
    calls = CALLS
    which = [int(x) for x in args.groups.split(",")] if args.groups else list(range(len(GROUPS)))
    for i, grp in enumerate(GROUPS):
        if i not in which:
            continue
        out = os.path.join(tmp, "g%d" % i)
        if args.bench_args:
            target = [sys.executable, os.path.join(ROOT, "bench.py")] + args.bench_args.split()
            calls = 1
        else:
            target = [sys.executable, os.path.join(ROOT, "bench.py"), "--pmc-child", "--batch", str(args.batch)]
        cmd = [exe, "--kernel-trace", "--pmc"] + grp + ["-d", out, "-o", "r", "--"] + target
        p = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=600)
        dbs = glob.glob(os.path.join(out, "**", "*.db"), recursive=True)
        if p.returncode != 0 or not dbs:
            print("# group %s failed: %s" % (grp, (p.stderr or p.stdout)[-200:]))
            continue
        cur = sqlite3.connect(dbs[0]).cursor()
        for name, ctr, cnt, val, dur in cur.execute(
                "select kernel_name, counter_name, count(*), sum(value), avg(duration) from counters_collection group by kernel_name, counter_name"):
            if args.filter not in name:
                continue
            k = short(name)
            table.setdefault(k, {})[ctr] = float(val) / (calls if not args.bench_args else cnt)
            durs[k] = dur
    shutil.rmtree(tmp, ignore_errors=True)
    n, K = 65536, 15
    transforms = 2 * args.batch * K
    if args.bench_args:
        print("# python bench.py %s : kernels matching %r, values per DISPATCH (average)" % (args.bench_args, args.filter))
    else:
        print("# ntt_forward over %d transforms of 2^16 (batch %d): algorithmic bytes %.3f GB per launch (16 N per transform)" % (
            transforms, args.batch, 16.0 * n * transforms / 1e9))
        print("# per launch, per kernel; kernel durations under the profiler (ns, averaged over the PMC passes' dispatches)")
    for k in sorted(table):
        t = table[k]
        print("\n%s   (avg duration %.0f us)" % (k, durs.get(k, 0) / 1e3))
        if "FETCH_SIZE" in t:
            print("  fabric reads   FETCH_SIZE x2      %10.3f GB" % (2 * t["FETCH_SIZE"] * 1024 / 1e9))
        if "WRITE_SIZE" in t:
            print("  fabric writes  WRITE_SIZE         %10.3f GB" % (t["WRITE_SIZE"] * 1024 / 1e9))
        for c in sorted(t):
            if c in ("FETCH_SIZE", "WRITE_SIZE"):
                continue
            print("  %-28s %16.0f" % (c, t[c]))
        if "TCC_HIT_sum" in t and "TCC_MISS_sum" in t and t["TCC_HIT_sum"] + t["TCC_MISS_sum"] > 0:
            print("  L2 hit rate                  %16.3f" % (t["TCC_HIT_sum"] / (t["TCC_HIT_sum"] + t["TCC_MISS_sum"])))
        if "SQ_WAVE_CYCLES" in t and t["SQ_WAVE_CYCLES"] > 0:
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"):
                if c in t:
                    print("  %-28s %15.1f %% of wave cycles" % (c, 100.0 * t[c] / t["SQ_WAVE_CYCLES"]))


if __name__ == "__main__":
    main()
