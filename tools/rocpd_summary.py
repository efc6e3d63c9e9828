#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database (the default ROCm 7.2 output) as text.

usage: rocpd_summary.py <results.db> [--pmc]
  default : per-kernel statistics, the same table `rocprofv3 --kernel-trace --stats` prints
            (name, calls, total ms, average us, share) plus launch geometry and register use
  --pmc   : per-kernel average of every collected counter (FETCH_SIZE / WRITE_SIZE are in KiB;
            on gfx950 FETCH_SIZE counts 64 B per 128-B request, i.e. half of the bytes read —
            MI355X_MICROARCH.md §HBM — the doubled figure is printed next to it)
"""
import sqlite3
import sys


def short(name):
    name = name.replace("sealhip::(anonymous namespace)::", "").replace("void ", "")
    cut = name.find("(")
    return (name[:cut] if cut > 0 else name)[:70]


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    if "--pmc" in sys.argv:
        rows = cur.execute(
            "select kernel_name, counter_name, grid_size, count(*), avg(value), avg(duration) from counters_collection "
            "group by kernel_name, counter_name, grid_size order by kernel_name").fetchall()
        print("%-70s %-12s %10s %6s %14s %14s" % ("kernel", "counter", "grid", "calls", "avg value", "x2 (FETCH)"))
        for name, ctr, grid, n, val, dur in rows:
            print("%-70s %-12s %10d %6d %14.1f %14s" % (short(name), ctr, grid, n, val,
                                                       "%.1f" % (2 * val) if ctr == "FETCH_SIZE" else ""))
        return
    rows = cur.execute(
        "select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), "
        "max(vgpr_count), max(sgpr_count), max(lds_size), max(workgroup_x) from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    print("%-70s %6s %11s %11s %11s %11s %6s %5s %5s %7s %5s" % (
        "kernel", "calls", "total ms", "avg us", "min us", "max us", "%", "vgpr", "sgpr", "lds B", "wg"))
    for name, n, total, avg, mn, mx, vg, sg, lds, wg in rows:
        print("%-70s %6d %11.3f %11.2f %11.2f %11.2f %6.1f %5d %5d %7d %5d" % (
            short(name), n, total / 1e6, avg / 1e3, mn / 1e3, mx / 1e3, 100.0 * total / tot, vg, sg, lds, wg))


if __name__ == "__main__":
    main()
