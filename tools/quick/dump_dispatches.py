#!/usr/bin/env python
"""dispatches of a rocprofv3 --kernel-trace database whose name contains PATTERN, in start order: queue, start, duration, gap"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); pat = sys.argv[2]; limit = int(sys.argv[3]) if len(sys.argv) > 3 else 60
rows = db.execute("select name, start, end, queue_id from kernels order by start").fetchall()
sel = [r for r in rows if pat in r[0]]
t0 = sel[0][1]; prev_end = t0
for name, s, e, q in sel[:limit]:
    nm = name.replace("sealhip::(anonymous namespace)::", "").replace("void ", "")
    nm = nm[:nm.find("(")] if "(" in nm else nm
    print("%-34s q%-3s start %10.1f us  dur %8.1f us  gap %8.1f" % (nm[:34], q, (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3))
    prev_end = max(prev_end, e)
