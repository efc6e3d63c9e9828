#!/usr/bin/env python
"""Time line of ONE bench step from a rocprofv3 --kernel-trace database: every dispatch of the last timed step in start
order with its stream (queue), duration, the idle gap since the previous dispatch ended on any queue, and a per-kernel sum.
The step is found as the last `ckks_multiply_2x2_kernel` (headline) dispatch and everything up to the next one / the end.

usage: rocprofv3 --kernel-trace -d <dir> -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pmc --no-verify
       python tools/step_timeline.py <results.db> [--anchor ckks_multiply_2x2_kernel] [--step -2]
"""
import argparse
import sqlite3


def short(name):
    name = name.replace("sealhip::(anonymous namespace)::", "").replace("void ", "")
    cut = name.find("(")
    return (name[:cut] if cut > 0 else name)[:48]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--anchor", default="ckks_multiply_2x2")
    ap.add_argument("--step", type=int, default=-2, help="index of the anchor dispatch that starts the step (default: the one before last)")
    ap.add_argument("--tail", type=int, default=0, help="no anchor: list the last TAIL dispatches of the trace (workloads without the headline's first kernel)")
    args = ap.parse_args()
    db = sqlite3.connect(args.db)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
    rows = db.execute("select name, start, end, %s from kernels order by start" % q).fetchall()
    anchors = [i for i, r in enumerate(rows) if args.anchor in r[0]]
    if len(anchors) < 2 and not args.tail:
        # round 6: with the tensor product deferred into the relinearisation the headline step has no kernel of its own at its start;
        # a window from one step's ks_last_coeff_kernel to the next holds the same dispatches (the folded tail first, the key switch after)
        anchors = [i for i, r in enumerate(rows) if "ks_last_coeff_kernel" in r[0]]
        if len(anchors) >= 2:
            print("# (no %s in the trace: the window runs from one step's ks_last_coeff_kernel to the next)" % args.anchor)
    if args.tail:
        a0, a1 = max(0, len(rows) - args.tail), len(rows)
    elif len(anchors) < 2:
        raise SystemExit("need at least two dispatches of %s" % args.anchor)
    else:
        a0 = anchors[args.step]
        later = [a for a in anchors if a > a0]
        a1 = later[0] if later else len(rows)
    step = rows[a0:a1]
    t0 = step[0][1]
    busy_end = t0
    sums = {}
    print("%-48s %6s %10s %10s %9s" % ("kernel", "queue", "start us", "dur us", "gap us"))
    idle = 0.0
    for name, s, e, qid in step:
        gap = max(0, s - busy_end) / 1e3
        idle += gap
        print("%-48s %6s %10.1f %10.1f %9.1f" % (short(name), qid, (s - t0) / 1e3, (e - s) / 1e3, gap))
        busy_end = max(busy_end, e)
        k = short(name)
        sums[k] = sums.get(k, 0.0) + (e - s) / 1e3
    span = (busy_end - t0) / 1e3
    nxt = (rows[a1][1] - t0) / 1e3 if a1 < len(rows) else span
    print("\nstep span %.1f us (to the next step's first dispatch: %.1f us), idle between dispatches %.1f us, %d dispatches" % (span, nxt, idle, len(step)))
    print("\n%-48s %10s %6s" % ("kernel", "sum us", "%"))
    for k, v in sorted(sums.items(), key=lambda kv: -kv[1]):
        print("%-48s %10.1f %6.1f" % (k, v, 100.0 * v / span))


if __name__ == "__main__":
    main()
