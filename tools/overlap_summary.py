#!/usr/bin/env python3
"""How much do the device workers' kernels overlap?  Reads a rocprofv3 --kernel-trace database (rocpd SQLite) of a
multi-worker bench run and prints, for the second half of the trace (the timed passes): the wall time, the sum of kernel
durations, the average number of kernels in flight, and per kernel the average duration next to its count.

    rocprofv3 --kernel-trace -d /tmp/ov -o run -- python3 bench.py --headline-only --no-timing --steps 10 --warmup 2
    python3 tools/overlap_summary.py /tmp/ov
"""
import glob
import os
import re
import sqlite3
import sys
from collections import defaultdict


def main(directory):
    db = sqlite3.connect(sorted(glob.glob(os.path.join(directory, "**", "*_results.db"), recursive=True))[-1])
    rows = db.execute("select name, start, end from kernels order by start").fetchall()
    if not rows:
        raise SystemExit("no kernels")
    t0, t1 = rows[0][1], max(r[2] for r in rows)
    mid = t0 + (t1 - t0) * 0.55
    rows = [r for r in rows if r[1] >= mid and "at::native" not in r[0] and "rocprim" not in r[0]]
    wall = max(r[2] for r in rows) - rows[0][1]
    busy = sum(r[2] - r[1] for r in rows)
    events = sorted([(r[1], 1) for r in rows] + [(r[2], -1) for r in rows])
    level, last, hist = 0, events[0][0], defaultdict(int)
    for t, d in events:
        hist[level] += t - last
        last = t
        level += d
    print("wall %.2f ms, sum of kernel durations %.2f ms (%.2fx), kernels in flight: %s" % (
        wall / 1e6, busy / 1e6, busy / wall, ", ".join("%d: %.0f%%" % (k, 100.0 * v / wall) for k, v in sorted(hist.items()))))
    per = defaultdict(lambda: [0, 0])
    for n, s, e in rows:
        k = re.sub(r"\(anonymous namespace\)::|^void |mlsgpu::", "", n)
        k = re.sub(r"\(.*$", "", k)[:60]
        per[k][0] += e - s
        per[k][1] += 1
    for k, (tot, cnt) in sorted(per.items(), key=lambda kv: -kv[1][0])[:14]:
        print("%-62s %8.1f us x %5d = %7.2f ms" % (k, tot / cnt / 1e3, cnt, tot / 1e6))


if __name__ == "__main__":
    main(sys.argv[1])
