#!/usr/bin/env python3
"""One feed as a timeline, from a rocprofv3 kernel trace CSV: every kernel between two fbank launches with its start offset,
duration, gap to the predecessor and grid size; then per-kernel-name sums for the feed.  The last complete 2-chunk and 3-chunk
feeds of the trace are printed (a feed's chunk count = its number of first-round decide launches is not known here, so feeds are
told apart by their kernel count).
usage: feed_timeline.py <..._kernel_trace.csv> [n_feeds_from_the_end=2]"""
import csv
import sys
from collections import defaultdict


def short(name):
    n = name.replace("void aprilx::", "").replace("aprilx::", "").replace("(anonymous namespace)::", "")
    cut = n.find("(")
    return n[:cut] if cut > 0 else n


rows = []
for r in csv.DictReader(open(sys.argv[1])):
    g = [int(r.get(k, 0) or 0) for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z")]
    w = [int(r.get(k, 1) or 1) for k in ("Workgroup_Size_X", "Workgroup_Size_Y", "Workgroup_Size_Z")]
    wgs = 1
    for a, b in zip(g, w):
        wgs *= max(1, a // max(1, b))
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), wgs))
rows.sort()
starts = [i for i, r in enumerate(rows) if r[2].startswith("fbank_kernel")]
feeds = [rows[a:b] for a, b in zip(starts, starts[1:])]
nshow = int(sys.argv[2]) if len(sys.argv) > 2 else 2
seen = set()
for f in reversed(feeds):
    if len(f) in seen:
        continue
    seen.add(len(f))
    t0 = f[0][0]
    busy = sum(e - s for s, e, _, _ in f)
    print("==== feed of %d kernels: span %.1f us (first start -> last end), busy %.1f us" % (len(f), (f[-1][1] - t0) / 1e3, busy / 1e3))
    prev = None
    for s, e, n, w in f:
        print("%9.1f  %7.2f  gap %6.2f  wgs %5d  %s" % ((s - t0) / 1e3, (e - s) / 1e3, 0.0 if prev is None else (s - prev) / 1e3, w, n))
        prev = e
    by = defaultdict(lambda: [0, 0.0])
    for s, e, n, w in f:
        by[n][0] += 1
        by[n][1] += (e - s) / 1e3
    for n, (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1]):
        print("   %-70s %3d  %8.1f us  %5.1f %%" % (n, c, t, 100 * t / (busy / 1e3)))
    if len(seen) >= nshow:
        break
