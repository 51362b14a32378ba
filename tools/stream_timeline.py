#!/usr/bin/env python3
"""Kernel trace of a multi-stream run as lanes: for a time window, every kernel with start offset, duration, queue (stream) and
grid size, in start order; then per-queue busy time and the overlap between queues.
usage: stream_timeline.py <..._kernel_trace.csv> [window_start_frac=0.8] [window_ms=4]"""
import csv
import sys
from collections import defaultdict


def short(name):
    n = name.replace("void aprilx::", "").replace("aprilx::", "").replace("(anonymous namespace)::", "")
    cut = n.find("(")
    return (n[:cut] if cut > 0 else n)[:60]


rows = []
for r in csv.DictReader(open(sys.argv[1])):
    g = [int(r.get(k, 0) or 0) for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z")]
    w = [int(r.get(k, 1) or 1) for k in ("Workgroup_Size_X", "Workgroup_Size_Y", "Workgroup_Size_Z")]
    wgs = 1
    for a, b in zip(g, w):
        wgs *= max(1, a // max(1, b))
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), wgs, r.get("Queue_Id", "?")))
rows.sort()
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.8
win = float(sys.argv[3]) * 1e6 if len(sys.argv) > 3 else 4e6
t_lo = rows[0][0] + int((rows[-1][1] - rows[0][0]) * frac)
# start the window at an fbank launch
for s, e, n, w, q in rows:
    if s >= t_lo and n.startswith("fbank_kernel"):
        t_lo = s
        break
sel = [r for r in rows if t_lo <= r[0] < t_lo + win]
queues = sorted({r[4] for r in sel})
lane = {q: i for i, q in enumerate(queues)}
print("queues:", queues)
for s, e, n, w, q in sel:
    print("%9.1f %7.2f  %s%-4s wgs %5d  %s" % ((s - t_lo) / 1e3, (e - s) / 1e3, "        " * lane[q], "Q" + str(lane[q]), w, n))
busy = defaultdict(float)
for s, e, n, w, q in sel:
    busy[q] += (e - s) / 1e3
span = (max(r[1] for r in sel) - t_lo) / 1e3
print("window span %.1f us; busy per queue: %s" % (span, {"Q%d" % lane[q]: round(v, 1) for q, v in busy.items()}))
# union of busy intervals
iv = sorted((s, e) for s, e, _, _, _ in sel)
u = 0
cs, ce = iv[0]
for s, e in iv[1:]:
    if s > ce:
        u += ce - cs
        cs, ce = s, e
    else:
        ce = max(ce, e)
u += ce - cs
print("union busy %.1f us of %.1f us span (%.3f)" % (u / 1e3, span, u / 1e3 / span))
