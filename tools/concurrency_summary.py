#!/usr/bin/env python3
"""Kernel concurrency from a rocprofv3 kernel trace CSV: span, busy time (union), summed kernel time, average number of
kernels in flight, per-queue counts.  usage: concurrency_summary.py <..._kernel_trace.csv> [skip_first_ms]"""
import csv, sys, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
skip = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 0
t_first = rows[0][0]
rows = [r for r in rows if r[0] - t_first >= skip]
span = rows[-1][1] - rows[0][0]
summed = sum(e - s for s, e, _, _ in rows)
ev = []
for s, e, _, _ in rows:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
busy = 0; cur = 0; last = ev[0][0]; hist = collections.Counter()
for t, d in ev:
    if cur > 0: busy += t - last
    hist[cur] += t - last
    cur += d; last = t
print("kernels %d  span %.2f ms  union-busy %.2f ms (%.1f%%)  summed %.2f ms  avg in flight while busy %.2f" % (len(rows), span / 1e6, busy / 1e6, 100.0 * busy / span, summed / 1e6, summed / max(1, busy)))
print("time share by kernels in flight:", {k: round(100.0 * v / span, 1) for k, v in sorted(hist.items())})
q = collections.Counter(r[3] for r in rows)
print("kernels per queue:", dict(q))
for qid in sorted(q):      # per queue (= per stream here): busy time and the gaps between consecutive kernels
    rs = [r for r in rows if r[3] == qid]
    b = sum(e - s for s, e, _, _ in rs)
    gaps = [max(0, rs[i + 1][0] - rs[i][1]) for i in range(len(rs) - 1)]
    small = [g for g in gaps if g < 50000]
    print("  queue %s: %d kernels, busy %.2f ms, active span %.2f ms, gaps < 50 us: sum %.2f ms (mean %.2f us), larger gaps: %d sum %.2f ms" % (
        qid, len(rs), b / 1e6, (rs[-1][1] - rs[0][0]) / 1e6, sum(small) / 1e6, (sum(small) / max(1, len(small))) / 1e3, len(gaps) - len(small), (sum(gaps) - sum(small)) / 1e6))
names = collections.defaultdict(lambda: [0, 0])
for s, e, n, _ in rows:
    names[n][0] += 1; names[n][1] += e - s
for n, (c, t) in sorted(names.items(), key=lambda x: -x[1][1])[:16]:
    print("  %7d x %8.2f us  %s" % (c, t / c / 1e3, n[:110]))
