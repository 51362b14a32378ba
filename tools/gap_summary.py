#!/usr/bin/env python3
"""Inter-kernel gaps from a rocprofv3 kernel trace CSV: how much of a chunk step is kernels, how much is the space
between them.  Gaps above 60 us are host turnarounds (between joiner rounds / feeds) and are reported separately.
usage: gap_summary.py <..._kernel_trace.csv>"""
import csv
import statistics
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
busy = sum(e - s for s, e, _ in rows)
gaps = [rows[i + 1][0] - rows[i][1] for i in range(len(rows) - 1)]
small = [g for g in gaps if g < 60000]
big = [g for g in gaps if g >= 60000]
print("kernels %d  busy %.2f ms  span %.2f ms" % (len(rows), busy / 1e6, (rows[-1][1] - rows[0][0]) / 1e6))
print("gaps < 60 us: n=%d  sum %.2f ms  median %.2f us  mean %.2f us  p90 %.2f us" % (
    len(small), sum(small) / 1e6, statistics.median(small) / 1e3, statistics.mean(small) / 1e3, sorted(small)[int(len(small) * 0.9)] / 1e3))
print("gaps >= 60 us (host turnarounds): n=%d  sum %.2f ms" % (len(big), sum(big) / 1e6))
print("kernel time / (kernel time + small gaps) = %.3f" % (busy / (busy + sum(small))))

# The same for the streaming steps alone: session set-up (zero_slot per created session, buffer fills / copies, weight re-packs,
# the decoder table build) is launched eagerly, one small kernel at a time, and its launch gaps say nothing about a feed.
SETUP = ("zero_slot_kernel", "__amd_rocclr", "repack_x32", "cvt_f16", "dec_embed_kernel")
step = [r for r in rows if not any(k in r[2] for k in SETUP)]
if step and len(step) < len(rows):
    sbusy = sum(e - s for s, e, _ in step)
    # a gap belongs to the steps when both neighbours (in the full, time-ordered trace) are step kernels
    sg = [rows[i + 1][0] - rows[i][1] for i in range(len(rows) - 1)
          if not any(k in rows[i][2] for k in SETUP) and not any(k in rows[i + 1][2] for k in SETUP)]
    ssmall = [g for g in sg if g < 60000]
    if ssmall:
        print("streaming steps only (set-up kernels excluded): kernels %d  busy %.2f ms;  gaps < 60 us: n=%d  sum %.2f ms  median %.2f us  mean %.2f us  p90 %.2f us" % (
            len(step), sbusy / 1e6, len(ssmall), sum(ssmall) / 1e6, statistics.median(ssmall) / 1e3, statistics.mean(ssmall) / 1e3, sorted(ssmall)[int(len(ssmall) * 0.9)] / 1e3))
        print("streaming steps only: kernel time / (kernel time + small gaps) = %.3f" % (sbusy / (sbusy + sum(ssmall))))
