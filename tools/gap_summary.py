#!/usr/bin/env python3
"""Inter-kernel gaps from a rocprofv3 kernel trace CSV: how much of the GPU's time under a step is kernels, how much is the space
between them.  The engine runs three streams (front end, layer chain, search) that overlap, so "busy" is the UNION of the kernels'
intervals (two kernels running side by side count once) and a gap is a stretch in which NO kernel of any stream runs: the ratio
busy / (busy + gaps) cannot exceed 1.  Gaps above 60 us are host turnarounds (between feeds) and are reported separately.  A
per-queue breakdown (rocprofv3's Queue_Id = one HIP stream) follows: each stream's own kernel time, its own idle gaps, and how
much of its kernel time ran beside another stream's kernels.
usage: gap_summary.py <..._kernel_trace.csv>"""
import csv
import statistics
import sys

SETUP = ("zero_slot_kernel", "__amd_rocclr", "repack_x32", "cvt_f16", "dec_embed_kernel")


def union(iv):
    """merged, sorted list of [start, end) intervals"""
    out = []
    for s, e in sorted(iv):
        if out and s <= out[-1][1]:
            if e > out[-1][1]:
                out[-1][1] = e
        else:
            out.append([s, e])
    return out


def report(label, rows):
    merged = union([(s, e) for s, e, _, _ in rows])
    busy = sum(e - s for s, e in merged)
    summed = sum(e - s for s, e, _, _ in rows)
    gaps = [merged[i + 1][0] - merged[i][1] for i in range(len(merged) - 1)]
    small = [g for g in gaps if g < 60000]
    big = [g for g in gaps if g >= 60000]
    print("%skernels %d  busy (union over streams) %.2f ms  sum of kernel durations %.2f ms  span %.2f ms" % (
        label, len(rows), busy / 1e6, summed / 1e6, (merged[-1][1] - merged[0][0]) / 1e6))
    if small:
        print("%sgaps < 60 us (no kernel of any stream running): n=%d  sum %.2f ms  median %.2f us  mean %.2f us  p90 %.2f us" % (
            label, len(small), sum(small) / 1e6, statistics.median(small) / 1e3, statistics.mean(small) / 1e3, sorted(small)[int(len(small) * 0.9)] / 1e3))
    print("%sgaps >= 60 us (host turnarounds): n=%d  sum %.2f ms" % (label, len(big), sum(big) / 1e6))
    print("%sbusy / (busy + small gaps) = %.3f" % (label, busy / (busy + sum(small)) if busy else 0.0))
    return merged


rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")))
rows.sort()
report("", rows)

# The same for the streaming steps alone: session set-up (zero_slot per created session, buffer fills / copies, weight re-packs,
# the decoder table build) is launched eagerly, one small kernel at a time, and its launch gaps say nothing about a feed.
step = [r for r in rows if not any(k in r[2] for k in SETUP)]
if step and len(step) < len(rows):
    # a gap belongs to the steps when it lies between two step kernels with no set-up kernel in between: cut the trace at set-up kernels
    segs, cur = [], []
    for r in rows:
        if any(k in r[2] for k in SETUP):
            if cur:
                segs.append(cur)
            cur = []
        else:
            cur.append(r)
    if cur:
        segs.append(cur)
    busy = 0
    small = []
    for seg in segs:
        m = union([(s, e) for s, e, _, _ in seg])
        busy += sum(e - s for s, e in m)
        small += [g for g in (m[i + 1][0] - m[i][1] for i in range(len(m) - 1)) if g < 60000]
    if small:
        print("streaming steps only (set-up kernels excluded): kernels %d  busy (union) %.2f ms;  gaps < 60 us: n=%d  sum %.2f ms  median %.2f us  mean %.2f us  p90 %.2f us" % (
            len(step), busy / 1e6, len(small), sum(small) / 1e6, statistics.median(small) / 1e3, statistics.mean(small) / 1e3, sorted(small)[int(len(small) * 0.9)] / 1e3))
        print("streaming steps only: busy / (busy + small gaps) = %.3f" % (busy / (busy + sum(small))))

# per queue (= HIP stream): own kernel time, own idle gaps, and the share of its kernel time during which another queue was running too
queues = sorted({r[3] for r in step})
if len(queues) > 1:
    print("per queue (streaming-step kernels):")
    for q in queues:
        mine = [r for r in step if r[3] == q]
        others = union([(s, e) for s, e, _, qq in step if qq != q])
        own = union([(s, e) for s, e, _, _ in mine])
        own_busy = sum(e - s for s, e in own)
        og = [g for g in (own[i + 1][0] - own[i][1] for i in range(len(own) - 1)) if g < 60000]
        # overlap of `own` with `others` (both sorted, disjoint)
        ov, j = 0, 0
        for s, e in own:
            while j < len(others) and others[j][1] <= s:
                j += 1
            k = j
            while k < len(others) and others[k][0] < e:
                ov += min(e, others[k][1]) - max(s, others[k][0])
                k += 1
        print("  queue %s: kernels %d  kernel time %.2f ms  own gaps < 60 us %.2f ms  own busy / (busy + gaps) %.3f  beside another queue's kernels %.1f %%" % (
            q, len(mine), own_busy / 1e6, sum(og) / 1e6, own_busy / (own_busy + sum(og)) if own_busy else 0.0, 100.0 * ov / own_busy if own_busy else 0.0))
