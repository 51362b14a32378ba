#!/usr/bin/env python3
"""usage: tools/trace_by_grid.py <kernel_trace.csv> [name filter]: launches and mean duration of every (kernel, grid size) pair of a
rocprofv3 --kernel-trace CSV -- z-batched kernels run 1..5 problems under one name; the grid size tells them apart"""
import csv
import sys
from collections import defaultdict

acc = defaultdict(lambda: [0, 0.0, 1e30, 0.0])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for r in csv.DictReader(open(sys.argv[1])):
    name = r.get("Kernel_Name") or r.get("Name")
    if flt and flt not in name:
        continue
    grid = (r.get("Grid_Size_X") or r.get("Grid_Size"), r.get("Grid_Size_Y"), r.get("Grid_Size_Z"))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = acc[(name[:70], grid)]
    a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
for (name, grid), a in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print("%-70s grid %-22s n %5d  mean %8.2f us  min %8.2f  max %8.2f" % (name, "x".join(str(g) for g in grid), a[0], a[1] / a[0], a[2], a[3]))
