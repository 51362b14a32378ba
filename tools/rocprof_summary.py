#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace result (rocpd sqlite .db) as a per-kernel CSV:
name, calls, total_us, avg_us, min_us, max_us, percent.  Usage: rocprof_summary.py results.db [out.csv]"""
import csv
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = list(db.execute("select name, count(*), sum(end-start)/1000.0, avg(end-start)/1000.0, min(end-start)/1000.0, "
                           "max(end-start)/1000.0 from kernels group by name order by 3 desc"))
    total = sum(r[2] for r in rows) or 1.0
    out = open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout
    w = csv.writer(out)
    w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "MinUs", "MaxUs", "Percentage"])
    for r in rows:
        w.writerow([r[0], r[1], "%.2f" % r[2], "%.3f" % r[3], "%.3f" % r[4], "%.3f" % r[5], "%.2f" % (100 * r[2] / total)])


if __name__ == "__main__":
    main()
