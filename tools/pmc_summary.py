#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel: average counter value per dispatch, plus
average duration from the kernel trace.  Usage: pmc_summary.py <dir> [<dir> ...] > summary.txt"""
import collections
import csv
import glob
import sys

for d in sys.argv[1:]:
    for f in glob.glob(d + "/*counter_collection.csv"):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        disp = collections.defaultdict(set)
        dur = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Dispatch_Id"] not in disp[k]:
                disp[k].add(r["Dispatch_Id"])
                dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0
        print("==", f)
        for k in sorted(agg, key=lambda k: -dur[k]):
            n = len(disp[k])
            print("%-70s n=%5d avg_us=%9.2f " % (k[:70], n, dur[k] / n) + " ".join("%s=%.4g" % (c, v / n) for c, v in sorted(agg[k].items())))
