#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: mean per dispatch of each counter,
per kernel name.  usage: pmc_summary.py <prof dir>"""
import csv, glob, os, sys, collections
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(d, "p*", "*", "*_counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"][:60]
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print("   %-26s n=%3d mean=%.6g" % (c, len(v), sum(v) / len(v)))
