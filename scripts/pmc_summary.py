#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc csv output per kernel: usage pmc_summary.py <dir> [kernel-substring ...]"""
import collections
import csv
import glob
import sys

root = sys.argv[1]
keys = sys.argv[2:] or ["spmv", "axpy2", "ewise", "dot_partials", "cg_"]
for d in sorted(glob.glob(root + "/*/*/*_counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(d)):
        k = row["Kernel_Name"].replace("void cgamd::", "")[:56]
        agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    print("==", d.split("/")[-3])
    for k, v in agg.items():
        if any(t in k for t in keys):
            print("  %-58s" % k, {c: (len(x), round(sum(x) / len(x), 1)) for c, x in v.items()})
