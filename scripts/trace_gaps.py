#!/usr/bin/env python3
"""kernel durations and inter-kernel gaps from a rocprofv3 kernel trace: usage trace_gaps.py <dir> [name-substring]"""
import collections
import csv
import glob
import statistics
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv"):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
key = sys.argv[2] if len(sys.argv) > 2 else "cgamd"
dur, gap_after = collections.defaultdict(list), collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
    n = a["Kernel_Name"].replace("void cgamd::", "")[:48]
    if key in a["Kernel_Name"]:
        dur[n].append(int(a["End_Timestamp"]) - int(a["Start_Timestamp"]))
        if key in b["Kernel_Name"]:
            gap_after[n].append(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]))
for n in dur:
    if len(dur[n]) < 20:
        continue
    g = gap_after[n]
    print(f"{n:50s} calls {len(dur[n]):6d}  dur us median {statistics.median(dur[n]) / 1e3:7.2f} mean {statistics.mean(dur[n]) / 1e3:7.2f}"
          f"   gap to next us median {statistics.median(g) / 1e3 if g else 0:6.2f}")
