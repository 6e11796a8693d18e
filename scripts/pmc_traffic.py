#!/usr/bin/env python3
"""profiles/<round>_pmc_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py.
Corrections per MI355X_MICROARCH.md §HBM: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts each
128-B read request of a wide coalesced stream as 64 B, i.e. reports exactly half the bytes -> doubled here
(re-checked on this pool with a known-bytes stream: profiles/r1_experiments/pmc_r1a_*); WRITE_SIZE is exact."""
import collections
import csv
import glob
import json
import sys

root, kernel, out = sys.argv[1], sys.argv[2], sys.argv[3]
vals = collections.defaultdict(list)
for f in glob.glob(root + "/*/*/*_counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        if kernel in row["Kernel_Name"]:
            vals[row["Counter_Name"]].append(float(row["Counter_Value"]))
fetch = sum(vals["FETCH_SIZE"]) / len(vals["FETCH_SIZE"])
write = sum(vals["WRITE_SIZE"]) / len(vals["WRITE_SIZE"])
res = {"kernel": kernel, "launches_averaged": len(vals["FETCH_SIZE"]), "FETCH_SIZE_KiB_raw": fetch, "WRITE_SIZE_KiB_raw": write,
       "read_bytes": fetch * 1024 * 2, "write_bytes": write * 1024,
       "hbm_bytes_per_launch": fetch * 1024 * 2 + write * 1024,
       "correction": "FETCH_SIZE x2 (gfx950 counts 128-B read requests as 64 B), WRITE_SIZE exact"}
json.dump(res, open(out, "w"), indent=1)
print(res)
