#!/usr/bin/env python3
"""Per-launch averages of rocprofv3 --pmc counters for one kernel.

    python tools/profile/summarize_pmc.py <rocprof output dir> [kernel substring] [skip first N dispatches | -N = keep only the last N]
"""
import collections
import csv
import glob
import json
import sys

d = sys.argv[1]
kern = sys.argv[2] if len(sys.argv) > 2 else "c4_step_kernel"
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
fs = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)
acc, order = collections.defaultdict(float), []
for row in csv.DictReader(open(fs[0])):
    if kern not in row["Kernel_Name"]:
        continue
    did = row["Dispatch_Id"]
    if did not in order:
        order.append(did)
keep = set(order[skip:])   # (a negative N keeps the last N: Python's slice)
for row in csv.DictReader(open(fs[0])):
    if kern in row["Kernel_Name"] and row["Dispatch_Id"] in keep:
        acc[row["Counter_Name"]] += float(row["Counter_Value"])
n = max(1, len(keep))
print(json.dumps({"kernel": kern, "dispatches_averaged": len(keep), **{k: v / n for k, v in sorted(acc.items())}}))
