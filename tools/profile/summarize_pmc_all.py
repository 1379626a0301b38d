#!/usr/bin/env python3
"""Per-kernel, per-launch averages of every counter in a rocprofv3 --pmc output directory
(the first `skip` dispatches of each kernel are warm-up).
    python tools/profile/summarize_pmc_all.py <dir> [skip=5] [only kernels containing ...]"""
import collections
import csv
import glob
import json
import re
import sys

d = sys.argv[1]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 5
only = sys.argv[3:]


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\[clone[^\]]*\]", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0].strip()[:110]


seen = collections.defaultdict(list)     # kernel -> dispatch ids in order
vals = collections.defaultdict(lambda: collections.defaultdict(float))
meta = {}
for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = short(row["Kernel_Name"])
        if only and not any(o in k for o in only):
            continue
        did = row["Dispatch_Id"]
        if did not in seen[k]:
            seen[k].append(did)
        vals[(k, did)][row["Counter_Name"]] += float(row["Counter_Value"])
        meta[k] = {x: row.get(x) for x in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Workgroup_Size", "Grid_Size") if x in row}
out = {}
for k, dids in seen.items():
    keep = dids[skip:] if len(dids) > skip else dids
    acc = collections.defaultdict(float)
    for did in keep:
        for c, v in vals[(k, did)].items():
            acc[c] += v
    out[k] = {"dispatches_averaged": len(keep), **meta.get(k, {}), **{c: v / len(keep) for c, v in sorted(acc.items())}}
print(json.dumps(out, indent=1))
