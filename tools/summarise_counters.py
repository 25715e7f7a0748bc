#!/usr/bin/env python3
"""Per-kernel means of the PMC passes of tools/collect_counters.sh -> profiles/<tag>_<workload>_pmc_pipes.csv

    python tools/summarise_counters.py r01 c2_q10k
"""
import collections
import csv
import glob
import os
import sys

tag, workload = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    import re
    m = re.search(r"msda_\w+?_kernel", name)
    return (m.group(0) + name.split(m.group(0), 1)[1].split("(")[0]) if m else name


vals = collections.defaultdict(lambda: collections.defaultdict(list))  # kernel -> counter -> [per dispatch]
for d in sorted(glob.glob(os.path.join(root, "gpurun_out", f"{tag}_pmc_*"))):
    files = glob.glob(os.path.join(d, "*", "*counter_collection.csv"))
    if not files:
        continue
    f = max(files, key=os.path.getmtime)
    per_dispatch = collections.defaultdict(float)
    meta = {}
    for r in csv.DictReader(open(f)):
        if "msda" not in r["Kernel_Name"]:
            continue
        key = (r["Dispatch_Id"], r["Counter_Name"])
        per_dispatch[key] += float(r["Counter_Value"])  # rows are per dimension instance (XCD/SE...): sum them
        meta[r["Dispatch_Id"]] = short(r["Kernel_Name"])
    for (disp, ctr), v in per_dispatch.items():
        vals[meta[disp]][ctr].append(v)

counters = sorted({c for k in vals.values() for c in k})
out = os.path.join(root, "profiles", f"{tag}_{workload}_pmc_pipes.csv")
with open(out, "w") as f:
    f.write("kernel,dispatches," + ",".join(counters) + "\n")
    for k in sorted(vals):
        n = max(len(v) for v in vals[k].values())
        f.write(k.replace(",", ";") + f",{n}," + ",".join(
            ("%.6g" % (sum(vals[k][c]) / len(vals[k][c]))) if vals[k].get(c) else "" for c in counters) + "\n")
print(open(out).read())
