#!/usr/bin/env python3
"""Turn the rocprofv3 outputs of tools/collect_profiles.sh into the committed summaries under profiles/.

    python tools/summarise_profiles.py r02 c2_q10k
"""
import collections
import csv
import re
import glob
import json
import os
import shutil
import sys

tag, workload = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_dir = os.path.join(root, "profiles")
os.makedirs(out_dir, exist_ok=True)


def short(name):
    m = re.search(r"msda_\w+?_kernel", name)  # demangled ("void msda::x<...>") and mangled ("_ZN4msda..x...") names alike
    return m.group(0) if m else name


GROUPS = {  # launcher-level groups timed by bench.py's KernelTimer
    "msda_fwd": ["msda_fwd_kernel", "msda_fwd_unit_kernel"],
    "msda_bwd_sample": ["msda_bwd_sample_kernel"],
    "msda_bwd_value": ["msda_cell_pass_kernel", "msda_cell_scan_kernel", "msda_cell_place_lm_kernel", "msda_value_gather_kernel",
                       "msda_value_finish_kernel", "msda_value_small_kernel", "msda_cell_place_det_kernel"],
}

def newest(pattern):  # gpurun merges new outputs next to older ones: take the most recent
    return max(glob.glob(pattern), key=os.path.getmtime)


stats = newest(os.path.join(root, "gpurun_out", f"{tag}_{workload}_stats", "*", "*kernel_stats.csv"))
shutil.copy(stats, os.path.join(out_dir, f"{tag}_{workload}_kernel_stats.csv"))
per_kernel = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(stats)):
    if "msda" in r["Name"]:
        k = short(r["Name"])
        per_kernel[k][0] += int(r["Calls"])
        per_kernel[k][1] += float(r["TotalDurationNs"])

pmc = {}
rows = []
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    f = newest(os.path.join(root, "gpurun_out", f"{tag}_{workload}_{counter.split('_')[0].lower()}", "*", "*counter_collection.csv"))
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter and "msda" in r["Kernel_Name"]:
            acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        pmc.setdefault(k, {})[counter] = (sum(v) / len(v), len(v))
        rows.append((counter, k, len(v), sum(v) / len(v), min(v), max(v)))
with open(os.path.join(out_dir, f"{tag}_{workload}_pmc_hbm.csv"), "w") as f:
    f.write("counter,kernel,dispatches,mean_KiB_per_dispatch,min_KiB,max_KiB\n")
    for r in rows:
        f.write("%s,%s,%d,%.3f,%.3f,%.3f\n" % r)

steps = per_kernel["msda_bwd_sample_kernel"][0] or 1  # one backward per step
summary, traffic = {}, {}
for group, kernels in GROUPS.items():
    tot_ns = sum(per_kernel[k][1] for k in kernels if k in per_kernel)
    launches = per_kernel["msda_fwd_kernel"][0] if group == "msda_fwd" else steps
    # HBM bytes per step of the group: every dispatch of its kernels, (2*FETCH + WRITE) KiB (guide: gfx950
    # FETCH_SIZE reports half of wide coalesced reads; WRITE_SIZE is exact)
    kib = 0.0
    for k in kernels:
        if k in pmc and k in per_kernel:
            per_step = per_kernel[k][0] / (launches if group == "msda_fwd" else steps)
            kib += per_step * (2 * pmc[k].get("FETCH_SIZE", (0, 0))[0] + pmc[k].get("WRITE_SIZE", (0, 0))[0])
    summary[group] = {"avg_us_per_call": round(tot_ns / launches / 1e3, 2),
                      "kernels": {k: {"calls": per_kernel[k][0], "avg_us": round(per_kernel[k][1] / per_kernel[k][0] / 1e3, 2)}
                                  for k in kernels if k in per_kernel}}
    traffic[group] = int(kib * 1024)
json.dump(summary, open(os.path.join(out_dir, f"{tag}_{workload}_summary.json"), "w"), indent=1)
tpath = os.path.join(out_dir, "hbm_traffic.json")
allt = json.load(open(tpath)) if os.path.exists(tpath) else {}
allt["_comment"] = ("HBM bytes per call of each launcher-level group, from rocprofv3 PMC passes (separate --pmc FETCH_SIZE / "
                    "--pmc WRITE_SIZE runs of bench.py, tools/collect_profiles.sh): bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 "
                    "summed over the group's kernels, per guides/MI355X_MICROARCH.md (gfx950 FETCH_SIZE reports half of wide "
                    "coalesced reads; WRITE_SIZE is exact). Raw per-kernel means: <tag>_<workload>_pmc_hbm.csv.  CAVEAT: the "
                    "guide calibrates the 2x FETCH_SIZE correction for 16-byte-per-lane streaming reads only; it is applied "
                    "to every kernel here, so for the scatter / gather-heavy kernels (place pass, value_gather, the row "
                    "gathers of msda_fwd / msda_bwd_sample) the figures are upper-ish bounds, and Infinity-Cache hits are "
                    "counted as traffic.  Ratios between rounds of the same kernel are unaffected.")
allt[workload] = traffic
json.dump(allt, open(tpath, "w"), indent=1)
print(json.dumps(summary, indent=1))
print(traffic)
