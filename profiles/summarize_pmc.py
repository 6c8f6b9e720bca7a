#!/usr/bin/env python3
"""Per-dispatch means of rocprofv3 --pmc passes for one kernel.
usage: python profiles/summarize_pmc.py gpurun_out/pmc_<tag> [kernel-substring] > profiles/<name>.json"""
import csv
import glob
import json
import sys
from collections import defaultdict

root = sys.argv[1]
needle = sys.argv[2] if len(sys.argv) > 2 else "k_trace_primary<false, true, false>"
out = {}
for path in sorted(glob.glob(f"{root}/*/**/*counter_collection.csv", recursive=True)):
    acc = defaultdict(lambda: defaultdict(float))  # counter -> dispatch -> value
    for row in csv.DictReader(open(path)):
        if needle not in row["Kernel_Name"]:
            continue
        acc[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
    for name, per in acc.items():
        vals = list(per.values())
        out[name] = {"mean_per_dispatch": sum(vals) / len(vals), "dispatches": len(vals)}
json.dump({"kernel": needle, "counters": out}, sys.stdout, indent=1)
print()
