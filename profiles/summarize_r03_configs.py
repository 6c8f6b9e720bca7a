#!/usr/bin/env python3
"""Condenses the runs of profiles/collect_r03_configs.sh into profiles/r03_config_counters.json, the file
tools/config_legs.py reads: per leg the timed kernel's vector instructions, lanes active, HBM bytes (gfx950 FETCH_SIZE
correction applied) and its kernel-trace average, all per launch.
usage: python profiles/summarize_r03_configs.py gpurun_out > profiles/r03_config_counters.json"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

base = sys.argv[1]
needle = "k_trace_primary_lean<false, true>"
out = {}
for root in sorted(glob.glob(os.path.join(base, "r03_cfg_*"))):
    leg = os.path.basename(root)[len("r03_cfg_"):]
    counters = {}
    for path in sorted(glob.glob(f"{root}/*/**/*counter_collection.csv", recursive=True)):
        acc = defaultdict(lambda: defaultdict(float))
        for row in csv.DictReader(open(path)):
            if needle in row["Kernel_Name"]:
                acc[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
        for name, per in acc.items():
            vals = list(per.values())[1:] or list(per.values())  # the first launch is the warm-up one
            counters[name] = sum(vals) / len(vals)
    avg_ns = calls = None
    for path in glob.glob(f"{root}/trace/**/*kernel_stats.csv", recursive=True):
        for row in csv.DictReader(open(path)):
            if needle in row["Name"]:
                avg_ns, calls = float(row["AverageNs"]), int(row["Calls"])
    if "SQ_INSTS_VALU" not in counters:
        continue
    rec = {"source": f"rocprofv3 --pmc passes of profiles/collect_r03_configs.sh {leg} (mean over the timed kernel's dispatches; "
                     "gfx950 FETCH_SIZE correction applied)",
           "kernel": needle, "kernel_trace_average_ns": avg_ns, "kernel_trace_calls": calls,
           "SQ_INSTS_VALU_per_launch": counters["SQ_INSTS_VALU"], "SQ_INSTS_SALU_per_launch": counters.get("SQ_INSTS_SALU"),
           "SQ_WAVES_per_launch": counters.get("SQ_WAVES")}
    if counters.get("SQ_THREAD_CYCLES_VALU") and counters.get("SQ_ACTIVE_INST_VALU"):
        rec["lanes_active_per_valu_instruction"] = counters["SQ_THREAD_CYCLES_VALU"] / counters["SQ_ACTIVE_INST_VALU"]
    if counters.get("SQ_WAVE_CYCLES") and counters.get("SQ_WAIT_ANY"):
        rec["memory_wait_share_of_wave_cycles"] = counters["SQ_WAIT_ANY"] / counters["SQ_WAVE_CYCLES"]
    if "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
        rec["hbm_read_bytes_per_launch"] = 2 * counters["FETCH_SIZE"] * 1024
        rec["hbm_write_bytes_per_launch"] = counters["WRITE_SIZE"] * 1024
        rec["hbm_bytes_per_launch"] = rec["hbm_read_bytes_per_launch"] + rec["hbm_write_bytes_per_launch"]
    out[leg] = rec
json.dump(out, sys.stdout, indent=1)
print()
