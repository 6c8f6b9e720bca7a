#!/usr/bin/env python3
"""Summary of one profiles/collect_r02.sh run: per-dispatch means of every counter for the timed trace kernel (the
non-counting k_trace_primary instantiation: dispatches of the counting launches are told apart by their names), the
kernel-trace statistics line of the same kernel, and the derived figures bench.py reads.
usage: python profiles/summarize_r02.py gpurun_out/r02_<tag> > profiles/r02_<tag>_summary.json"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
needle = sys.argv[2] if len(sys.argv) > 2 else "k_trace_primary_lean<false, true>"
counters = {}
for path in sorted(glob.glob(f"{root}/*/**/*counter_collection.csv", recursive=True)):
    acc = defaultdict(lambda: defaultdict(float))  # counter -> dispatch -> value
    for row in csv.DictReader(open(path)):
        if needle not in row["Kernel_Name"]:
            continue
        acc[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
    for name, per in acc.items():
        vals = list(per.values())
        counters[name] = {"mean_per_dispatch": sum(vals) / len(vals), "dispatches": len(vals)}
stats = None
for path in glob.glob(f"{root}/trace/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        if needle in row["Name"]:
            stats = {k: row[k] for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage") if k in row}
bench = None
bpath = os.path.join(root, "bench.json")
if os.path.exists(bpath):
    txt = open(bpath).read().strip()
    bench = json.loads(txt.splitlines()[-1]) if txt else None
m = lambda k: counters[k]["mean_per_dispatch"] if k in counters else None
derived = {}
if m("FETCH_SIZE") is not None and m("WRITE_SIZE") is not None:
    # gfx950: FETCH_SIZE (KB) counts 128-byte read requests as 64 bytes -> read bytes = 2 * FETCH_SIZE * 1024;
    # WRITE_SIZE (KB) is exact for wide stores
    derived["hbm_read_bytes_per_launch"] = 2 * m("FETCH_SIZE") * 1024
    derived["hbm_write_bytes_per_launch"] = m("WRITE_SIZE") * 1024
    derived["hbm_bytes_per_launch"] = derived["hbm_read_bytes_per_launch"] + derived["hbm_write_bytes_per_launch"]
if m("TCC_HIT_sum") is not None and m("TCC_MISS_sum") is not None:
    derived["l2_hit_rate"] = m("TCC_HIT_sum") / (m("TCC_HIT_sum") + m("TCC_MISS_sum"))
if m("SQ_THREAD_CYCLES_VALU") and m("SQ_ACTIVE_INST_VALU"):
    derived["lanes_active_per_valu_instruction"] = m("SQ_THREAD_CYCLES_VALU") / m("SQ_ACTIVE_INST_VALU")
if m("SQ_INSTS_VALU") and m("SQ_WAVES"):
    derived["valu_per_wave"] = m("SQ_INSTS_VALU") / m("SQ_WAVES")
    derived["salu_per_wave"] = m("SQ_INSTS_SALU") / m("SQ_WAVES")
if m("SQ_WAVE_CYCLES"):
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
        if m(k):
            derived[k + "_share_of_wave_cycles"] = m(k) / m("SQ_WAVE_CYCLES")
if m("TA_BUSY_avr") and m("GRBM_GUI_ACTIVE"):
    # GRBM_GUI_ACTIVE sums the 8 XCDs; TA_BUSY_avr is the mean over the texture-address units (one per CU)
    derived["ta_busy_frac"] = m("TA_BUSY_avr") / (m("GRBM_GUI_ACTIVE") / 8)
if m("SQ_INSTS_VALU") and m("GRBM_GUI_ACTIVE"):
    # 1 024 SIMDs, one vector instruction per 2 cycles each
    derived["valu_issue_frac"] = m("SQ_INSTS_VALU") * 2 / 1024 / (m("GRBM_GUI_ACTIVE") / 8)
json.dump({"kernel": needle, "kernel_trace_stats": stats, "derived": derived, "counters": counters, "bench_line": bench},
          sys.stdout, indent=1)
print()
