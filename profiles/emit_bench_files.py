#!/usr/bin/env python3
"""Writes the small counter files bench.py reads from a committed summary of profiles/summarize_r02.py:
    python profiles/emit_bench_files.py headline  profiles/r03_headline_summary.json
    python profiles/emit_bench_files.py streaming profiles/r03_streaming_summary.json
The files are named after the summary's round (its first four characters: r02_, r03_); bench.py reads the newest."""
import json
import os
import re
import sys

leg, path = sys.argv[1], sys.argv[2]
d = json.load(open(path))
derived, counters, bench = d["derived"], d["counters"], d["bench_line"]
m = lambda k: counters[k]["mean_per_dispatch"]
here = os.path.dirname(os.path.abspath(__file__))
rnd = os.path.basename(path)[:4]  # "r03_"
source = (f"profiles/{os.path.basename(path)} (rocprofv3 --pmc passes of profiles/collect_" + rnd[:3] + ".sh, mean over the timed kernel's "
          "dispatches; gfx950 FETCH_SIZE correction applied)")
if leg == "headline":
    json.dump({"source": source, "bytes_per_launch": derived["hbm_bytes_per_launch"],
               "read_bytes_per_launch": derived["hbm_read_bytes_per_launch"],
               "write_bytes_per_launch": derived["hbm_write_bytes_per_launch"], "l2_hit_rate": derived["l2_hit_rate"]},
              open(os.path.join(here, rnd + "headline_traffic.json"), "w"), indent=1)
    json.dump({"source": source, "SQ_INSTS_VALU_per_launch": m("SQ_INSTS_VALU"), "SQ_INSTS_SALU_per_launch": m("SQ_INSTS_SALU"),
               "SQ_INSTS_BRANCH_per_launch": m("SQ_INSTS_BRANCH"), "SQ_INSTS_VMEM_RD_per_launch": m("SQ_INSTS_VMEM_RD"),
               "SQ_WAVES_per_launch": m("SQ_WAVES"),
               "lanes_active_per_valu_instruction": derived["lanes_active_per_valu_instruction"]},
              open(os.path.join(here, rnd + "headline_valu.json"), "w"), indent=1)
else:
    rs = bench["roofline_streaming"]
    rec = {"source": source, "triangles": int(re.match(r"(\d+) distinct", rs["workload"]).group(1)),
           "cell_size": float(re.search(r"grid cell ([0-9.]+)", rs["workload"]).group(1)),
           "repack_candidates": int(rs.get("repack_candidates", 0)), "kernel": d["kernel"],
           "bytes_per_launch": derived["hbm_bytes_per_launch"], "read_bytes_per_launch": derived["hbm_read_bytes_per_launch"],
           "write_bytes_per_launch": derived["hbm_write_bytes_per_launch"]}
    for k in ("l2_hit_rate", "lanes_active_per_valu_instruction", "ta_busy_frac", "valu_issue_frac"):
        if k in derived:
            rec[k] = derived[k]
    json.dump(rec, open(os.path.join(here, rnd + "streaming_traffic.json"), "w"), indent=1)
