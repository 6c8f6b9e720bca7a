#!/usr/bin/env python3
"""Development tool: the headline figure and the config legs of a saved bench.py line, one per row.
Usage: python3 tools/show_configs.py <file holding bench.py's JSON line>"""
import sys,json
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["configs_wall_s"])
for c in d["configs"]:
    print(c["config"][:100], "| ms", c.get("ms"), "| one", c.get("ms_one_frame_at_a_time"), "| inflight", c.get("frames_in_flight"), "| Grays", c.get("Grays_per_s"), "| frac", c.get("frac"))
