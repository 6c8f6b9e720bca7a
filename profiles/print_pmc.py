import json, sys
d = json.load(open(sys.argv[1]))["counters"]
c = {k: v["mean_per_dispatch"] for k, v in d.items()}
w = c["SQ_WAVES"]
print(f"waves {w:.0f}  VALU/wave {c['SQ_INSTS_VALU']/w:.0f}  SALU/wave {c['SQ_INSTS_SALU']/w:.0f}  VMEM_RD/wave {c['SQ_INSTS_VMEM_RD']/w:.1f}  branch/wave {c['SQ_INSTS_BRANCH']/w:.0f}")
print(f"lanes/VALU {c['SQ_THREAD_CYCLES_VALU']/c['SQ_ACTIVE_INST_VALU']:.1f}  wave_cycles {c['SQ_WAVE_CYCLES']:.3g}  wait_any {c['SQ_WAIT_ANY']/c['SQ_WAVE_CYCLES']:.2f}  wait_inst {c['SQ_WAIT_INST_ANY']/c['SQ_WAVE_CYCLES']:.2f}  active {c['SQ_ACTIVE_INST_ANY']/c['SQ_WAVE_CYCLES']:.2f}  busy_cycles {c['SQ_BUSY_CYCLES']:.3g}")
