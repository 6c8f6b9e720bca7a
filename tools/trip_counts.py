#!/usr/bin/env python3
"""Development tool: per-wave loop trip counts of the instrumented variant on the headline frame."""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RT_LIB_PATH", os.path.join(ROOT, "ray-tracer_amd", "librt_mi355_exp.so"))
import numpy as np
import torch
import ray_tracer_amd as rt
from ray_tracer_amd import _lib
W, H = 1920, 1080
core = rt.TraceCore(0)
core.set_scene(None, rt.triangles_from_xyz9(np.load(os.path.join(ROOT, "tests/golden/car_tris.npy"))))
core.build_grid()
cam = rt.Camera.for_scene("gltf").to_uniform(0.0, float(H))
p = torch.empty((W * H, 2), dtype=torch.int32, device="cuda"); s = torch.empty_like(p)
L = _lib.load()
out = (C.c_ulonglong * 8)()
L.rt_dbg_trips(core._h, out)
os.environ["RT_VARIANT"] = sys.argv[1] if len(sys.argv) > 1 else "2"
core.trace_primary(cam, W, H, out_primary=p, out_shadow=s)
L.rt_dbg_trips(core._h, out)
names = ["primary waves", "primary waves in grid", "primary DDA trips", "primary MT trips", "shadow waves", "shadow waves in grid", "shadow DDA trips", "shadow MT trips"]
nw = (W // 8) * (H // 8)
for n, v in zip(names, out):
    print(f"{n:24s} {v:10d}  per wave {v / nw:7.2f}")
