#!/usr/bin/env python3
"""Development tool: host-side cost of one rt_trace_primary call at several frame sizes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import ray_tracer_amd as rt
core = rt.TraceCore(0)
core.set_scene(None, rt.triangles_from_xyz9(np.load(os.path.join(ROOT, "tests/golden/car_tris.npy"))))
core.build_grid()
p = torch.empty((1920 * 1080, 2), dtype=torch.int32, device="cuda"); s = torch.empty_like(p)
for (W, H) in ((64, 36), (480, 270), (1920, 1080)):
    cam = rt.Camera.for_scene("gltf").to_uniform(0.0, float(H))
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(300):
            core.trace_primary(cam, W, H, out_primary=p, out_shadow=s)
        host = (time.perf_counter() - t0) / 300 * 1e6
        torch.cuda.synchronize()
        tot = (time.perf_counter() - t0) / 300 * 1e6
    print(f"{W}x{H}: host {host:.1f} us per call, end to end {tot:.1f} us per frame")
