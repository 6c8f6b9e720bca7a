#!/usr/bin/env python3
"""Development tool: throughput of the headline frame with 1, 2, 3 frames in flight (one stream each, own output
buffers), against back-to-back launches on one stream."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import ray_tracer_amd as rt

W, H = 1920, 1080
tris = rt.triangles_from_xyz9(np.load(os.path.join(ROOT, "tests/golden/car_tris.npy")))
cores = []
for _ in range(4):  # one context per frame in flight: each has its own stream and scratch buffers
    c = rt.TraceCore(0)
    c.set_scene(None, tris)
    c.build_grid()
    cores.append(c)
cam = rt.Camera.for_scene("gltf").to_uniform(0.0, float(H))
for depth in (1, 2, 3, 4):
    bufs = [(torch.empty((W * H, 2), dtype=torch.int32, device="cuda"), torch.empty((W * H, 2), dtype=torch.int32, device="cuda"))
            for _ in range(depth)]
    steps = 400
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            cores[k % depth].trace_primary(cam, W, H, out_primary=bufs[k % depth][0], out_shadow=bufs[k % depth][1])
        host = (time.perf_counter() - t0) / steps * 1e6
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps * 1e6
    ok = all(torch.equal(b[0], bufs[0][0]) and torch.equal(b[1], bufs[0][1]) for b in bufs)
    print(f"{depth} frame(s) in flight: {dt:.1f} us/frame (host side of a launch: {host:.1f} us)  identical={ok}")
