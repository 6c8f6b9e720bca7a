#!/usr/bin/env python3
"""Development tool: where the time of a SHORT timed region goes (bench.py --steps 20 is 1.3 ms of GPU work). The
headline frame, three streams, K frames between two synchronisations; an event before and after every frame on its
stream gives each frame's start and end on the GPU's clock, the host clock brackets the whole."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import ray_tracer_amd as rt  # noqa: E402

W, H = 1920, 1080
K = int(os.environ.get("RT_K", "20"))
NS = int(os.environ.get("RT_STREAMS", "3"))
core = rt.TraceCore(0, rt.default_params(cull_empty_blocks=1))
core.set_scene(None, rt.triangles_from_xyz9(np.load(os.path.join(ROOT, "tests/golden/car_tris.npy"))))
core.build_grid()
cam = rt.Camera.for_scene("gltf").to_uniform(0.0, float(H))
streams = [torch.cuda.Stream() for _ in range(NS)]
bufs = [(torch.empty((W * H, 2), dtype=torch.int32, device="cuda"), torch.empty((W * H, 2), dtype=torch.int32, device="cuda")) for _ in range(NS)]


def frame(k):
    core.set_stream(streams[k % NS].cuda_stream)
    core.trace_primary(cam, W, H, shadow=True, out_primary=bufs[k % NS][0], out_shadow=bufs[k % NS][1])


t_end = time.perf_counter() + 0.3
while time.perf_counter() < t_end:
    for k in range(9):
        frame(k)
    core.sync()
for rep in range(4):
    for k in range(5):
        frame(k)
    core.sync()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
    t0 = time.perf_counter()
    for k in range(K):
        ev[k][0].record(streams[k % NS])
        frame(k)
        ev[k][1].record(streams[k % NS])
    t_sub = time.perf_counter()
    if os.environ.get("RT_SPIN") == "1":  # poll the streams instead of sleeping in the driver
        while not all(st.query() for st in streams):
            pass
    t_spin = time.perf_counter()
    core.sync()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    starts = [ev[0][0].elapsed_time(ev[k][0]) * 1e3 for k in range(K)]
    ends = [ev[0][0].elapsed_time(ev[k][1]) * 1e3 for k in range(K)]
    print(f"rep {rep}: host {1e6 * (t1 - t0):7.1f} us ({1e6 * (t1 - t0) / K:5.1f} per frame), submitted after {1e6 * (t_sub - t0):6.1f}, streams idle after {1e6 * (t_spin - t0):7.1f}; "
          f"GPU first start -> last end {max(ends):7.1f} us")
    if rep == 3:
        for k in range(K):
            print(f"  frame {k:2d}: start {starts[k]:7.1f}  end {ends[k]:7.1f}  ({ends[k] - starts[k]:6.1f})")
