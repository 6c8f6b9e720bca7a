#!/usr/bin/env python3
"""rt_params.persistent_waves on and off, the shipped kernels: one frame at a time and three frames in flight, on the
bundled static model's headline frame and on the animated model's bind pose (whose frame is a few long waves).
One process per configuration (python tools/queue_probe.py <car|phoenix> <0|1> <frames in flight>), or all eight in
turn without arguments. Records of the two modes are compared by tests/test_gpu_queue.py."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) < 4:
    for model in ("car", "phoenix"):
        for flight in (1, 3):
            for q in (0, 1):
                subprocess.check_call([sys.executable, os.path.abspath(__file__), model, str(q), str(flight)])
    raise SystemExit(0)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import ray_tracer_amd as rt  # noqa: E402

model, q, flight = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
W, H = 1920, 1080
pose = rt.Camera.for_scene("gltf") if model == "car" else rt.Camera((-300, 250, 600), 2.7, -0.2)
cam = pose.to_uniform(0.0, float(H))
core = rt.TraceCore(0, rt.default_params(persistent_waves=q))
core.set_scene(None, rt.triangles_from_xyz9(np.load(os.path.join(ROOT, "tests", "golden", f"{model}_tris.npy"))))
core.build_grid()
streams = [torch.cuda.Stream() for _ in range(flight)]
bufs = [(torch.empty((W * H, 2), dtype=torch.int32, device="cuda"), torch.empty((W * H, 2), dtype=torch.int32, device="cuda")) for _ in range(flight)]


def frames(n):
    for k in range(n):
        core.set_stream(streams[k % flight].cuda_stream)
        core.trace_primary(cam, W, H, out_primary=bufs[k % flight][0], out_shadow=bufs[k % flight][1])


frames(600)  # clocks up
core.sync()
times = []
for rnd in range(5):
    t0 = time.perf_counter()
    frames(600)
    core.sync()
    times.append((time.perf_counter() - t0) / 600 * 1e6)
print(f"{model}, {flight} frame(s) in flight, persistent_waves={q}: median {np.median(times):7.1f} us per frame, min {min(times):7.1f}", flush=True)
core.close()
