"""Development aid: host-side cost of FrameSharder.trace_group (launches + assemble, no collective at world size 1) per
group of 8, 16 and 32 frames, beside the GPU time."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
import ray_tracer_amd as rt
from ray_tracer_amd import sharding
W, H = 1920, 1080
core = rt.TraceCore(0)
core.set_scene(None, rt.triangles_from_xyz9(np.load("/root/repo/tests/golden/car_tris.npy")))
core.build_grid()
cam = rt.Camera.for_scene("gltf").to_uniform(0.0, float(H))
for g in (8, 16, 32):
    sh = sharding.FrameSharder(core, W, H, 32, 32, rank=0, world=1, device=torch.device("cuda", 0), shadow=True, mode="rgba", max_bounces=1)
    sh.enable_groups(g, 2, rgb8=True)
    cams = [cam] * g
    for _ in range(5): sh.trace_group(cams)
    sh.synchronize()
    n = 40
    t0 = time.perf_counter()
    for _ in range(n): sh.trace_group(cams)
    host = (time.perf_counter() - t0) / n * 1e6
    sh.synchronize()
    tot = (time.perf_counter() - t0) / n * 1e6
    print(f"group {g}: host side {host:.0f} us per group ({host / g:.1f} per frame), with GPU {tot:.0f} us per group ({tot / g:.1f} per frame)")
