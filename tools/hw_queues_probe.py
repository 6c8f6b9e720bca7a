#!/usr/bin/env python3
"""Development tool: what a process's earlier streams do to contexts that are meant to overlap. The animated model's frame
(skin + rebuild + trace) with two contexts used in turn is timed before and after `what`:
  streams        three torch streams used once each              frames1     300 headline frames on ONE torch stream
  bigalloc       4 GiB allocated and released                    frames_own  ... on the context's own stream
  frames         300 headline frames on three torch streams      frames_few  six frames on three torch streams
Run with GPU_MAX_HW_QUEUES=4 (HIP's default), 5, 8 ... in the environment: with four queues the two contexts share a
hardware queue after `frames` / `frames_few` (0.19 -> 0.35 ms per frame), with five or more they do not."""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
import ray_tracer_amd as rt
import config_legs
dev = torch.device("cuda", 0)
what = sys.argv[1]
os.environ["RT_ANIM_ONLY"] = "2,0,0"
def anim():
    l = config_legs.animated(rt, dev)[0]
    return round(l["frame_ms_wall"], 4)
print(what, "before:", anim())
W, H = 1920, 1080
if what in ("frames", "frames_events", "frames1", "frames_own", "frames_nomask", "frames_few"):
    core = rt.TraceCore(0, rt.default_params(cull_empty_blocks=0 if what == "frames_nomask" else 1))
    core.set_scene(None, rt.triangles_from_xyz9(np.load(os.path.join(ROOT, "tests/golden/car_tris.npy"))))
    core.build_grid()
    cam = rt.Camera.for_scene("gltf").to_uniform(0.0, float(H))
    streams = [torch.cuda.Stream() for _ in range(3)]
    bufs = [(torch.empty((W * H, 2), dtype=torch.int32, device="cuda"), torch.empty((W * H, 2), dtype=torch.int32, device="cuda")) for _ in range(3)]
    ns = 1 if what == "frames1" else 3
    for k in range(6 if what == "frames_few" else 300):
        if what != "frames_own":
            core.set_stream(streams[k % ns].cuda_stream)
        core.trace_primary(cam, W, H, shadow=True, out_primary=bufs[k % 3][0], out_shadow=bufs[k % 3][1])
    core.sync()
    if what == "frames_events":
        for _ in range(30):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(streams[0]); core.trace_primary(cam, W, H, shadow=True, out_primary=bufs[0][0], out_shadow=bufs[0][1]); e1.record(streams[0]); core.sync(); e0.elapsed_time(e1)
    core.close()
elif what == "bigalloc":
    a = torch.zeros(1 << 29, dtype=torch.float32, device=dev); b = torch.empty_like(a); torch.cuda.synchronize(); del a, b
elif what == "streams":
    s = [torch.cuda.Stream() for _ in range(3)]
    x = torch.zeros(10, device=dev)
    for st in s:
        with torch.cuda.stream(st): x += 1
    torch.cuda.synchronize()
print(what, "after:", anim())
