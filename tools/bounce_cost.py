#!/usr/bin/env python3
"""Development tool: what each reflection bounce of the shaded picture costs (bundled static model with its materials
and textures, 1920x1080, two frames in flight) and how many rays it traces."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import ray_tracer_amd as rt  # noqa: E402
from render_scenes import bundled_model  # noqa: E402

W, H = 1920, 1080
PLAIN = os.environ.get("RT_PLAIN_BOUNCES") == "1"  # rt_render_opts.plain_bounces: every ray scans its own lists
core = rt.TraceCore(0)
tris, mats, textures = bundled_model(rt, "car", np.load(os.path.join(ROOT, "tests/golden/car_tris.npy")))
cam = rt.Camera.for_scene("gltf").to_uniform(0.0, float(H))
NF = int(os.environ.get("RT_INFLIGHT", "2"))  # frames in flight
streams = [rt.shared_stream("cuda:0", k) for k in range(NF)]
outs = [torch.empty((W * H, 4), dtype=torch.uint8, device="cuda") for _ in range(NF)]
if os.environ.get("RT_ALPHA_OPAQUE") == "1":  # no hit is decided by a texture's alpha: the picture kernel that shares bounce rays' lists may run
    mats = mats.copy()
    mats["alpha_mode"] = 0
core.set_scene(None, tris)
core.set_materials(mats)
core.set_textures(textures)
core.build_grid()
for bounces in (1, 2, 3, 4, 8):
    core.set_collect_stats(1)
    core.set_stream(None)
    core.render(cam, W, H, max_bounces=bounces, shadows=False, out=outs[0])
    rays = core.stats()["rays"]
    core.set_collect_stats(0)

    def frames(n):
        for k in range(n):
            core.set_stream(streams[k % NF].cuda_stream)
            core.render(cam, W, H, max_bounces=bounces, shadows=False, out=outs[k % NF], plain_bounces=PLAIN)
        core.sync()
    frames(10)
    t0 = time.perf_counter()
    frames(200)
    print(f"bounces {bounces}: {rays} rays ({rays - W * H} beyond the primary), {(time.perf_counter() - t0) / 200 * 1e6:7.1f} us per frame", flush=True)
