#!/usr/bin/env python3
"""Development tool: GPU time of the shaded picture (rt_render) for the bundled static model with its own materials and
textures at 1920x1080: the shader's 8 bounces, direct lighting only, with and without the shadow term; frames
alternate between two streams as in bench.py."""
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
core = rt.TraceCore(0)
tris, mats, textures = bundled_model(rt, "car", np.load(os.path.join(ROOT, "tests/golden/car_tris.npy")))
cam = rt.Camera.for_scene("gltf").to_uniform(0.0, float(H))
streams = [torch.cuda.Stream() for _ in range(2)]
outs = [torch.empty((W * H, 4), dtype=torch.uint8, device="cuda") for _ in range(2)]
for label, m, t in (("grey (no materials)", None, None), ("materials", mats, None), ("materials + textures", mats, textures)):
    core.set_scene(None, tris)
    core.set_materials(m)
    core.set_textures(t)
    core.build_grid()
    for bounces, shadows in ((8, False), (8, True), (1, True)):
        def frames(n):
            for k in range(n):
                core.set_stream(streams[k % 2].cuda_stream)
                core.render(cam, W, H, max_bounces=bounces, shadows=shadows, out=outs[k % 2])
            core.sync()
        frames(10)
        t0 = time.perf_counter()
        frames(200)
        print(f"{label:22s} bounces {bounces} shadows {int(shadows)}: {(time.perf_counter() - t0) / 200 * 1e6:7.1f} us per frame", flush=True)
