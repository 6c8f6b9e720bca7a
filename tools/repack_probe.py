#!/usr/bin/env python3
"""Development tool: rt_params.repack_candidates on and off, the shipped kernels, launches back to back on one stream.
    python tools/repack_probe.py [stream[:n[:cell factor]] | car | phoenix]"""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import ray_tracer_amd as rt  # noqa: E402

spec = (sys.argv[1] if len(sys.argv) > 1 else "stream").split(":")
cell = 0.0
if spec[0] == "stream":
    n = int(spec[1]) if len(spec) > 1 else 10000000
    w, h, steps = 3840, 2160, 10
    rng = np.random.default_rng(1)
    size = 100.0 / n ** (1.0 / 3.0)
    c = rng.uniform(-50, 50, (n, 1, 3)).astype(np.float32)
    tris = rt.triangles_from_xyz9((c + rng.normal(0, 0.6 * size, (n, 3, 3)).astype(np.float32)).astype(np.float32).reshape(-1, 9))
    del c
    cell = float(spec[2] if len(spec) > 2 else 1.0) * size
    cam = rt.Camera((0, 0, 75), math.pi, 0.0).to_uniform(0.0, float(h))
else:
    w, h, steps = 1920, 1080, 100
    tris = rt.triangles_from_xyz9(np.load(os.path.join(ROOT, f"tests/golden/{spec[0]}_tris.npy")))
    cam = (rt.Camera.for_scene("gltf") if spec[0] == "car" else rt.Camera((-300, 250, 600), 2.7, -0.2)).to_uniform(0.0, float(h))
RAYS = os.environ.get("RT_RAYS") == "1"  # rt_trace_rays on 4 M rays with random origins and directions instead of a frame
if RAYS:
    g = torch.Generator(device="cuda").manual_seed(3)
    n_rays = 1 << 22
    ray_o = (torch.rand((n_rays, 3), device="cuda", generator=g) * 100.0 - 50.0).contiguous()
    ray_d = torch.nn.functional.normalize(torch.randn((n_rays, 3), device="cuda", generator=g), dim=1).contiguous()
    w, h = n_rays, 1
outs = {}
times = {0: [], 1: []}
cores = {}
for repack in (0, 1):
    core = rt.TraceCore(0, rt.default_params(repack_candidates=repack, cull_empty_blocks=0))
    core.set_scene(None, tris)
    core.build_grid(cell_size=cell)
    cores[repack] = core
    outs[repack] = (torch.empty((w * h, 2), dtype=torch.int32, device="cuda"), torch.empty((w * h, 2), dtype=torch.int32, device="cuda"))
for rnd in range(3):
    for repack in (0, 1):
        core = cores[repack]
        p, s = outs[repack]
        def frame():
            if RAYS:
                core.trace_rays(ray_o, ray_d, out=p, n=n_rays)
            else:
                core.trace_primary(cam, w, h, out_primary=p, out_shadow=s)
        for _ in range(2):
            frame()
        core.sync()
        core.timer_begin()
        for _ in range(steps):
            frame()
        times[repack].append(core.timer_end() / steps * 1e3)
same = torch.equal(outs[0][0], outs[1][0]) and (RAYS or torch.equal(outs[0][1], outs[1][1]))
print(f"{':'.join(spec)} {w}x{h}: records {'identical' if same else 'DIFFER'}")
for repack in (0, 1):
    print(f"repack_candidates={repack}: {np.median(times[repack]):9.1f} us per frame (rounds: {', '.join(f'{t:.1f}' for t in times[repack])})")
