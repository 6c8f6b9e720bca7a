#!/usr/bin/env python3
"""Development tool: GPU time of the gathering rank's assemble kernels on an otherwise idle GPU, per 1080p frame,
for the layouts bench.py uses at N ranks (32x32 tiles, groups of 32 frames). Random bytes in, result checked against
a numpy scatter of the same bytes."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import ray_tracer_amd as rt  # noqa: E402
from ray_tracer_amd import sharding  # noqa: E402

W, H, T, G = 1920, 1080, int(os.environ.get("RT_TILE", "32")), 32
core = rt.TraceCore(0)


def expected(packed, world, wire):
    """numpy restatement: frame 0 only."""
    tx, ty = -(-W // T), -(-H // T)
    img = np.zeros((H, W, 4), np.uint8)
    img[..., 3] = 255
    for tile in range(tx * ty):
        r, lt = tile % world, tile // world
        blk = packed[r, 0, lt * T * T:(lt + 1) * T * T].reshape(T, T, wire)
        x0, y0 = (tile % tx) * T, (tile // tx) * T
        h, w = min(T, H - y0), min(T, W - x0)
        img[y0:y0 + h, x0:x0 + w, :wire] = blk[:h, :w]
    return img.reshape(H * W, 4)


for world in (1, 2, 4, 8):
    per_rank = sharding.per_rank_pixels(W, H, T, T, world)
    for wire, fn in ((3, core.assemble_frames_rgb8), (4, core.assemble_frames_rgba8)):
        packed = torch.randint(0, 256, (world, G, per_rank, wire), dtype=torch.uint8, device="cuda")
        if wire == 4:
            packed[..., 3] = 255
        images = torch.zeros((G, H * W, 4), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()  # the library works on its own stream
        fn(packed, W, H, T, T, world, G, G * per_rank, per_rank, images)
        core.sync()
        ok = np.array_equal(images[0].cpu().numpy(), expected(packed.cpu().numpy(), world, wire))
        for _ in range(3):
            fn(packed, W, H, T, T, world, G, G * per_rank, per_rank, images)
        core.timer_begin()
        for _ in range(10):
            fn(packed, W, H, T, T, world, G, G * per_rank, per_rank, images)
        us = core.timer_end() / 10 / G * 1e3
        traffic = (wire + 4) * W * H
        print(f"N={world} wire {wire} B/px: {us:6.2f} us per frame, {traffic / us / 1e6:5.2f} TB/s  {'ok' if ok else 'MISMATCH'}", flush=True)
