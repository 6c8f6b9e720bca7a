"""rt_params.persistent_waves: the frame's 8x8 pixel blocks taken from a queue by as many workgroups as the GPU holds,
instead of one workgroup per 16x16 pixels. The reference dispatches the plain grid (src/renderer.rs:865-867); which
wave traces a block must not matter, so the records are compared bit for bit with the plain launch and the oracle:
both bundled models, both constant sets, with and without shadow rays, frames off the 8- and 16-pixel raster, shards
(plain and with an owner table), the block mask, a frame of fewer blocks than the queue has takers, frames in flight
on several streams (one counter per stream), and more streams than the context keeps counters for."""
import math

import numpy as np
import pytest
import torch

from test_gpu_parity import assert_hits_equal, ocam, oparams

pytestmark = pytest.mark.gpu


def pair(rt, tris, cell_size=0.0, **kw):
    cores = []
    for queue in (1, 0):
        c = rt.TraceCore(0, rt.default_params(persistent_waves=queue, **kw))
        c.set_scene(None, tris)
        c.build_grid(cell_size=cell_size)
        cores.append(c)
    return cores


def same(on, off, cam, w, h, what, shard=None, shadow=True):
    p1, s1 = on.trace_primary(cam, w, h, shard=shard, shadow=shadow)
    p0, s0 = off.trace_primary(cam, w, h, shard=shard, shadow=shadow)
    assert_hits_equal(p1, p0, f"{what}: primary, block queue against the plain grid")
    if shadow:
        assert_hits_equal(s1, s0, f"{what}: shadow, block queue against the plain grid")
    return p1, s1


@pytest.mark.parametrize("wgsl", [0, 1])
@pytest.mark.parametrize("cull", [0, 2])
def test_block_queue_equals_the_plain_grid_and_the_oracle(rt, ob, car_xyz9, wgsl, cull):
    on, off = pair(rt, rt.triangles_from_xyz9(car_xyz9), wgsl_compat=wgsl, cull_empty_blocks=cull)
    try:
        scene = ob.Scene(None, ob.triangles_from_xyz9(car_xyz9))
        for name, pose, (w, h) in (("start pose", rt.Camera.for_scene("gltf"), (480, 270)),
                                    ("off the raster", rt.Camera((150, 120, 250), 3.4, -0.3), (203, 117)),
                                    ("inside the grid", rt.Camera((10, 30, 20), 0.7, -0.2), (160, 96)),
                                    ("fewer blocks than takers", rt.Camera.for_scene("gltf"), (40, 24))):
            cam = pose.to_uniform(0.0, float(h))
            for shadow in (False, True):
                p, s = same(on, off, cam, w, h, f"{name} shadow={shadow}", shadow=shadow)
            op, os_, _ = ob.trace_primary(scene, ocam(ob, cam), w, h, params=oparams(ob, on.params), n_threads=8)
            assert_hits_equal(p, op, f"{name}: primary against the oracle")
            assert_hits_equal(s, os_, f"{name}: shadow against the oracle")
        cam = rt.Camera.for_scene("gltf").to_uniform(0.0, 270.0)
        for shard in (rt.RtShard(32, 32, 1, 3), rt.RtShard(64, 16, 0, 2), rt.RtShard(32, 32, 1, owner=[0, 1, 1, 2, 1])):
            same(on, off, cam, 480, 270, f"shard {shard.tile_w}x{shard.tile_h}", shard=shard)
    finally:
        on.close(), off.close()


def test_block_queue_full_frame_and_frames_in_flight(rt, ob, car_xyz9, phoenix_xyz9):
    for xyz9, pose in ((car_xyz9, rt.Camera.for_scene("gltf")), (phoenix_xyz9, rt.Camera((-300, 250, 600), 2.7, -0.2))):
        on, off = pair(rt, rt.triangles_from_xyz9(xyz9))
        try:
            w, h = 1920, 1080
            cam = pose.to_uniform(0.0, float(h))
            want = [torch.empty((w * h, 2), dtype=torch.int32, device="cuda") for _ in range(2)]
            off.trace_primary(cam, w, h, out_primary=want[0], out_shadow=want[1])
            off.sync()
            # twelve streams: the context keeps eight queue counters, the other four frames take the plain grid
            streams = [torch.cuda.Stream() for _ in range(12)]
            outs = [(torch.zeros_like(want[0]), torch.zeros_like(want[1])) for _ in streams]
            for rnd in range(3):
                for st, (p, s) in zip(streams, outs):
                    on.set_stream(st.cuda_stream)
                    on.trace_primary(cam, w, h, out_primary=p, out_shadow=s)
            on.sync()
            for k, (p, s) in enumerate(outs):
                assert torch.equal(p, want[0]) and torch.equal(s, want[1]), f"frame on stream {k} differs from the plain grid"
            on.set_stream(None)
        finally:
            on.close(), off.close()
