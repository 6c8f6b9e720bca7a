"""The multi-GPU entry points of the C ABI (rt_comm_*, rt_broadcast_scene, rt_gather_frames) on the real backend, as far
as one GPU allows: a communicator of ONE rank. RCCL refuses two ranks on one device, so N > 1 is covered by the gloo
tests of the sharding logic (tests/test_sharding_gloo.py) and runs on the driver's 8-GPU node."""
import math

import numpy as np
import pytest

from test_gpu_parity import assert_hits_equal, random_boxes

pytestmark = pytest.mark.gpu


def test_one_rank_communicator_broadcast_and_gather(rt, ob):
    import torch
    core = rt.TraceCore(0)
    plain = rt.TraceCore(0)
    try:
        with pytest.raises(rt.RtError) as e:  # no communicator yet
            core.gather_frames(0, torch.zeros(16, dtype=torch.uint8, device="cuda"), 16, torch.zeros(16, dtype=torch.uint8, device="cuda"))
        assert e.value.code == rt.RT_ERR_STATE
        cid = rt.TraceCore.comm_unique_id()
        assert len(cid) == rt.RT_COMM_ID_BYTES
        core.comm_init(cid, 0, 1)
        with pytest.raises(rt.RtError) as e:
            core.comm_init(cid, 0, 1)
        assert e.value.code == rt.RT_ERR_STATE

        # the scene arrives through the broadcast: same grid, same hits as a context that was handed the arrays
        boxes, tris = random_boxes(ob, 30, 5, extent=45.0, moving_last=3), ob.generate_test_triangles(5000)
        assert core.broadcast_scene(0, boxes, tris) == (30, 5000)
        plain.set_scene(boxes, tris)
        a, b = core.build_grid(cell_size=8.0), plain.build_grid(cell_size=8.0)
        assert bytes(a[0]) == bytes(b[0]) and a[1:] == b[1:]
        for x, y in zip(core.read_grid_csr(), plain.read_grid_csr()):
            assert np.array_equal(x, y)
        cam = rt.Camera((10, 30, 140), math.pi, -0.2).to_uniform(0.7, 90.0)
        w, h = 160, 90
        p, s = core.trace_primary(cam, w, h)
        q, t = plain.trace_primary(cam, w, h)
        assert_hits_equal(p, q, "broadcast scene primary")
        assert_hits_equal(s, t, "broadcast scene shadow")
        scene = ob.Scene(boxes, tris, cell_size=8.0)
        op, os_, _ = ob.trace_primary(scene, ob.camera_to_uniform((10, 30, 140), math.pi, -0.2, 0.7, 90.0), w, h)
        assert_hits_equal(p, op, "broadcast scene vs oracle")
        assert (p["id"] != rt.RT_MISS).sum() > 200
        # an empty scene travels too
        assert core.broadcast_scene(0, None, None) == (0, 0)
        core.broadcast_scene(0, boxes, tris)
        core.build_grid(cell_size=8.0)

        # gather: the lone rank's block lands in slot 0, asynchronously on the context's stream
        sh = rt.RtShard(32, 32, 0, 1)
        n = rt.shard_pixels(w, h, sh)
        local = torch.empty((2, n, 2), dtype=torch.int32, device="cuda")
        gathered = torch.zeros((1, 2, n, 2), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        core.trace_primary(cam, w, h, shard=sh, out_primary=local[0], out_shadow=local[1])
        core.gather_frames(0, local, local.numel() * 4, gathered)
        image = torch.empty((2, w * h, 2), dtype=torch.int32, device="cuda")
        for k in range(2):
            core.assemble_tiles(gathered[0, k], w, h, 32, 32, 1, image[k], rank_stride=2 * n)
        core.sync()
        assert torch.equal(gathered[0], local)
        assert_hits_equal(image[0].cpu().numpy().view(rt.HIT_DTYPE).reshape(-1), p, "gathered + assembled primary")
        assert_hits_equal(image[1].cpu().numpy().view(rt.HIT_DTYPE).reshape(-1), s, "gathered + assembled shadow")
        with pytest.raises(rt.RtError):  # host memory is refused
            core.gather_frames(0, np.zeros(16, np.uint8), 16, gathered)
        with pytest.raises(rt.RtError):
            core.gather_frames(3, local, 16, gathered)  # no such rank

        # the frame sharder on a core with a communicator takes the C-ABI route (groups of frames, RGB8 on the wire)
        from ray_tracer_amd import sharding
        shd = sharding.FrameSharder(core, w, h, 32, 32, 0, 1, "cuda", shadow=True, mode="rgba", max_bounces=1, collective=True)
        assert shd.native_comm
        shd.enable_groups(4, 2, rgb8=True)
        cams = [rt.Camera((10, 30, 140), math.pi, -0.2).to_uniform(0.1 * k, 90.0) for k in range(3)]
        pics = shd.trace_group(cams)
        shd.synchronize()
        for k, c in enumerate(cams):
            want = plain.render(c, w, h, max_bounces=1, shadows=True)
            assert np.array_equal(pics[k].cpu().numpy(), want), f"frame {k}"
        core.comm_destroy()
        with pytest.raises(rt.RtError):
            core.broadcast_scene(0, boxes, tris)
    finally:
        core.close()
        plain.close()
