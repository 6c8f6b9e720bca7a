"""The block mask (rt_params.cull_empty_blocks): 8x8 pixel blocks whose primary rays provably meet no occupied grid cell
are written as misses without a traversal. The reference lets every ray walk the grid (raytracer_unified.wgsl:520-588)
and knows no such mask, so the check is: with the mask and without it the records are the same bit for bit - and equal
to the oracle's - for cameras outside, above, inside, on the bounds of and far away from the grid, for frame sizes that
are no multiple of 8, for shards, for both constant sets, and for random poses; and the mask does cull where there is
something to cull."""
import math

import numpy as np
import pytest

from test_gpu_parity import assert_hits_equal, ocam, oparams

pytestmark = pytest.mark.gpu


def both(rt, xyz9, **kw):
    cores = []
    for cull in (2, 0):  # 2: a mask for every frame (1, the default, only while earlier frames are still running)
        c = rt.TraceCore(0, rt.default_params(cull_empty_blocks=cull, **kw))
        c.set_scene(None, rt.triangles_from_xyz9(xyz9))
        c.build_grid()
        cores.append(c)
    return cores


def same(on, off, cam, w, h, what, shard=None):
    p1, s1 = on.trace_primary(cam, w, h, shard=shard)
    stats = on.cull_stats()
    p0, s0 = off.trace_primary(cam, w, h, shard=shard)
    assert off.cull_stats() == (0, 0)
    assert_hits_equal(p1, p0, f"{what}: primary, mask on against off")
    assert_hits_equal(s1, s0, f"{what}: shadow, mask on against off")
    return p1, s1, stats


POSES = [
    ("scene start camera", (200, 200, 300), 3.35, -0.28),
    ("far outside, model small in the frame", (2500, 900, 2600), 3.9, -0.25),
    ("outside, looking away", (900, 300, 900), 0.6, 0.1),
    ("outside, looking straight down", (0, 900, 0), 0.0, -1.55),
    ("inside the grid", (40, 60, -30), 1.0, -0.1),
    ("inside, close above the floor", (-100, 3, 200), 2.2, 0.3),
    ("grazing along the roof", (-700, 190, -900), 0.65, -0.02),
    ("below the model looking up", (0, -400, 0), 0.3, 1.2),
]


@pytest.mark.parametrize("wgsl", [0, 1])
def test_car_poses_mask_on_equals_off_and_oracle(rt, ob, car_xyz9, wgsl):
    on, off = both(rt, car_xyz9, wgsl_compat=wgsl)
    try:
        scene = ob.Scene(None, ob.triangles_from_xyz9(car_xyz9))
        culled_somewhere = 0
        for k, (name, pos, yaw, pitch) in enumerate(POSES):
            for (w, h) in ((480, 270), (333, 187)):
                cam = rt.Camera(pos, yaw, pitch).to_uniform(0.0, float(h))
                p, s, (blocks, kept) = same(on, off, cam, w, h, f"{name} {w}x{h}")
                assert blocks == -(-w // 8) * -(-h // 8), (name, blocks)
                ids = p["id"].reshape(h, w)
                hit_blocks = sum(bool((ids[by:by + 8, bx:bx + 8] != rt.RT_MISS).any()) for by in range(0, h, 8) for bx in range(0, w, 8))
                assert kept >= hit_blocks, (name, kept, hit_blocks)  # a block with a hit is never culled
                culled_somewhere += blocks - kept
                if k < 3 and (w, h) == (480, 270):  # against the oracle as well
                    op, os_, _ = ob.trace_primary(scene, ocam(ob, cam), w, h, params=oparams(ob, on.params), n_threads=8)
                    assert_hits_equal(p, op, f"{name}: primary against the oracle")
                    assert_hits_equal(s, os_, f"{name}: shadow against the oracle")
        assert culled_somewhere > 0
    finally:
        on.close(), off.close()


def test_mask_culls_most_of_an_empty_sky_and_keeps_the_model(rt, car_xyz9):
    on, off = both(rt, car_xyz9)
    try:
        w, h = 960, 540
        cam = rt.Camera((2500, 900, 2600), 3.9, -0.25).to_uniform(0.0, float(h))
        p, _, (blocks, kept) = same(on, off, cam, w, h, "far camera")
        ids = p["id"].reshape(h, w)
        hit_blocks = sum(bool((ids[by:by + 8, bx:bx + 8] != rt.RT_MISS).any()) for by in range(0, h, 8) for bx in range(0, w, 8))
        assert hit_blocks > 0 and kept >= hit_blocks
        # the grown cells' outline is a little larger than the model's: allow for it, but most of the frame must go
        assert kept < 0.5 * blocks and kept < 4 * hit_blocks + 200, (blocks, kept, hit_blocks)
    finally:
        on.close(), off.close()


def test_shards_and_tiles_that_do_not_line_up(rt, car_xyz9):
    on, off = both(rt, car_xyz9)
    try:
        w, h = 500, 281
        cam = rt.Camera.for_scene("gltf").to_uniform(0.0, float(h))
        for (tw, th, first, stride) in ((32, 32, 0, 1), (64, 40, 1, 3), (32, 16, 2, 5), (8, 8, 0, 2)):
            _, _, (blocks, kept) = same(on, off, cam, w, h, f"tiles {tw}x{th}", shard=rt.RtShard(tw, th, first, stride))
            assert blocks > 0  # tiles of whole 8x8 blocks: a mask is made
        for (tw, th) in ((30, 32), (32, 20), (17, 9)):  # no mask: a wave's block would straddle the frame's blocks
            _, _, st = same(on, off, cam, w, h, f"tiles {tw}x{th}", shard=rt.RtShard(tw, th, 0, 2))
            assert st == (0, 0)
        # one tile = the whole frame, whatever its size
        _, _, st = same(on, off, cam, w, h, "whole frame")
        assert st[0] == -(-w // 8) * -(-h // 8)
    finally:
        on.close(), off.close()


def test_random_poses_two_models_frames_in_flight(rt, ob, car_xyz9, phoenix_xyz9):
    import torch
    rng = np.random.default_rng(20261112)
    for name, xyz9 in (("car", car_xyz9), ("phoenix", phoenix_xyz9)):
        on, off = both(rt, xyz9)
        try:
            meta, _, _ = on.grid_info()
            lo, hi = np.array(meta.bounds_min[:3]), np.array(meta.bounds_max[:3])
            size = hi - lo
            streams = [torch.cuda.Stream() for _ in range(3)]
            for trial in range(40):
                kind = trial % 4
                if kind == 0:    # anywhere inside
                    pos = lo + rng.uniform(0, 1, 3) * size
                elif kind == 1:  # a shell around the grid
                    pos = lo + rng.uniform(-0.6, 1.6, 3) * size
                elif kind == 2:  # far away
                    pos = (lo + hi) / 2 + rng.normal(0, 1, 3) * size.max() * 4
                else:            # exactly on a face of the bounds, or within a hair of it
                    pos = lo + rng.uniform(0, 1, 3) * size
                    ax = int(rng.integers(3))
                    pos[ax] = (lo, hi)[int(rng.integers(2))][ax] + (0.0, 1e-4, -1e-4)[int(rng.integers(3))]
                yaw, pitch = rng.uniform(0, 2 * math.pi), rng.uniform(-1.5, 1.5)
                w, h = int(rng.integers(40, 400)), int(rng.integers(30, 260))
                cam = rt.Camera(tuple(float(v) for v in pos), yaw, pitch).to_uniform(float(rng.uniform(0, 3)), float(h))
                on.set_stream(streams[trial % 3].cuda_stream)  # masks are per stream
                same(on, off, cam, w, h, f"{name} trial {trial} kind {kind} pos {pos} yaw {yaw:.3f} pitch {pitch:.3f} {w}x{h}")
        finally:
            on.close(), off.close()


def test_degenerate_camera_basis_gets_no_mask(rt, car_xyz9):
    on, off = both(rt, car_xyz9)
    try:
        w, h = 160, 90
        cam = rt.Camera.for_scene("gltf").to_uniform(0.0, float(h))
        cam.right[0], cam.right[1], cam.right[2] = cam.forward[0], cam.forward[1], cam.forward[2]  # not a basis
        _, _, st = same(on, off, cam, w, h, "right = forward")
        assert st == (0, 0)
        # a basis that is skewed and scaled is still a basis: the mask follows the pixel directions the kernel makes
        cam = rt.Camera.for_scene("gltf").to_uniform(0.0, float(h))
        cam.right[0] *= 1.7
        cam.up[1] *= 0.6
        cam.up[0] += 0.2
        _, _, st = same(on, off, cam, w, h, "skewed basis")
        assert st[0] > 0
    finally:
        on.close(), off.close()


def test_default_mode_masks_only_with_three_frames_in_flight_or_frame_groups(rt, car_xyz9):
    import torch
    core = rt.TraceCore(0, rt.default_params())  # cull_empty_blocks = 1
    off = rt.TraceCore(0, rt.default_params(cull_empty_blocks=0))
    try:
        for c in (core, off):
            c.set_scene(None, rt.triangles_from_xyz9(car_xyz9))
            c.build_grid()
        w, h = 960, 540
        cam = rt.Camera((2500, 900, 2600), 3.9, -0.25).to_uniform(0.0, float(h))
        want_p, want_s = off.trace_primary(cam, w, h)
        # one frame at a time (the call returns host arrays, so the GPU is idle when the next one starts): no mask
        for _ in range(3):
            p, s = core.trace_primary(cam, w, h)
            assert core.cull_stats() == (0, 0)
            assert_hits_equal(p, want_p, "lone frame: primary")
            assert_hits_equal(s, want_s, "lone frame: shadow")
        # two frames in flight (two streams): the launch would cost what the mask saves, so none is made
        streams = [torch.cuda.Stream() for _ in range(3)]
        outs = [(torch.empty((w * h, 2), dtype=torch.int32, device="cuda"), torch.empty((w * h, 2), dtype=torch.int32, device="cuda"))
                for _ in range(3)]
        for k in range(12):
            core.set_stream(streams[k % 2].cuda_stream)
            core.trace_primary(cam, w, h, out_primary=outs[k % 2][0], out_shadow=outs[k % 2][1])
        assert core.cull_stats() == (0, 0)
        core.sync()
        # three in flight: a frame queued after frames on two other streams (since the last rt_sync) gets a mask
        for k in range(36):
            core.set_stream(streams[k % 3].cuda_stream)
            core.trace_primary(cam, w, h, out_primary=outs[k % 3][0], out_shadow=outs[k % 3][1])
        masked = core.cull_stats()  # of the last launch; waits for its stream
        core.sync()  # rt_sync: nothing is in flight any more, the next frame is a lone one again
        assert masked[0] == -(-w // 8) * -(-h // 8) and 0 < masked[1] < masked[0], masked
        for op, os_ in outs:
            assert_hits_equal(op.cpu().numpy().view(rt.HIT_DTYPE).reshape(-1), want_p, "frames in flight: primary")
            assert_hits_equal(os_.cpu().numpy().view(rt.HIT_DTYPE).reshape(-1), want_s, "frames in flight: shadow")
        # the decision follows the call sequence alone: the same three-stream sequence again takes the same decisions
        for k in range(3):
            core.set_stream(streams[k].cuda_stream)
            core.trace_primary(cam, w, h, out_primary=outs[k][0], out_shadow=outs[k][1])
            assert (core.cull_stats() != (0, 0)) == (k == 2), k
            if k < 2:  # cull_stats waited for that stream: queue its frame again so that it counts as in flight
                core.trace_primary(cam, w, h, out_primary=outs[k][0], out_shadow=outs[k][1])
        core.sync()
        # a group of frames in one launch shares the mask launch: masks even on an idle GPU; fewer than four frames: none
        core.set_stream(streams[0].cuda_stream)
        cams = [rt.Camera((2500 - 100 * k, 900, 2600), 3.9, -0.25).to_uniform(0.0, float(h)) for k in range(4)]
        a = core.render_frames(cams, w, h, max_bounces=1, shadows=True)
        assert core.cull_stats()[0] > 0
        b = off.render_frames(cams, w, h, max_bounces=1, shadows=True)
        assert np.array_equal(a, b)
        core.render_frames(cams[:3], w, h, max_bounces=1, shadows=True)
        assert core.cull_stats() == (0, 0)
    finally:
        core.close(), off.close()


def test_pictures_with_the_mask_equal_pictures_without(rt, ob, car_xyz9):
    """rt_render and the frame groups of rt_render_frames: a culled block is sky, byte for byte what the traversal's miss
    gives (raytracer_unified.wgsl:591-596)."""
    on, off = both(rt, car_xyz9)
    try:
        w, h = 480, 270
        for name, pos, yaw, pitch in POSES:
            cam = rt.Camera(pos, yaw, pitch).to_uniform(0.0, float(h))
            for bounces, shadows in ((8, False), (1, True)):
                a, b = on.render(cam, w, h, max_bounces=bounces, shadows=shadows), off.render(cam, w, h, max_bounces=bounces, shadows=shadows)
                assert on.cull_stats()[0] > 0 and off.cull_stats() == (0, 0)
                assert np.array_equal(a, b), f"{name}: picture, {bounces} bounces"
        # sixteen frames of a camera path in one launch, every frame its own mask; shards; the three-byte format
        cams = [rt.Camera((2500 - 150 * k, 900 - 40 * k, 2600 - 160 * k), 3.9 + 0.01 * k, -0.25).to_uniform(0.1 * k, float(h)) for k in range(16)]
        for shard in (None, rt.RtShard(32, 32, 1, 3), rt.RtShard(64, 40, 0, 2)):
            for rgb8 in (False, True):
                a = on.render_frames(cams, w, h, shard=shard, max_bounces=1, shadows=True, rgb8=rgb8)
                b = off.render_frames(cams, w, h, shard=shard, max_bounces=1, shadows=True, rgb8=rgb8)
                assert on.cull_stats()[0] > 0
                assert np.array_equal(a, b), f"frame group, shard {shard is not None}, rgb8 {rgb8}"
        frames = on.render_frames(cams, w, h, max_bounces=1, shadows=True)
        assert not np.array_equal(frames[0], frames[15])  # the path moves
        # against the oracle, one picture
        scene = ob.Scene(None, ob.triangles_from_xyz9(car_xyz9))
        cam = rt.Camera((2500, 900, 2600), 3.9, -0.25).to_uniform(0.0, float(h))
        want, _ = ob.render(scene, ocam(ob, cam), w, h, params=oparams(ob, on.params), max_bounces=8)
        assert np.array_equal(on.render(cam, w, h), want)
    finally:
        on.close(), off.close()


def test_frames_of_a_camera_path_in_one_launch(rt, ob, car_xyz9):
    """rt_trace_primary_frames == rt_trace_primary frame by frame: device outputs take the one-launch kernel (with and without
    block masks, shards, a frame stride, primary only), host outputs and scenes with boxes the frame-by-frame route."""
    import torch
    w, h = 333, 187
    cams = [rt.Camera((2500 - 150 * k, 900 - 40 * k, 2600 - 160 * k), 3.9 + 0.01 * k, -0.25).to_uniform(0.1 * k, float(h)) for k in range(16)]
    for cull in (2, 1, 0):
        core = rt.TraceCore(0, rt.default_params(cull_empty_blocks=cull))
        try:
            core.set_scene(None, rt.triangles_from_xyz9(car_xyz9))
            core.build_grid()
            for shard in (None, rt.RtShard(32, 32, 1, 3), rt.RtShard(30, 20, 0, 2)):
                n = rt.shard_pixels(w, h, shard) if hasattr(rt, "shard_pixels") else None
                want = [core.trace_primary(cam, w, h, shard=shard) for cam in cams]
                n = len(want[0][0])
                for stride in (0, n + 37):
                    fs = stride or n
                    dp = torch.full((16 * fs, 2), -7, dtype=torch.int32, device="cuda")
                    ds = torch.full((16 * fs, 2), -7, dtype=torch.int32, device="cuda")
                    core.trace_primary_frames(cams, w, h, shard=shard, out_primary=dp, out_shadow=ds, frame_stride=stride)
                    core.sync()
                    gp = dp.cpu().numpy().view(rt.HIT_DTYPE).reshape(16, fs)
                    gs = ds.cpu().numpy().view(rt.HIT_DTYPE).reshape(16, fs)
                    for k in range(16):
                        assert_hits_equal(gp[k, :n], want[k][0], f"cull {cull} frame {k} primary (stride {stride})")
                        assert_hits_equal(gs[k, :n], want[k][1], f"cull {cull} frame {k} shadow (stride {stride})")
                    if stride:
                        assert (gp[:, n:]["id"].view(np.int32) == -7).all()  # the gap between frames is not written
                    if cull == 2 and (shard is None or shard.tile_w % 8 == 0):
                        assert core.cull_stats()[0] > 0
            # primary only, five frames, host arrays (the frame-by-frame route)
            hp, hs = core.trace_primary_frames(cams[:5], w, h, shadow=False)
            assert hs is None
            for k in range(5):
                assert_hits_equal(hp[k], core.trace_primary(cams[k], w, h, shadow=False)[0], f"host output frame {k}")
            dp = torch.empty((5 * w * h, 2), dtype=torch.int32, device="cuda")
            core.trace_primary_frames(cams[:5], w, h, shadow=False, out_primary=dp)
            core.sync()
            gp = dp.cpu().numpy().view(rt.HIT_DTYPE).reshape(5, w * h)
            for k in range(5):
                assert_hits_equal(gp[k], hp[k], f"primary only, device output, frame {k}")
        finally:
            core.close()
    # against the oracle, one group
    core = rt.TraceCore(0, rt.default_params(cull_empty_blocks=2))
    try:
        core.set_scene(None, rt.triangles_from_xyz9(car_xyz9))
        core.build_grid()
        scene = ob.Scene(None, ob.triangles_from_xyz9(car_xyz9))
        dp = torch.empty((4 * w * h, 2), dtype=torch.int32, device="cuda")
        ds = torch.empty_like(dp)
        core.trace_primary_frames(cams[:4], w, h, out_primary=dp, out_shadow=ds)
        core.sync()
        gp, gs = (t.cpu().numpy().view(rt.HIT_DTYPE).reshape(4, w * h) for t in (dp, ds))
        for k in range(4):
            op, os_, _ = ob.trace_primary(scene, ocam(ob, cams[k]), w, h, params=oparams(ob, core.params), n_threads=8)
            assert_hits_equal(gp[k], op, f"oracle frame {k} primary")
            assert_hits_equal(gs[k], os_, f"oracle frame {k} shadow")
    finally:
        core.close()


def test_large_frames(rt, car_xyz9):
    """3840x2160 has a mask (4 050 words); 7680x4320 would need more LDS than a wave of the mask kernel may take and is
    traced without one. Sampled rows of the frames with the mask against without."""
    import torch
    on, off = both(rt, car_xyz9)
    try:
        for (w, h, masked) in ((3840, 2160, True), (7680, 4320, False)):
            cam = rt.Camera((2500, 900, 2600), 3.9, -0.25).to_uniform(0.0, float(h))
            outs = []
            for c in (on, off):
                p = torch.empty((w * h, 2), dtype=torch.int32, device="cuda")
                s = torch.empty_like(p)
                c.trace_primary(cam, w, h, out_primary=p, out_shadow=s)
                c.sync()
                outs.append((p, s))
            assert (on.cull_stats()[0] > 0) == masked, on.cull_stats()
            assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), f"{w}x{h}"
            del outs
    finally:
        on.close(), off.close()
