"""The shaded picture (rt_render) against the CPU oracle, through the C ABI: every byte equal."""
import ctypes as C
import math
import os

import numpy as np
import pytest

from render_scenes import demo_boxes, demo_materials

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def ocam(ob, cam):
    o = ob.CameraUniform()
    C.memmove(C.byref(o), C.byref(cam), C.sizeof(o))
    return o


def oparams(ob, p):
    return ob.Params.default(**{k: getattr(p, k) for k in ("finest_cell_size", "max_objects_per_cell", "max_dda_steps",
                                                           "wgsl_compat", "lod_cull", "moving_box_count", "fov",
                                                           "shadow_offset")})


def check(core, ob, scene, cam, w, h, materials=None, what="", **kw):
    got = core.render(cam, w, h, **kw)
    okw = dict(kw)
    if "shard" in okw and okw["shard"] is not None:
        s = okw["shard"]
        okw["shard"] = ob.Shard(s.tile_w, s.tile_h, s.first, s.stride, s.owner_list())
    want, st = ob.render(scene, ocam(ob, cam), w, h, params=oparams(ob, core.params), materials=materials, **okw)
    bad = np.flatnonzero((got != want).any(axis=1))
    assert bad.size == 0, f"{what}: {bad.size} pixels differ, first {bad[:5]}: {got[bad[:5]]} vs {want[bad[:5]]}"
    return want, st


def test_committed_pictures(core, ob, rt, car_xyz9):
    g = np.load(os.path.join(HERE, "golden", "render_golden.npz"))
    tris = ob.pyramid_triangles()
    tris["material_id"] = [0, 1, 2, 7, 0, 1]
    core.set_scene(None, tris)
    core.set_materials(demo_materials(ob))
    core.build_grid(cell_size=1.0)
    cam = rt.Camera.for_scene("pyramid").to_uniform(0.0, 36.0)
    assert np.array_equal(core.render(cam, 64, 36), g["pyramid_rgba"])
    core.set_materials(None)
    core.set_scene(None, rt.triangles_from_xyz9(car_xyz9))
    core.build_grid()
    cam = rt.Camera.for_scene("gltf").to_uniform(0.0, 36.0)
    assert np.array_equal(core.render(cam, 64, 36), g["car_rgba"])
    assert np.array_equal(core.render(cam, 64, 36, shadows=True), g["car_rgba_shadows"])
    core.set_scene(demo_boxes(ob), None)
    core.build_grid(cell_size=4.0)
    cam = rt.Camera((4, 12, 40), math.pi, -0.25).to_uniform(0.4, 54.0)
    assert np.array_equal(core.render(cam, 96, 54), g["boxes_rgba"])


@pytest.mark.parametrize("w,h", [(160, 90), (1920, 1080)])
def test_car(core, ob, rt, car_xyz9, w, h):
    tris = ob.triangles_from_xyz9(car_xyz9)
    core.set_scene(None, tris)
    core.set_materials(None)
    core.build_grid()
    scene = ob.Scene(None, tris)
    cam = rt.Camera.for_scene("gltf").to_uniform(0.0, float(h))
    check(core, ob, scene, cam, w, h, what="car")
    _, st = check(core, ob, scene, cam, w, h, shadows=True, what="car shadows")
    assert st["shadow_hits"] > 0


def test_materials_and_reflections(core, ob, rt, car_xyz9):
    """Metallic materials on the car: up to eight bounces per pixel (raytracer_unified.wgsl:655-688)."""
    tris = ob.triangles_from_xyz9(car_xyz9)
    tris["material_id"] = np.arange(len(tris)) % 5  # 3 and 4 are out of range -> grey
    mats = demo_materials(ob)
    mats["metallic"] = [0.3, 0.0, 0.85]
    core.set_scene(None, tris)
    core.set_materials(mats)
    core.build_grid()
    scene = ob.Scene(None, tris)
    for pos, yaw, pitch in (((200, 200, 300), 3.35, -0.28), ((0, 150, 50), 1.0, -0.4)):
        cam = rt.Camera(pos, yaw, pitch).to_uniform(0.0, 135.0)
        _, st = check(core, ob, scene, cam, 240, 135, materials=mats, what=f"materials from {pos}")
        assert st["rays"] > 240 * 135 * 1.1
        check(core, ob, scene, cam, 240, 135, materials=mats, shadows=True, max_bounces=3, what="materials, shadows, 3 bounces")
    core.set_materials(None)


def test_boxes_moving_mirror_overlay_debug(core, ob, rt):
    boxes = demo_boxes(ob)
    core.set_scene(boxes, None)
    core.build_grid(cell_size=4.0)
    scene = ob.Scene(boxes, None, cell_size=4.0)
    for time in (0.0, 0.4, 2.2):
        cam = rt.Camera((4, 12, 40), math.pi, -0.25).to_uniform(time, 108.0)
        _, st = check(core, ob, scene, cam, 192, 108, what=f"boxes t={time}")
        assert st["rays"] > 192 * 108 * 1.2
    cam = rt.Camera((4, 12, 40), math.pi, -0.25).to_uniform(0.4, 108.0, show_grid=True)
    check(core, ob, scene, cam, 192, 108, what="grid overlay")
    check(core, ob, scene, cam, 192, 108, debug_pixel=(100, 60), what="debug pixel")
    check(core, ob, scene, cam, 192, 108, debug_pixel=(0, 0), shadows=True, max_bounces=2, what="debug corner")


def test_mixed_scene_both_constant_sets(rt, ob):
    from test_gpu_parity import random_boxes
    for wgsl in (0, 1):
        core = rt.TraceCore(0, rt.default_params(wgsl_compat=wgsl))
        boxes = random_boxes(ob, 40, 11, extent=45.0, moving_last=3)
        boxes["reflectivity"] = np.linspace(0.0, 0.9, len(boxes), dtype=np.float32)
        boxes["color"] = np.random.default_rng(2).uniform(0, 1, (len(boxes), 3)).astype(np.float32)
        tris = ob.generate_test_triangles(3000)
        tris["material_id"] = np.arange(len(tris)) % 4
        mats = demo_materials(ob)
        core.set_scene(boxes, tris)
        core.set_materials(mats)
        core.build_grid(cell_size=8.0)
        scene = ob.Scene(boxes, tris, cell_size=8.0)
        cam = rt.Camera((10, 30, 140), math.pi, -0.2).to_uniform(0.6, 90.0)
        check(core, ob, scene, cam, 160, 90, materials=mats, what=f"mixed wgsl={wgsl}")
        core.close()


def test_shards_device_output_and_assembly(core, ob, rt, car_xyz9):
    import torch
    tris = ob.triangles_from_xyz9(car_xyz9)
    core.set_scene(None, tris)
    core.set_materials(None)
    core.build_grid()
    scene = ob.Scene(None, tris)
    w, h, tw, th, n = 300, 170, 64, 64, 3
    cam = rt.Camera.for_scene("gltf").to_uniform(0.0, float(h))
    full, _ = check(core, ob, scene, cam, w, h, shadows=True, what="full")
    per_rank = max(rt.shard_pixels(w, h, rt.RtShard(tw, th, r, n)) for r in range(n))
    gathered = torch.zeros((n, per_rank, 4), dtype=torch.uint8, device="cuda")
    for r in range(n):
        shard = rt.RtShard(tw, th, r, n)
        check(core, ob, scene, cam, w, h, shadows=True, shard=shard, what=f"shard {r}")
        core.render(cam, w, h, shard=shard, shadows=True, out=gathered[r])
    image = torch.empty((h * w, 4), dtype=torch.uint8, device="cuda")
    core.assemble_tiles_rgba8(gathered, w, h, tw, th, n, image, rank_stride=per_rank)
    core.sync()
    assert np.array_equal(image.cpu().numpy(), full)


def test_frame_sharder_on_the_gpu_single_rank(core, ob, rt, car_xyz9):
    """The multi-GPU driver with world size 1 (no collective): tile packing, per-frame streams and buffers,
    RGBA and hit-record modes, through the real kernels."""
    from ray_tracer_amd import sharding
    tris = ob.triangles_from_xyz9(car_xyz9)
    core.set_scene(None, tris)
    core.set_materials(None)
    core.build_grid()
    scene = ob.Scene(None, tris)
    w, h = 300, 170
    cams = [rt.Camera(p, y, pi).to_uniform(0.0, float(h)) for p, y, pi in
            (((200, 200, 300), 3.35, -0.28), ((0, 150, 50), 1.0, -0.4), ((-400, 50, -500), 0.7, 0.0))]
    sh = sharding.FrameSharder(core, w, h, 64, 64, rank=0, world=1, device="cuda", shadow=True, mode="rgba",
                               frames_in_flight=2, max_bounces=1)
    pics = [sh.trace(c) for c in cams]
    sh.synchronize()
    for k in (1, 2):  # frame 0's buffers were reused by frame 2
        want, _ = ob.render(scene, ocam(ob, cams[k]), w, h, max_bounces=1, shadows=True)
        assert np.array_equal(pics[k].cpu().numpy(), want), f"rgba frame {k}"
    sh = sharding.FrameSharder(core, w, h, 64, 64, rank=0, world=1, device="cuda", shadow=True, mode="hits",
                               frames_in_flight=3)
    outs = [sh.trace(c) for c in cams]
    sh.synchronize()
    core.set_stream(None)
    for k, c in enumerate(cams):
        wp, ws, _ = ob.trace_primary(scene, ocam(ob, c), w, h)
        gp = outs[k][0].cpu().numpy().view(rt.HIT_DTYPE).reshape(-1)
        gs = outs[k][1].cpu().numpy().view(rt.HIT_DTYPE).reshape(-1)
        assert np.array_equal(gp["id"], wp["id"]) and np.array_equal(gp["t"].view(np.uint32), wp["t"].view(np.uint32))
        assert np.array_equal(gs["id"], ws["id"]) and np.array_equal(gs["t"].view(np.uint32), ws["t"].view(np.uint32))


def test_render_frames_equals_frame_by_frame(core, ob, rt, car_xyz9):
    """Several frames of a camera path in one launch: every frame byte for byte what rt_render gives for it, and the
    oracle's picture; boxes move and are LOD-culled per frame (time and position differ between the frames)."""
    from render_scenes import demo_boxes
    import torch
    boxes = demo_boxes(ob)
    tris = ob.triangles_from_xyz9(car_xyz9[:600] * np.float32(0.1))
    core.set_scene(boxes, tris)
    core.set_materials(None)
    core.build_grid(cell_size=4.0)
    scene = ob.Scene(boxes, tris, cell_size=4.0)
    w, h = 200, 120
    cams = [rt.Camera((4.0 + 3 * j, 12.0 + j, 40.0 - 2 * j), math.pi + 0.05 * j, -0.25).to_uniform(0.3 * j, float(h), show_grid=(j == 2))
            for j in range(16)]
    got = core.render_frames(cams, w, h, max_bounces=3, shadows=True)
    assert got.shape == (16, w * h, 4)
    for j, cam in enumerate(cams):
        assert np.array_equal(got[j], core.render(cam, w, h, max_bounces=3, shadows=True)), f"frame {j} vs rt_render"
    for j in (0, 2, 9, 15):
        assert np.array_equal(got[j], ob.render(scene, ocam(ob, cams[j]), w, h, max_bounces=3, shadows=True)[0]), f"frame {j} vs oracle"
    # tile-packed and on the device
    shard = rt.RtShard(64, 32, 1, 3)
    n = rt.shard_pixels(w, h, shard)
    dev = torch.zeros((5, n, 4), dtype=torch.uint8, device="cuda")
    core.render_frames(cams[3:8], w, h, shard=shard, out=dev)
    core.sync()
    for j in range(5):
        assert np.array_equal(dev[j].cpu().numpy(), core.render(cams[3 + j], w, h, shard=shard)), f"sharded frame {j}"
    with pytest.raises(rt.RtError):
        core.render_frames(cams + cams[:1], w, h)


def test_frame_groups_on_the_gpu_single_rank(core, ob, rt, car_xyz9):
    """FrameSharder.trace_group with world size 1: one launch, one assemble per group of frames, two groups in flight."""
    from ray_tracer_amd import sharding
    tris = ob.triangles_from_xyz9(car_xyz9)
    core.set_scene(None, tris)
    core.set_materials(None)
    core.build_grid()
    scene = ob.Scene(None, tris)
    w, h = 300, 170
    cams = [rt.Camera((200.0 - 20 * j, 200.0, 300.0 - 10 * j), 3.35 + 0.02 * j, -0.28).to_uniform(0.0, float(h)) for j in range(7)]
    sh = sharding.FrameSharder(core, w, h, 64, 64, rank=0, world=1, device="cuda", shadow=True, mode="rgba", max_bounces=1)
    wants = [ob.render(scene, ocam(ob, c), w, h, max_bounces=1, shadows=True)[0] for c in cams]
    for rgb8 in (False, True):  # 4- and 3-byte pixels on the wire, the same pictures out
        sh.enable_groups(group=4, groups_in_flight=2, rgb8=rgb8)
        a, b = sh.trace_group(cams[:4]), sh.trace_group(cams[4:])
        sh.synchronize()
        core.set_stream(None)
        assert a.shape == (4, w * h, 4) and b.shape == (3, w * h, 4)
        for j, pic in enumerate(list(a) + list(b)):
            assert np.array_equal(pic.cpu().numpy(), wants[j]), f"frame {j}, rgb8={rgb8}"
    # a group longer than one launch holds (16 frames): 20 frames, two launches, one assemble
    sh.enable_groups(group=20, groups_in_flight=1, rgb8=True)
    many = sh.trace_group([cams[j % 7] for j in range(20)])
    sh.synchronize()
    core.set_stream(None)
    for j in (0, 15, 16, 19):
        assert np.array_equal(many[j].cpu().numpy(), wants[j % 7]), f"frame {j} of 20"
    # the 3-byte picture itself: host output, one frame and several
    got = core.render_frames(cams[:3], w, h, max_bounces=1, shadows=True, rgb8=True)
    assert got.shape == (3, w * h, 3)
    for j in range(3):
        assert np.array_equal(got[j], wants[j][:, :3])


def test_alpha_rules_hit_records_and_picture(core, ob, rt, car_xyz9):
    """Untextured alpha MASK / BLEND materials (raytracer_unified.wgsl:310-321) remove triangles from every
    ray's view; set before the grid, after it, and taken away again."""
    from test_oracle_render import alpha_materials
    tris = ob.triangles_from_xyz9(car_xyz9)
    tris["material_id"] = np.arange(len(tris)) % 6
    mats = alpha_materials(ob)
    scene = ob.Scene(None, tris, materials=mats)
    plain = ob.Scene(None, tris)
    w, h = 240, 135
    cam = rt.Camera.for_scene("gltf").to_uniform(0.0, float(h))

    def hits_equal(sc, what):
        p, s = core.trace_primary(cam, w, h)
        op, os_, _ = ob.trace_primary(sc, ocam(ob, cam), w, h)
        for got, want in ((p, op), (s, os_)):
            assert np.array_equal(got["id"], want["id"]), what
            assert np.array_equal(got["t"].view(np.uint32), want["t"].view(np.uint32)), what
        return op

    core.set_scene(None, tris)
    core.set_materials(mats)           # before the grid exists
    core.build_grid()
    op = hits_equal(scene, "materials before grid")
    assert not np.isin(op["id"][op["id"] != ob.MISS] % 6, (1, 3)).any()
    check(core, ob, scene, cam, w, h, what="picture with alpha")
    check(core, ob, scene, cam, w, h, shadows=True, max_bounces=2, what="picture with alpha, shadows")
    core.set_materials(None)           # taken away: everything visible again, grid untouched
    op0 = hits_equal(plain, "materials removed")
    assert np.isin(op0["id"][op0["id"] != ob.MISS] % 6, (1, 3)).any()
    core.set_materials(mats)           # after the grid exists
    hits_equal(scene, "materials after grid")
    cs, idx = core.read_grid_csr()     # caller-supplied grid takes the same rule
    meta = core.grid_info()[0]
    core.set_grid_csr(meta, cs, idx)
    hits_equal(scene, "caller-supplied grid")
    core.set_collect_stats(True)       # the counters still count the tests the shader makes
    core.trace_primary(cam, w, h)
    _, _, st = ob.trace_primary(scene, ocam(ob, cam), w, h)
    got = core.stats()
    assert {k: got[k] for k in st} == st
    core.set_collect_stats(False)
    core.set_materials(None)


def _scatter(packed, w, h, tw, th, world, n_frames, alpha, owner=None):
    """numpy restatement of the assemble kernels: packed is [rank][frame][record][bytes]; owner: the rank of every
    tile of a round (None: one tile each, in rank order)."""
    owner = owner or list(range(world))
    tx, ty = -(-w // tw), -(-h // th)
    per = packed.shape[-1]
    out = np.zeros((n_frames, h, w, per if alpha is None else 4), packed.dtype)
    if alpha is not None:
        out[..., 3] = alpha
    for tile in range(tx * ty):
        o = tile % len(owner)
        r = owner[o]
        lt = tile // len(owner) * owner.count(r) + owner[:o].count(r)
        x0, y0 = (tile % tx) * tw, (tile // tx) * th
        hh, ww = min(th, h - y0), min(tw, w - x0)
        blk = packed[r, :, lt * tw * th:(lt + 1) * tw * th].reshape(n_frames, th, tw, per)
        out[:, y0:y0 + hh, x0:x0 + ww, :per] = blk[:, :hh, :ww]
    return out.reshape(n_frames, h * w, -1)


@pytest.mark.parametrize("w,h,tw,th,world", [
    (1920, 1080, 32, 32, 8),   # what the benchmark uses: vector kernels, bottom tiles cut by the image
    (200, 120, 64, 16, 3),     # right and bottom tiles cut
    (202, 61, 20, 7, 2),       # width not a multiple of 4: the pixel-at-a-time kernels
    (120, 50, 6, 5, 5),        # tile width not a multiple of 4
    (64, 64, 64, 64, 1),
])
def test_assemble_kernels_equal_a_numpy_scatter(core, rt, w, h, tw, th, world):
    """The gathering rank's kernels move bytes and nothing else: every layout - 16-byte vector path or one pixel per
    thread - must equal a plain scatter of the same blocks (hit records, RGBA8 and RGB8 pictures, several frames)."""
    import torch
    from ray_tracer_amd import sharding
    rng = np.random.default_rng(w * 31 + tw)
    per_rank = sharding.per_rank_pixels(w, h, tw, th, world)
    n_frames = 3
    for per, dtype in ((3, np.uint8), (4, np.uint8)):
        packed = rng.integers(0, 256, (world, n_frames, per_rank, per)).astype(dtype)
        if per == 4:
            packed[..., 3] = 255
        dev = torch.from_numpy(packed).cuda()
        images = torch.zeros((n_frames, h * w, 4), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        fn = core.assemble_frames_rgb8 if per == 3 else core.assemble_frames_rgba8
        fn(dev, w, h, tw, th, world, n_frames, n_frames * per_rank, per_rank, images)
        core.sync()
        want = _scatter(packed, w, h, tw, th, world, n_frames, 255)
        assert np.array_equal(images.cpu().numpy(), want), f"{per}-byte pictures, {w}x{h} tiles {tw}x{th} over {world}"
    # unequal shares: the first rank owns 3 tiles of every round, the others 4 each
    owner = sharding.owner_table([3] + [4] * (world - 1))
    shards = [rt.RtShard(tw, th, r, owner=owner) for r in range(world)]
    per_w = max(rt.shard_pixels(w, h, s) for s in shards)
    for per in (3, 4):
        packed = rng.integers(0, 256, (world, n_frames, per_w, per)).astype(np.uint8)
        if per == 4:
            packed[..., 3] = 255
        dev = torch.from_numpy(packed).cuda()
        images = torch.zeros((n_frames, h * w, 4), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        core.assemble_frames(dev, per == 3, w, h, tw, th, world, owner, n_frames, n_frames * per_w, per_w, images)
        core.sync()
        want = _scatter(packed, w, h, tw, th, world, n_frames, 255, owner)
        assert np.array_equal(images.cpu().numpy(), want), f"{per}-byte pictures with owner table {owner}"
    # hit records (8 bytes), one frame, and the one-frame RGBA8 entry point
    hits = rng.integers(-2**31, 2**31 - 1, (world, 1, per_rank, 2)).astype(np.int32)
    dev = torch.from_numpy(hits).cuda()
    image = torch.zeros((h * w, 2), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    core.assemble_tiles(dev, w, h, tw, th, world, image, rank_stride=per_rank)
    core.sync()
    assert np.array_equal(image.cpu().numpy(), _scatter(hits, w, h, tw, th, world, 1, None)[0])
    pic = rng.integers(0, 256, (world, 1, per_rank, 4)).astype(np.uint8)
    dev = torch.from_numpy(pic).cuda()
    image = torch.zeros((h * w, 4), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    core.assemble_tiles_rgba8(dev, w, h, tw, th, world, image, rank_stride=per_rank)
    core.sync()
    assert np.array_equal(image.cpu().numpy(), _scatter(pic, w, h, tw, th, world, 1, None)[0])


def test_unequal_shares_render_and_assemble(core, ob, rt, car_xyz9):
    """Ranks that own different numbers of tiles per round (rt_shard.owner): each rank's packed output equals the
    oracle's for the same shard - hit records and the picture - and three frames rendered with a common frame stride,
    assembled with the counts, equal the whole pictures."""
    import torch
    tris = rt.triangles_from_xyz9(car_xyz9)
    core.set_scene(None, tris)
    core.set_materials(None)
    core.build_grid()
    scene = ob.Scene(None, ob.triangles_from_xyz9(car_xyz9))
    w, h, tw, th = 200, 120, 32, 16
    from ray_tracer_amd import sharding
    counts = [1, 3, 2]
    owner = sharding.owner_table(counts)
    cams = [rt.Camera.for_scene("gltf").to_uniform(0.0, float(h)), rt.Camera((0, 150, 50), 1.0, -0.4).to_uniform(0.0, float(h)),
            rt.Camera((150, 120, 250), 3.4, -0.3).to_uniform(0.0, float(h))]
    shards = [rt.RtShard(tw, th, r, owner=owner) for r in range(len(counts))]
    per = max(rt.shard_pixels(w, h, s) for s in shards)
    packed = torch.zeros((len(counts), len(cams), per, 3), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    for r, s in enumerate(shards):
        check(core, ob, scene, cams[0], w, h, what=f"rank {r} of counts {counts}", shard=s, max_bounces=2)
        p, sdw = core.trace_primary(cams[0], w, h, shard=s, shadow=True)
        op, osd, _ = ob.trace_primary(scene, ocam(ob, cams[0]), w, h, shard=ob.Shard(tw, th, s.first, s.stride, owner),
                                      params=oparams(ob, core.params), n_threads=4)
        assert np.array_equal(p["id"], op["id"]) and np.array_equal(p["t"].view(np.uint32), op["t"].view(np.uint32))
        assert np.array_equal(sdw["id"], osd["id"]) and np.array_equal(sdw["t"].view(np.uint32), osd["t"].view(np.uint32))
        core.render_frames(cams, w, h, shard=s, max_bounces=1, out=packed[r], rgb8=True, frame_stride=per)
    images = torch.zeros((len(cams), h * w, 4), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    core.assemble_frames(packed, True, w, h, tw, th, len(counts), owner, len(cams), len(cams) * per, per, images)
    core.sync()
    for j, cam in enumerate(cams):
        want, _ = ob.render(scene, ocam(ob, cam), w, h, params=oparams(ob, core.params), max_bounces=1)
        assert np.array_equal(images[j].cpu().numpy(), want), f"frame {j}"
    # a rank the table does not name renders nothing; a table that names a rank beyond n_ranks, or a frame stride
    # smaller than a frame, is refused
    assert rt.shard_pixels(w, h, rt.RtShard(tw, th, 7, owner=owner)) == 0
    core.render_frames(cams, w, h, shard=rt.RtShard(tw, th, 7, owner=owner), out=packed[0], rgb8=True)
    with pytest.raises(rt.RtError):
        core.assemble_frames(packed, True, w, h, tw, th, 2, owner, len(cams), len(cams) * per, per, images)
    with pytest.raises(rt.RtError):
        core.render_frames(cams, w, h, shard=shards[1], out=packed[1], rgb8=True, frame_stride=per // 4)
    # frames a stride apart that is wider than a frame: the last frame must still end inside the rank's block
    with pytest.raises(rt.RtError) as e:
        core.assemble_frames(packed, True, w, h, tw, th, len(counts), owner, len(cams), len(cams) * per, per + per // 2, images)
    assert e.value.code == rt.RT_ERR_INVALID


def test_sky_tiles_stay_at_home(core, ob, rt, car_xyz9):
    """rt_tile_flags + RT_OWNER_SKY + rt_assemble_frames_sky: tiles that show nothing but sky are neither rendered nor
    gathered; the assemble kernel writes them from the camera, byte for byte what rendering them would have given.
    Three ranks played in turn on this GPU, both wire formats, both assemble kernels (tile widths that split into 16-byte
    pieces and one that does not), cameras far away (mostly sky), at the start pose and inside the grid (no sky)."""
    import torch
    from ray_tracer_amd import sharding
    tris = rt.triangles_from_xyz9(car_xyz9)
    core.set_scene(None, tris)
    core.set_materials(None)
    core.build_grid()
    scene = ob.Scene(None, ob.triangles_from_xyz9(car_xyz9))
    world = 3
    poses = [rt.Camera((900, 500, 1200), 3.7, -0.3), rt.Camera.for_scene("gltf"), rt.Camera((10, 30, 20), 0.7, -0.2)]
    for (w, h, tw, th), rgb8 in (((320, 184, 32, 32), True), ((320, 184, 32, 16), False), ((203, 117, 24, 8), True)):
        cams = [p.to_uniform(0.0, float(h)) for p in poses]
        flags = [core.tile_flags(c, w, h, tw, th) for c in cams]
        n_tiles = sharding.n_tiles(w, h, tw, th)
        assert all(len(f) == n_tiles for f in flags)
        assert 0 < flags[0].sum() < n_tiles // 2 and flags[2].all(), (flags[0].sum(), flags[2].sum(), n_tiles)
        # a tile flagged sky holds no primary hit (the flags may keep more tiles than that, never fewer)
        for cam, f in zip(cams, flags):
            op, _, _ = ob.trace_primary(scene, ocam(ob, cam), w, h, shadow=False, params=oparams(ob, core.params), n_threads=8)
            hit = (op["id"] != ob.MISS).reshape(h, w)
            tiles_x = -(-w // tw)
            for t in np.flatnonzero(f == 0):
                ty, tx = divmod(int(t), tiles_x)
                assert not hit[ty * th:(ty + 1) * th, tx * tw:(tx + 1) * tw].any(), f"tile {t} flagged sky but hit"
        # one group of three frames: a tile stays at home only if it is sky in all three (here: none, the last camera is
        # inside); then groups of the first two cameras, where many do
        for group in (cams, cams[:2], cams[:1]):
            kept = np.zeros(n_tiles, bool)
            for cam in group:
                kept |= flags[cams.index(cam)] != 0
            owner = np.full(n_tiles, rt.RT_OWNER_SKY, np.uint32)
            owner[kept] = np.resize(np.asarray(sharding.owner_table([1, 2, 2]), np.uint32), int(kept.sum()))
            owner = [int(k) for k in owner]
            shards = [rt.RtShard(tw, th, r, owner=owner) for r in range(world)]
            per = max(1, max(rt.shard_pixels(w, h, s) for s in shards))
            assert sum(rt.shard_pixels(w, h, s) for s in shards) == int(kept.sum()) * tw * th
            packed = torch.full((world, len(group), per, 3 if rgb8 else 4), 77, dtype=torch.uint8, device="cuda")
            for r, s in enumerate(shards):
                core.render_frames(group, w, h, shard=s, max_bounces=1, shadows=True, out=packed[r], rgb8=rgb8, frame_stride=per)
            images = torch.zeros((len(group), h * w, 4), dtype=torch.uint8, device="cuda")
            core.assemble_frames_sky(packed, rgb8, w, h, tw, th, world, owner, group, len(group) * per, per, images)
            core.sync()
            for j, cam in enumerate(group):
                want, _ = ob.render(scene, ocam(ob, cam), w, h, params=oparams(ob, core.params), max_bounces=1, shadows=True)
                got = images[j].cpu().numpy()
                bad = np.flatnonzero((got != want).any(axis=1))
                assert bad.size == 0, f"{w}x{h} tiles {tw}x{th} rgb8={rgb8}, group of {len(group)}, frame {j}: {bad.size} pixels differ, first {bad[:4]}"
    # the sharder end to end with one rank: announced cameras take the sparse route, an unannounced one the whole one
    w, h = 320, 184
    cams = [p.to_uniform(0.0, float(h)) for p in poses]
    sh = sharding.FrameSharder(core, w, h, 32, 32, rank=0, world=1, device="cuda", shadow=True, mode="rgba", max_bounces=1)
    sh.enable_groups(group=2, groups_in_flight=2, rgb8=True)
    sh.enable_sparse()
    sh.prepare_cameras(cams[:2])
    a = sh.trace_group(cams[:2])
    sparse_bytes = sh.last_wire_bytes
    b = sh.trace_group([cams[0], cams[2]])
    assert sparse_bytes < sh.last_wire_bytes == 2 * sh.per_rank * 3
    sh.synchronize()
    core.set_stream(None)
    for cam, pic in ((cams[0], a[0]), (cams[1], a[1]), (cams[0], b[0]), (cams[2], b[1])):
        want, _ = ob.render(scene, ocam(ob, cam), w, h, params=oparams(ob, core.params), max_bounces=1, shadows=True)
        assert np.array_equal(pic.cpu().numpy(), want)
    # scenes the mask does not cover (boxes) flag every tile; sky entries without cameras are refused
    core.set_scene(demo_boxes(ob), None)
    core.build_grid()
    assert core.tile_flags(cams[0], w, h, 32, 32).all()
    with pytest.raises(rt.RtError):
        core.assemble_frames(torch.zeros((1, 1, 64, 4), dtype=torch.uint8, device="cuda"), False, 8, 8, 8, 8, 1, [rt.RT_OWNER_SKY], 1, 64, 64,
                             torch.zeros((1, 64, 4), dtype=torch.uint8, device="cuda"))
