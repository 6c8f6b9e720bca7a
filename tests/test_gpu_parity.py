"""Parity of the HIP path with the CPU oracle, through the C ABI (ctypes -> librt_mi355.so).

Bar: bit-exact. Object ids must be equal and hit distances must have identical f32 bit patterns
(the kernels evaluate the reference's expression trees with contraction off and correctly rounded
divide/sqrt), so no tolerance is written anywhere in this file. All tests need a real MI355X."""
import ctypes as C
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def assert_hits_equal(got, want, what=""):
    bad = np.flatnonzero(got["id"] != want["id"])
    assert bad.size == 0, f"{what}: {bad.size} ids differ, first at {bad[:5]}: {got[bad[:5]]} vs {want[bad[:5]]}"
    bad = np.flatnonzero(bits(got["t"]) != bits(want["t"]))
    assert bad.size == 0, f"{what}: {bad.size} t values differ, first at {bad[:5]}: {got[bad[:5]]} vs {want[bad[:5]]}"


def oparams(ob, p):
    return ob.Params.default(**{k: getattr(p, k) for k in ("finest_cell_size", "max_objects_per_cell", "max_dda_steps",
                                                           "wgsl_compat", "lod_cull", "moving_box_count", "fov",
                                                           "shadow_offset")})


def ocam(ob, cam):
    o = ob.CameraUniform()
    C.memmove(C.byref(o), C.byref(cam), C.sizeof(o))
    return o


def setup_scene(core, ob, boxes, tris, **grid_kw):
    """Uploads the scene, bins it on the GPU, checks the grid against the oracle's and returns the oracle scene."""
    core.set_scene(boxes, tris)
    meta, n_cells, n_refs = core.build_grid(cell_size=grid_kw.get("cell_size", 0.0),
                                            max_per_cell=grid_kw.get("max_per_cell", 0), mode=grid_kw.get("mode", 0))
    scene = ob.Scene(boxes, tris, **grid_kw)
    assert bytes(meta)[:32] == bytes(scene.grid.meta)[:32], "bounds / cell size differ"
    assert meta.fine_dims == scene.grid.meta.fine_dims
    cs, idx = core.read_grid_csr()
    assert n_cells == scene.grid.n_cells and n_refs == scene.grid.n_refs
    assert np.array_equal(cs, scene.grid.cell_start) and np.array_equal(idx, scene.grid.indices)
    return scene


def random_boxes(ob, n, seed, extent=40.0, moving_last=0):
    rng = np.random.default_rng(seed)
    boxes = np.zeros(n, ob.BOX_DTYPE)
    for i in range(n):
        c = rng.uniform(-extent, extent, 3).astype(np.float32)
        h = rng.uniform(0.5, 6.0, 3).astype(np.float32)
        if i >= n - moving_last:  # BoxData::create_moving_box, src/types.rs:123-151
            c1 = (c + rng.uniform(-10, 10, 3)).astype(np.float32)
            lo = np.minimum(c - h, c1 - h) - np.float32(0.5)
            hi = np.maximum(c + h, c1 + h) + np.float32(0.5)
            boxes[i] = ob.make_box(lo, hi, center0=c, center1=c1, half=h)
        else:
            boxes[i] = ob.make_box(c - h, c + h)
        boxes[i]["reflectivity"] = 0.0
    return boxes


# ------------------------------------------------------------------------------ primitive routines

def special_rays():
    o = [(0, 0, 0), (0, 0, 0), (0, 0, 0), (1, 2, 3), (0.5, 0.25, -7), (0, 0, 0), (0, 0, 0), (3, 3, 3)]
    d = [(0, 0, -1), (1, 0, 0), (0, -0.0, 1), (-0.0, 0, -1), (0, 1e-9, -1), (1e-9, -1e-9, -1), (0.6, 0, -0.8),
         (-0.57735026, -0.57735026, -0.57735026)]
    return np.array(o, np.float32), np.array(d, np.float32)


@pytest.mark.parametrize("wgsl", [0, 1])
@pytest.mark.parametrize("count", [100, 1000, 10000])
def test_aabb_pairs_criterion_particle_scene(rt, ob, count, wgsl):
    """benches/bounding_volumes.rs:147-179: 100 rays x {100, 1000, 10000} boxes; plus axis-aligned and
    signed-zero directions for the safe-inverse branch (src/math/ray.rs:8-12 / wgsl:162-166)."""
    core = rt.TraceCore(0, rt.default_params(wgsl_compat=wgsl))
    mn, mx = ob.particle_aabbs(count, 0.5)
    d = np.stack([ob.random_unit_vector(i) for i in range(100)])
    o = np.zeros_like(d)
    so, sd = special_rays()
    o, d = np.concatenate([o, so]), np.concatenate([d, sd])
    got = core.intersect_aabb_pairs(o, d, mn, mx)
    want = ob.intersect_aabb_pairs(o, d, mn, mx, wgsl)
    assert np.array_equal(bits(got), bits(want))
    assert (want[:100] > 0).sum() == (got[:100] > 0).sum() > 0  # the benchmark's hit_count
    core.close()


@pytest.mark.parametrize("n_rays,n_prims", [(1, 1), (7, 3), (3, 1023), (300, 1025), (5, 4099), (20000, 64), (70001, 5), (33, 4096)])
def test_aabb_pairs_ragged_sizes_host_and_device(rt, ob, n_rays, n_prims):
    """Every (ray, box) pair for sizes that do not fill the kernel's units (four boxes a thread, 1024 a workgroup, rays
    walked in steps of the grid's height), from host arrays and from device arrays, with the output on a 16-byte
    boundary and off it: bit-equal to the CPU routine (src/math/ray.rs:3-36) each time, nothing written past the end."""
    import torch
    rng = np.random.default_rng(n_rays * 131 + n_prims)
    o = rng.uniform(-30, 30, (n_rays, 3)).astype(np.float32)
    d = rng.normal(size=(n_rays, 3)).astype(np.float32)
    d[::5, 1] = 0.0  # safe-inverse branch
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    c = rng.uniform(-20, 20, (n_prims, 3)).astype(np.float32)
    h = rng.uniform(0.1, 6, (n_prims, 3)).astype(np.float32)
    mn, mx = c - h, c + h
    want = ob.intersect_aabb_pairs(o, d, mn, mx, 0)
    core = rt.TraceCore(0)
    assert np.array_equal(bits(core.intersect_aabb_pairs(o, d, mn, mx)), bits(want))
    to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    do, dd, dmn, dmx = to(o), to(d), to(mn), to(mx)
    for shift in (0, 1):  # output rows on 16-byte boundaries (when n_prims allows) and off them
        buf = torch.full((n_rays * n_prims + 8,), -7.0, dtype=torch.float32, device="cuda")
        out = buf[shift:shift + n_rays * n_prims].view(n_rays, n_prims)
        core.intersect_aabb_pairs(do, dd, dmn, dmx, out=out)
        core.sync()
        assert np.array_equal(bits(out.cpu().numpy()), bits(want)), shift
        assert bool((buf[:shift] == -7.0).all()) and bool((buf[shift + n_rays * n_prims:] == -7.0).all())
    core.close()


@pytest.mark.parametrize("any_hit", [0, 1])
def test_ray_lists_one_lane_and_four_lanes_to_a_ray(rt, ob, car_xyz9, any_hit):
    """rt_trace_rays walks short lists four lanes to a ray (a cell's list dealt to the four, results merged under (t, list
    position)) and long ones a lane to a ray: the same records either way, and the oracle's, closest hit and any hit."""
    import os
    tris = ob.triangles_from_xyz9(car_xyz9)
    cam = ob.camera_to_uniform((200, 200, 300), 3.35, -0.28, 0.0, 135.0)
    o, d = ob.primary_rays(cam, 240, 135)
    rng = np.random.default_rng(5)
    o2 = rng.uniform(-300, 300, (5000, 3)).astype(np.float32)  # rays in no order, many from inside the grid
    d2 = rng.normal(size=(5000, 3)).astype(np.float32)
    d2 /= np.linalg.norm(d2, axis=1, keepdims=True).astype(np.float32)
    o, d = np.concatenate([o, o2]), np.concatenate([d, d2])
    scene = ob.Scene(None, tris)
    want, _ = ob.trace_rays(scene, o, d, any_hit=bool(any_hit))
    core = rt.TraceCore(0)
    core.set_scene(None, tris)
    core.build_grid()
    try:
        for lanes in ("1", "4"):
            os.environ["RT_RAY_LANES"] = lanes
            got = core.trace_rays(o, d, any_hit=bool(any_hit))
            assert_hits_equal(got, want, f"{lanes} lane(s) to a ray")
            for n in (1, 3, 63, 65):  # lists that end inside a wave, inside a ray's four lanes' wave
                assert_hits_equal(core.trace_rays(o[:n], d[:n], any_hit=bool(any_hit)), want[:n], f"{lanes} lanes, {n} rays")
    finally:
        os.environ.pop("RT_RAY_LANES", None)
    core.close()


@pytest.mark.parametrize("wgsl", [0, 1])
def test_tri_pairs(rt, ob, car_xyz9, wgsl):
    core = rt.TraceCore(0, rt.default_params(wgsl_compat=wgsl))
    o, d = ob.generate_test_rays(200)
    tri = ob.generate_test_triangles(2000)
    v = np.concatenate([tri["v0"], tri["v1"], tri["v2"]], axis=1)
    got, want = core.intersect_tri_pairs(o, d, v), ob.intersect_tri_pairs(o, d, v, wgsl)
    assert np.array_equal(bits(got), bits(want)) and (want > 0).sum() > 10
    cam = ob.camera_to_uniform((200, 200, 300), 3.35, -0.28, 0.0, 16.0)
    o, d = ob.primary_rays(cam, 24, 16)
    got, want = core.intersect_tri_pairs(o, d, car_xyz9), ob.intersect_tri_pairs(o, d, car_xyz9, wgsl)
    assert np.array_equal(bits(got), bits(want)) and (want > 0).sum() > 100
    core.close()


def test_watertight_pairs(core, ob, car_xyz9):
    """watertight_intersect (src/core/triangle_intersection.rs:101-181) on the GPU: bit-identical to the oracle's
    restatement, close to Moller-Trumbore (the reference's own test asks for |dt| < 0.1, :286-301), and the
    reference's known answer t = 5."""
    o, d = special_rays()
    cam = ob.camera_to_uniform((200, 200, 300), 3.35, -0.28, 0.0, 16.0)
    po, pd = ob.primary_rays(cam, 24, 16)
    o, d = np.concatenate([o, po]), np.concatenate([d, pd])
    got, want = core.intersect_tri_pairs(o, d, car_xyz9, watertight=True), ob.watertight_pairs(o, d, car_xyz9)
    assert np.array_equal(bits(got), bits(want)) and (want > 0).sum() > 100
    mt = core.intersect_tri_pairs(o, d, car_xyz9)
    both = (got > 0) & (mt > 0)
    assert both.sum() > 100 and np.abs(got[both] - mt[both]).max() < 0.1
    tri = np.array([[-1, 0, -5, 1, 0, -5, 0, 1, -5]], np.float32)
    t = core.intersect_tri_pairs(np.zeros((1, 3), np.float32), np.array([[0, 0.1, -1]], np.float32) / np.float32(np.hypot(0.1, 1)), tri, watertight=True)
    assert abs(float(t[0, 0]) - 5.0 * np.hypot(0.1, 1)) < 1e-4


def test_reference_known_answers_on_gpu(core):
    """src/math/ray.rs:43-71 and src/core/triangle_intersection.rs:219-252,357-364 evaluated by the kernels."""
    t = core.intersect_aabb_pairs([(0, 0, 0), (0, 0, 0), (5, 0, 0)], [(1, 0, 0)] * 3,
                                  [(5, -1, -1), (5, 2, 2), (0, -1, -1)], [(10, 1, 1), (10, 3, 3), (10, 1, 1)])
    assert t[0, 0] == 5.0 and t[1, 1] == -1.0 and t[2, 2] == 5.0
    tri = [(-1, 0, -5, 1, 0, -5, 0, 1, -5)]
    d_miss = np.array([5, 0, -1], np.float32) * np.float32(1 / math.sqrt(26))
    t = core.intersect_tri_pairs([(0, 0, 0), (0, 0, 0), (0, 0, 0), (0, 0, -5)], [(0, 0, -1), d_miss, (0, 0, 1), (1, 0, 0)], tri)
    assert t[0, 0] == 5.0 and t[1, 0] == -1.0 and t[2, 0] == -1.0 and t[3, 0] == -1.0


# ------------------------------------------------------------------------------ grid binning

def test_grid_build_models(core, ob, car_xyz9, phoenix_xyz9, golden):
    for name, xyz in (("car", car_xyz9), ("phoenix", phoenix_xyz9)):
        setup_scene(core, ob, None, ob.triangles_from_xyz9(xyz))
        cs, idx = core.read_grid_csr()
        assert np.array_equal(cs, golden[f"{name}_cell_start"]) and np.array_equal(idx, golden[f"{name}_indices"])


@pytest.mark.parametrize("cell_size", [16.0, 5.0, 1.0])
def test_grid_build_pyramid(core, ob, cell_size):
    setup_scene(core, ob, None, ob.pyramid_triangles(), cell_size=cell_size)


def test_grid_build_boxes_and_triangles(core, ob):
    setup_scene(core, ob, random_boxes(ob, 300, 1, moving_last=3), ob.generate_test_triangles(5000), cell_size=8.0)
    setup_scene(core, ob, random_boxes(ob, 17, 2), None)


def test_grid_build_cap_drops_highest_ids(core, ob):
    """src/grid.rs:92-100 with a small cap: the lowest ids of every over-full cell survive."""
    tris = ob.generate_test_triangles(20000)
    scene = setup_scene(core, ob, None, tris, max_per_cell=16)
    assert scene.grid.dropped > 0
    setup_scene(core, ob, None, tris[:9000], max_per_cell=8192)  # reference cap, nothing dropped


def test_grid_build_triangle_grid_mode(core, ob, car_xyz9):
    """src/grid_triangles.rs:29-213: cap 256, dims clamped to [1,1024], i32 cell coordinates."""
    tris = ob.triangles_from_xyz9(car_xyz9)
    setup_scene(core, ob, None, tris, mode=ob.GRID_TRIANGLES)
    setup_scene(core, ob, None, ob.generate_test_triangles(40000), mode=ob.GRID_TRIANGLES)  # cap 256 bites
    setup_scene(core, ob, None, tris[:50], cell_size=0.3, mode=ob.GRID_TRIANGLES, max_per_cell=8)  # dims hit 1024


def test_grid_build_empty_scene(core, ob):
    setup_scene(core, ob, None, None)
    setup_scene(core, ob, None, None, mode=ob.GRID_TRIANGLES)
    cam = ob.camera_to_uniform((0, 0, 10), math.pi, 0.0, 0.0, 8.0)
    p, s = core.trace_primary(cam, 16, 8)
    assert np.all(p["id"] == ob.MISS) and np.all(s["id"] == ob.MISS)


def test_grid_rebuild_after_vertex_update(core, ob, phoenix_xyz9):
    """The animated path: new vertices every frame, rebuild on the GPU, same answer as a fresh CPU build."""
    core.set_scene(None, ob.triangles_from_xyz9(phoenix_xyz9))
    core.build_grid()
    centre = phoenix_xyz9.reshape(-1, 3).mean(axis=0).astype(np.float32)
    for frame in (7, 45):
        a = np.float32(2 * math.pi * frame / 120)
        ca, sa = np.float32(math.cos(a)), np.float32(math.sin(a))
        v = phoenix_xyz9.reshape(-1, 3) - centre
        moved = np.stack([v[:, 0] * ca + v[:, 2] * sa, v[:, 1], -v[:, 0] * sa + v[:, 2] * ca], axis=1) + centre
        moved[:, 1] += np.float32(10 * (math.sin(2 * frame / 60) + 1) / 2)
        moved = moved.astype(np.float32).reshape(-1, 9)
        core.update_vertices(moved)
        with pytest.raises(Exception):
            core.trace_primary(ob.camera_to_uniform((0, 0, 0), 0, 0), 8, 8)  # stale grid must be refused
        meta, nc, nr = core.build_grid()
        scene = ob.Scene(None, ob.triangles_from_xyz9(moved))
        cs, idx = core.read_grid_csr()
        assert meta.fine_dims == scene.grid.meta.fine_dims and bytes(meta)[:32] == bytes(scene.grid.meta)[:32]
        assert np.array_equal(cs, scene.grid.cell_start) and np.array_equal(idx, scene.grid.indices)
        cam = ob.camera_to_uniform((-300, 250, 600), 2.7, -0.2, 0.0, 90.0)
        p, s = core.trace_primary(cam, 160, 90)
        op, os_, _ = ob.trace_primary(scene, cam, 160, 90)
        assert_hits_equal(p, op, "animated primary")
        assert_hits_equal(s, os_, "animated shadow")
        assert (op["id"] != ob.MISS).sum() > 200


# ------------------------------------------------------------------------------ primary + shadow trace

def check_frame(core, ob, scene, cam, w, h, shard=None, what=""):
    p, s = core.trace_primary(cam, w, h, shard=shard)
    oshard = ob.Shard(shard.tile_w, shard.tile_h, shard.first, shard.stride, shard.owner_list()) if shard is not None else None
    op, os_, st = ob.trace_primary(scene, ocam(ob, cam), w, h, params=oparams(ob, core.params), shard=oshard)
    assert_hits_equal(p, op, what + " primary")
    assert_hits_equal(s, os_, what + " shadow")
    return op, os_, st


def test_trace_matches_committed_golden(core, ob, rt, golden, car_xyz9):
    cam = rt.Camera.for_scene("pyramid").to_uniform(0.0, 36.0)
    for name, cell in (("pyramid", 16.0), ("pyramid_c1", 1.0)):
        core.set_scene(None, ob.pyramid_triangles())
        core.build_grid(cell_size=cell)
        p, s = core.trace_primary(cam, 64, 36)
        assert_hits_equal(p, golden[f"{name}_primary"], name)
        assert_hits_equal(s, golden[f"{name}_shadow"], name)
    core.set_scene(None, rt.triangles_from_xyz9(car_xyz9))
    core.build_grid()
    p, s = core.trace_primary(rt.Camera.for_scene("gltf").to_uniform(0.0, 36.0), 64, 36)
    assert_hits_equal(p, golden["car_primary"], "car")
    assert_hits_equal(s, golden["car_shadow"], "car")


@pytest.mark.parametrize("w,h", [(64, 36), (333, 187), (1920, 1080)])
def test_trace_car(core, ob, rt, car_xyz9, w, h):
    """BASELINE config 2: the bundled static model from the `SCENE=gltf` start pose (src/camera.rs:63)."""
    scene = setup_scene(core, ob, None, ob.triangles_from_xyz9(car_xyz9))
    cam = rt.Camera.for_scene("gltf").to_uniform(0.0, float(h))
    op, os_, st = check_frame(core, ob, scene, cam, w, h, what=f"car {w}x{h}")
    assert st["primary_hits"] > 0.3 * w * h and st["shadow_hits"] > 0


def test_trace_car_other_views(core, ob, rt, car_xyz9):
    scene = setup_scene(core, ob, None, ob.triangles_from_xyz9(car_xyz9))
    views = [((0, 100, 0), 0.3, -1.2), ((-400, 50, -500), 0.7, 0.0), ((0, 150, 50), 1.0, -0.4),  # inside the grid
             ((2000, 900, 2500), 3.8, -0.3), ((0, -200, 0), 0.0, 1.3), ((0, 50, 3000), 0.0, 0.0)]  # last one looks away
    for pos, yaw, pitch in views:
        cam = rt.Camera(pos, yaw, pitch).to_uniform(0.0, 120.0)
        check_frame(core, ob, scene, cam, 200, 120, what=f"car from {pos}")


def test_trace_finer_grids(core, ob, rt, car_xyz9):
    """Tuned cell sizes (not the reference's 16.0): long traversals, the 200-step cap bites at 2.0."""
    for cell in (8.0, 2.0):
        scene = setup_scene(core, ob, None, ob.triangles_from_xyz9(car_xyz9), cell_size=cell)
        cam = rt.Camera.for_scene("gltf").to_uniform(0.0, 90.0)
        check_frame(core, ob, scene, cam, 160, 90, what=f"car cell {cell}")


@pytest.mark.parametrize("build_cap,trace_cap", [(8, 8192), (12, 5), (8192, 6)])
def test_trace_truncated_cell_lists(rt, ob, car_xyz9, build_cap, trace_cap):
    """Cell lists cut short at build time (overflow drops the highest ids, src/grid.rs:262-283) and / or at trace
    time (max_objects_per_cell). A triangle dropped from one cell but kept in its neighbour must still be tested
    in the neighbour: the kernel's skipping of repeated tests may not assume the earlier cell held it."""
    core = rt.TraceCore(0, rt.default_params(max_objects_per_cell=trace_cap))
    tris = ob.triangles_from_xyz9(car_xyz9)
    core.set_scene(None, tris)
    core.build_grid(max_per_cell=build_cap)
    scene = ob.Scene(None, tris, max_per_cell=build_cap)
    cs, idx = core.read_grid_csr()
    assert np.array_equal(cs, scene.grid.cell_start) and np.array_equal(idx, scene.grid.indices)
    assert np.diff(cs).max() == min(build_cap, 76)
    for pos, yaw, pitch in (((200, 200, 300), 3.35, -0.28), ((0, 150, 50), 1.0, -0.4)):
        cam = rt.Camera(pos, yaw, pitch).to_uniform(0.0, 135.0)
        check_frame(core, ob, scene, cam, 240, 135, what=f"caps {build_cap}/{trace_cap} from {pos}")
    core.close()


def test_trace_phoenix(core, ob, rt, phoenix_xyz9):
    scene = setup_scene(core, ob, None, ob.triangles_from_xyz9(phoenix_xyz9))
    cam = rt.Camera((-300, 250, 600), 2.7, -0.2).to_uniform(0.0, 180.0)
    _, _, st = check_frame(core, ob, scene, cam, 320, 180, what="phoenix")
    assert st["primary_hits"] > 1000


def test_trace_massive_benchmark_scene_ties(core, ob, rt):
    """BASELINE config 3's scene (src/core/benchmark.rs:99-123): positions repeat with period 10000, so with
    30000 triangles every hit is an exact three-way tie; list order must pick the same id as the oracle."""
    tris = ob.generate_test_triangles(30000)
    scene = setup_scene(core, ob, None, tris)
    cam = rt.Camera((0, 0, 150), math.pi, 0.0).to_uniform(0.0, 54.0)
    op, _, st = check_frame(core, ob, scene, cam, 96, 54, what="massive 30k")
    hit = op["id"] != ob.MISS
    assert hit.sum() > 100 and np.all(op["id"][hit] < 10000)  # the first copy always wins the tie


def test_trace_boxes_moving_and_lod(core, ob, rt):
    """Boxes only (the reference's procedural scenes), last three moving (wgsl:442-461), LOD cull on (wgsl:147-155),
    several times so the interpolated centres move."""
    boxes = random_boxes(ob, 120, 3, extent=60.0, moving_last=3)
    scene = setup_scene(core, ob, boxes, None)
    for time in (0.0, 0.37, 2.9):
        cam = rt.Camera((0, 20, 170), math.pi, -0.1).to_uniform(time, 108.0)
        op, _, st = check_frame(core, ob, scene, cam, 192, 108, what=f"boxes t={time}")
        assert st["primary_hits"] > 500 and st["box_tests"] > 0
    far = rt.Camera((0, 20, 400), math.pi, -0.05).to_uniform(0.0, 108.0)  # beyond 200 units: everything culled
    op, _, st = check_frame(core, ob, scene, far, 192, 108, what="boxes far")
    assert st["primary_hits"] < 200


def test_trace_mixed_scene(core, ob, rt):
    boxes = random_boxes(ob, 60, 5, extent=45.0, moving_last=3)
    tris = ob.generate_test_triangles(4000)
    scene = setup_scene(core, ob, boxes, tris, cell_size=8.0)
    cam = rt.Camera((10, 30, 140), math.pi, -0.2).to_uniform(1.1, 90.0)
    op, _, st = check_frame(core, ob, scene, cam, 160, 90, what="mixed")
    hit = op["id"][op["id"] != ob.MISS]
    assert (hit < 60).sum() > 50 and (hit >= 60).sum() > 50


@pytest.mark.parametrize("over", [dict(wgsl_compat=1), dict(lod_cull=0), dict(moving_box_count=0), dict(max_dda_steps=7),
                                  dict(shadow_offset=0.05), dict(max_objects_per_cell=3)])
def test_trace_parameter_variants(rt, ob, over):
    core = rt.TraceCore(0, rt.default_params(**over))
    boxes = random_boxes(ob, 40, 11, extent=45.0, moving_last=3)
    tris = ob.generate_test_triangles(3000)
    core.set_scene(boxes, tris)
    core.build_grid(cell_size=8.0, max_per_cell=8192)
    scene = ob.Scene(boxes, tris, cell_size=8.0)
    cam = rt.Camera((10, 30, 140), math.pi, -0.2).to_uniform(0.6, 90.0)
    check_frame(core, ob, scene, cam, 160, 90, what=str(over))
    core.close()


def test_grid_with_more_than_2_to_25_cells(core, ob, rt):
    """Sizes at which library routines have been seen to stop working without an error: the per-cell sort of the
    binning pass runs over the cells in batches from 2^24 cells on."""
    rng = np.random.default_rng(5)
    c = np.repeat(rng.uniform(-680, 680, (300, 1, 3)), 20, axis=0) + rng.normal(0, 3.0, (6000, 1, 3))  # 300 clusters
    v = (c + rng.normal(0, 6.0, (6000, 3, 3))).astype(np.float32)
    tris = ob.triangles_from_xyz9(v.reshape(-1, 9))
    scene = setup_scene(core, ob, None, tris, cell_size=4.0, max_per_cell=3)
    assert scene.grid.n_cells > 2 ** 25 + 2 ** 22
    p = rt.default_params(max_objects_per_cell=3, max_dda_steps=1000)
    core2 = rt.TraceCore(0, p)
    setup_scene(core2, ob, None, tris, cell_size=4.0, max_per_cell=3)
    w, h = 128, 72
    cam = rt.Camera((0.0, 300.0, 2100.0), math.pi, -0.15).to_uniform(0.0, float(h))
    gp, gs = core2.trace_primary(cam, w, h)
    op, os_, _ = ob.trace_primary(scene, ocam(ob, cam), w, h, params=oparams(ob, p))
    assert_hits_equal(gp, op, "primary")
    assert_hits_equal(gs, os_, "shadow")
    assert (gp["id"] != rt.RT_MISS).sum() > 20
    core2.close()


def test_reference_records_beyond_4_gib(core, ob, rt):
    """More than 2^32 / 48 references: the trace kernels switch from buffer descriptors (32-bit byte offsets) to
    plain pointers."""
    rng = np.random.default_rng(6)
    c = rng.uniform(-60, 60, (260, 1, 3))
    v = (c + rng.normal(0, 45.0, (260, 3, 3))).astype(np.float32)
    tris = ob.triangles_from_xyz9(v.reshape(-1, 9))
    scene = setup_scene(core, ob, None, tris, cell_size=1.0)
    assert (scene.grid.n_refs + 1) * 48 > 2 ** 32
    w, h = 96, 54
    cam = rt.Camera((30.0, 80.0, 420.0), math.pi, -0.15).to_uniform(0.0, float(h))
    for stats in (False, True):
        core.set_collect_stats(stats)
        gp, gs = core.trace_primary(cam, w, h)
        op, os_, st = ob.trace_primary(scene, ocam(ob, cam), w, h)
        assert_hits_equal(gp, op, "primary")
        assert_hits_equal(gs, os_, "shadow")
        if stats:
            got = core.stats()
            assert {k: got[k] for k in st} == st
    core.set_collect_stats(False)
    o, d = ob.primary_rays(ocam(ob, cam), w, h)
    assert_hits_equal(core.trace_rays(o, d, any_hit=True), ob.trace_rays(scene, o, d, any_hit=True)[0], "ray list")
    img = core.render(cam, w, h, max_bounces=2)
    assert np.array_equal(img, ob.render(scene, ocam(ob, cam), w, h, max_bounces=2)[0])


def test_duplicate_triangles_are_left_out_exactly(ob, rt):
    """The reference's benchmark generator repeats its 10 000 positions: copies of a triangle with a lower id are
    left out of the records the trace walks. Same ids, same t bits as the oracle (which tests every entry), with
    originals that are invisible or alpha-textured (their copies must stay), through closest-hit, any-hit, the
    picture and the counting kernels (which walk the full lists)."""
    from test_oracle_render import alpha_materials
    tris = ob.generate_test_triangles(35000)  # 10 000 positions, 3 to 4 copies each
    tris["material_id"] = (np.arange(len(tris)) * 7 // 3) % 6
    mats = alpha_materials(ob)
    w, h = 160, 90
    cam = rt.Camera((20.0, 30.0, 160.0), math.pi + 0.1, -0.15).to_uniform(0.0, float(h))
    o, d = ob.generate_test_rays(4000)
    results = {}
    for skip, use_mats, cell in ((1, True, 16.0), (0, True, 16.0), (1, False, 4.0), (1, True, 4.0)):
        core = rt.TraceCore(0, rt.default_params(skip_duplicate_triangles=skip))
        m = mats if use_mats else None
        if use_mats:
            core.set_materials(m)  # before the grid
        scene = setup_scene(core, ob, None, tris, cell_size=cell)  # also checks rt_read_grid_csr against the full lists
        _, _, n_refs = core.build_grid(cell_size=cell)
        scene = ob.Scene(None, tris, materials=m, cell_size=cell)
        n_trace = core.trace_refs()
        if skip:
            assert n_trace < (0.8 if use_mats else 0.32) * n_refs, (n_trace, n_refs)
            if use_mats:  # copies of originals with an alpha rule stay: more records than without materials
                core.set_materials(None)
                assert core.trace_refs() < n_trace
                core.set_materials(m)  # after the grid
                assert core.trace_refs() == n_trace
        else:
            assert n_trace == n_refs
        gp, gs = core.trace_primary(cam, w, h)
        op, os_, st = ob.trace_primary(scene, ocam(ob, cam), w, h)
        assert_hits_equal(gp, op, f"primary skip={skip} mats={use_mats} cell={cell}")
        assert_hits_equal(gs, os_, f"shadow skip={skip} mats={use_mats} cell={cell}")
        for any_hit in (False, True):
            assert_hits_equal(core.trace_rays(o, d, any_hit=any_hit), ob.trace_rays(scene, o, d, any_hit=any_hit)[0], "ray list")
        img = core.render(cam, w, h, max_bounces=2, shadows=True)
        assert np.array_equal(img, ob.render(scene, ocam(ob, cam), w, h, max_bounces=2, shadows=True)[0])
        core.set_collect_stats(True)  # the counting kernels walk the reference's lists
        assert core.trace_refs() == n_refs
        gp2, gs2 = core.trace_primary(cam, w, h)
        got = core.stats()
        assert {k: got[k] for k in st} == st
        assert_hits_equal(gp2, op, "primary, counting kernel")
        core.set_collect_stats(False)
        assert core.trace_refs() == n_trace
        assert_hits_equal(core.trace_primary(cam, w, h)[0], op, "primary, after counting")
        results[(skip, use_mats, cell)] = gp
        core.close()
    assert_hits_equal(results[(1, True, 16.0)], results[(0, True, 16.0)], "skip on vs off")
    # a shader-side cap below the build cap cuts lists after the fact: nothing is left out then
    core = rt.TraceCore(0, rt.default_params(max_objects_per_cell=7))
    core.set_scene(None, tris)
    _, _, n_refs = core.build_grid(cell_size=16.0, max_per_cell=8192)
    assert core.trace_refs() == n_refs
    p7 = rt.default_params(max_objects_per_cell=7)
    gp, gs = core.trace_primary(cam, w, h)
    op, os_, _ = ob.trace_primary(ob.Scene(None, tris, cell_size=16.0), ocam(ob, cam), w, h, params=oparams(ob, p7))
    assert_hits_equal(gp, op, "trace-time cap")
    core.close()


def test_stats_equal_oracle_counts(core, ob, rt, car_xyz9):
    scene = setup_scene(core, ob, None, ob.triangles_from_xyz9(car_xyz9))
    cam = rt.Camera.for_scene("gltf").to_uniform(0.0, 135.0)
    core.set_collect_stats(True)
    _, _, st = check_frame(core, ob, scene, cam, 240, 135, what="stats")
    got = core.stats()
    assert {k: got[k] for k in st} == st, (got, st)
    assert got["cell_records_fetched"] == got["ref_records_fetched"] == got["ref_records_staged"] == got["cell_records_uniform"] == 0
    # mode 2: the shipped kernels' own fetches (re-test skipping, look-ahead, staging): same answers, rays and hits
    # as counted above, no more candidate records asked for than twice the reference's tests plus one per visited cell
    core.set_collect_stats(2)
    check_frame(core, ob, scene, cam, 240, 135, what="stats, fetch counting")
    mov = core.stats()
    assert all(mov[k] == st[k] for k in ("rays", "primary_hits", "shadow_rays", "shadow_hits"))
    assert mov["cells_visited"] == mov["tri_tests"] == mov["box_tests"] == 0
    # the frame's block mask (rt_params.cull_empty_blocks) is followed by this count as by the timed kernels: pixel blocks
    # that cannot meet geometry fetch nothing
    assert 0 < mov["cell_records_fetched"] <= st["cells_visited"] + st["rays"]
    assert 0 < mov["cell_records_uniform"] <= mov["cell_records_fetched"]
    assert 0 < mov["ref_records_fetched"] + mov["ref_records_staged"] <= 2 * st["tri_tests"] + st["cells_visited"]
    core.set_collect_stats(False)
    plain = rt.TraceCore(0, rt.default_params(cull_empty_blocks=0, collect_stats=2))
    try:
        setup_scene(plain, ob, None, ob.triangles_from_xyz9(car_xyz9))
        check_frame(plain, ob, scene, cam, 240, 135, what="stats, fetch counting, no block mask")
        every = plain.stats()
        assert all(every[k] == st[k] for k in ("rays", "primary_hits", "shadow_rays", "shadow_hits"))
        assert st["cells_visited"] <= every["cell_records_fetched"] <= st["cells_visited"] + st["rays"]
        assert mov["cell_records_fetched"] < every["cell_records_fetched"]
        assert mov["ref_records_fetched"] + mov["ref_records_staged"] <= every["ref_records_fetched"] + every["ref_records_staged"]
    finally:
        plain.close()


def test_debug_pixel(core, ob, rt, car_xyz9):
    """raytracer_unified.wgsl:695-709 geometry fields."""
    scene = setup_scene(core, ob, None, ob.triangles_from_xyz9(car_xyz9))
    cam = rt.Camera.for_scene("gltf").to_uniform(0.0, 90.0)
    for px, py in ((80, 45), (0, 0), (120, 60), (159, 89)):
        got = core.debug_pixel(cam, 160, 90, px, py)
        want = ob.debug_pixel(scene, ocam(ob, cam), 160, 90, px, py)
        assert bytes(got) == bytes(want), (px, py)


# ------------------------------------------------------------------------------ grids supplied by the caller

def test_caller_supplied_grids(core, ob, rt):
    """rt_set_grid_dense (the reference's FineCellData[] upload) and rt_set_grid_csr give the same picture."""
    tris = ob.pyramid_triangles()
    boxes = random_boxes(ob, 5, 9, extent=6.0)
    grid = ob.Grid(boxes, tris, cell_size=4.0)
    scene = ob.Scene(boxes, tris, grid=grid)
    cam = rt.Camera.for_scene("pyramid").to_uniform(0.0, 48.0)
    want, wshadow, _ = ob.trace_primary(scene, ocam(ob, cam), 80, 48)
    core.set_scene(boxes, tris)
    core.set_grid_dense(grid.meta, grid.dense)
    p, s = core.trace_primary(cam, 80, 48)
    assert_hits_equal(p, want, "dense grid")
    assert_hits_equal(s, wshadow, "dense grid")
    core.set_scene(boxes, tris)
    with pytest.raises(rt.RtError):
        core.trace_primary(cam, 80, 48)  # new scene, no grid yet
    core.set_grid_csr(grid.meta, grid.cell_start, grid.indices)
    p, s = core.trace_primary(cam, 80, 48)
    assert_hits_equal(p, want, "csr grid")
    tg = ob.Grid(None, tris, cell_size=4.0, mode=ob.GRID_TRIANGLES)
    core.set_scene(None, tris)
    core.set_grid_dense(tg.meta, tg.dense, tri_cells=True)
    p, _ = core.trace_primary(cam, 80, 48)
    want, _, _ = ob.trace_primary(ob.Scene(None, tris, grid=tg), ocam(ob, cam), 80, 48)
    assert_hits_equal(p, want, "triangle-grid dense")


def test_out_of_range_ids_are_skipped(core, ob, rt):
    """A cell list naming a triangle past the array: skipped (raytracer_unified.wgsl:549)."""
    tris = ob.pyramid_triangles()
    grid = ob.Grid(None, tris)
    idx = np.concatenate([grid.indices, [17, 6]]).astype(np.uint32)
    cs = grid.cell_start.copy()
    cs[-1] += 2
    core.set_scene(None, tris)
    core.set_grid_csr(grid.meta, cs, idx)
    cam = rt.Camera.for_scene("pyramid").to_uniform(0.0, 36.0)
    p, _ = core.trace_primary(cam, 64, 36)
    want, _, _ = ob.trace_primary(ob.Scene(None, tris, grid=grid), ocam(ob, cam), 64, 36)
    assert_hits_equal(p, want)


# ------------------------------------------------------------------------------ ray lists

def test_trace_rays_reference_generators(core, ob, golden):
    """examples/massive_benchmark.rs:82-98: generate_test_rays against generate_test_triangles."""
    tris = ob.generate_test_triangles(10000)
    scene = setup_scene(core, ob, None, tris)
    o, d = ob.generate_test_rays(1000)
    got = core.trace_rays(o, d)
    assert_hits_equal(got, golden["bench10k_rays1000_hits"], "golden ray list")
    o, d = ob.generate_test_rays(50000)
    want, _ = ob.trace_rays(scene, o, d)
    assert_hits_equal(core.trace_rays(o, d), want, "50k rays closest")
    want, _ = ob.trace_rays(scene, o, d, any_hit=True)
    assert_hits_equal(core.trace_rays(o, d, any_hit=True), want, "50k rays any-hit")
    assert (want["id"] != ob.MISS).sum() > 1000


def test_trace_rays_degenerate_directions(core, ob, car_xyz9):
    """Axis-aligned, signed-zero and tiny components; origins inside, outside and on the bounds."""
    scene = setup_scene(core, ob, None, ob.triangles_from_xyz9(car_xyz9))
    bmin, bmax = np.array(scene.grid.meta.bounds_min), np.array(scene.grid.meta.bounds_max)
    o = [(0, 500, 0), (0, 100, 0), (0, 100, 0), (0, 100, 0), (-2000, 100, 0), (0, 100, 2000), tuple(bmax), tuple(bmin),
         (0, 100, 0), (0, 100, 0), (5000, 5000, 5000), (0, 100, 0)]
    d = [(0, -1, 0), (0, -1, -0.0), (-0.0, -1, 0), (1e-9, -1, -1e-9), (1, 0, 0), (0, 0, -1), (-0.57735026,) * 3,
         (0.57735026,) * 3, (0, 1, 0), (1, 0, -0.0), (0.57735026,) * 3, (0, 0, 0)]
    o, d = np.array(o, np.float32), np.array(d, np.float32)
    for any_hit in (False, True):
        want, _ = ob.trace_rays(scene, o, d, any_hit=any_hit)
        assert_hits_equal(core.trace_rays(o, d, any_hit=any_hit), want, f"degenerate any={any_hit}")


def test_trace_rays_from_the_far_bounds(core, ob):
    """Origins exactly on the far faces, edges and corner of a grid whose extent is a whole number of cells: the
    start cell index equals the grid dimension, the shader keeps going with it (its flattened index then names
    a cell of the next row or slab, raytracer_unified.wgsl:406-409) and only a step back inside, or out, ends that."""
    rng = np.random.default_rng(21)
    v = rng.uniform(2.0, 62.0, (600, 1, 3)) + rng.uniform(-6.0, 6.0, (600, 3, 3))
    tris = ob.triangles_from_xyz9(np.clip(v, 0.5, 63.5).reshape(-1, 9).astype(np.float32))
    grid = ob.Grid(None, tris, cell_size=16.0)
    assert grid.meta.fine_dims == (5, 5, 5) and tuple(grid.meta.bounds_min) == (-0.5, -0.5, -0.5)
    grid.meta.bounds_max[:] = (79.5, 79.5, 79.5)  # exactly five cells of 16 from -0.5
    scene = ob.Scene(None, tris, grid=grid)
    core.set_scene(None, tris)
    core.set_grid_csr(grid.meta, grid.cell_start, grid.indices)
    n = 3000
    o = rng.uniform(0.0, 79.0, (n, 3)).astype(np.float32)
    on = rng.integers(1, 8, n)  # bit k set: coordinate k sits on the far bound
    for k in range(3):
        o[(on >> k) & 1 == 1, k] = 79.5
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    slide = rng.random(n) < 0.3  # slide along a face: zero / signed-zero / tiny component on a far axis
    for k in range(3):
        m = slide & ((on >> k) & 1 == 1)
        d[m, k] = rng.choice(np.array([0.0, -0.0, 1e-9, -1e-9], np.float32), m.sum())
    for any_hit in (False, True):
        want, st = ob.trace_rays(scene, o, d, any_hit=any_hit)
        assert_hits_equal(core.trace_rays(o, d, any_hit=any_hit), want, f"far bounds any={any_hit}")
    assert (want["id"] != ob.MISS).sum() > 100


# ------------------------------------------------------------------------------ shards, device buffers, full size

def test_shards_match_oracle_and_assemble(core, ob, rt, car_xyz9):
    import torch
    scene = setup_scene(core, ob, None, ob.triangles_from_xyz9(car_xyz9))
    w, h, tw, th, n = 250, 141, 64, 64, 3
    cam = rt.Camera.for_scene("gltf").to_uniform(0.0, float(h))
    full, full_s, _ = check_frame(core, ob, scene, cam, w, h, what="full")
    per_rank = max(rt.shard_pixels(w, h, rt.RtShard(tw, th, r, n)) for r in range(n))
    gathered = torch.full((n, per_rank, 2), -1, dtype=torch.int32, device="cuda")
    gathered_s = torch.full((n, per_rank, 2), -1, dtype=torch.int32, device="cuda")
    for r in range(n):
        sh = rt.RtShard(tw, th, r, n)
        check_frame(core, ob, scene, cam, w, h, shard=sh, what=f"shard {r}")
        core.trace_primary(cam, w, h, shard=sh, out_primary=gathered[r], out_shadow=gathered_s[r])
    image = torch.empty((h * w, 2), dtype=torch.int32, device="cuda")
    core.assemble_tiles(gathered, w, h, tw, th, n, image, rank_stride=per_rank)
    core.sync()
    assert_hits_equal(image.cpu().numpy().view(rt.HIT_DTYPE).reshape(-1), full, "assembled primary")
    core.assemble_tiles(gathered_s, w, h, tw, th, n, image)
    core.sync()
    assert_hits_equal(image.cpu().numpy().view(rt.HIT_DTYPE).reshape(-1), full_s, "assembled shadow")


def test_full_size_properties_1080p(core, ob, rt, car_xyz9):
    """At BASELINE's frame size: repeatable, row bands == whole frame, counters consistent."""
    import torch
    core.set_scene(None, rt.triangles_from_xyz9(car_xyz9))
    core.build_grid()
    w, h = 1920, 1080
    cam = rt.Camera.for_scene("gltf").to_uniform(0.0, float(h))
    a = torch.empty((w * h, 2), dtype=torch.int32, device="cuda")
    b = torch.empty_like(a)
    sa, sb = torch.empty_like(a), torch.empty_like(a)
    core.trace_primary(cam, w, h, out_primary=a, out_shadow=sa)
    core.trace_primary(cam, w, h, out_primary=b, out_shadow=sb)
    core.sync()
    assert torch.equal(a, b) and torch.equal(sa, sb)
    bands = torch.empty_like(a)
    for r in range(2):  # 8-row bands dealt to two "ranks", written straight to their place
        sh = rt.RtShard(w, 8, r, 2)
        n = rt.shard_pixels(w, h, sh)
        tmp = torch.empty((n, 2), dtype=torch.int32, device="cuda")
        core.trace_primary(cam, w, h, shard=sh, shadow=False, out_primary=tmp)
        core.sync()
        rows = tmp.view(-1, 8, w, 2)
        for j in range(rows.shape[0]):
            y0 = (r + 2 * j) * 8
            k = min(8, h - y0)
            bands.view(h, w, 2)[y0:y0 + k] = rows[j, :k]
    assert torch.equal(bands, a)
    core.set_collect_stats(True)
    p, s = core.trace_primary(cam, w, h)
    st = core.stats()
    core.set_collect_stats(False)
    assert st["primary_hits"] == int((p["id"] != rt.RT_MISS).sum()) == st["shadow_rays"]
    assert st["shadow_hits"] == int((s["id"] != rt.RT_MISS).sum())
    assert st["rays"] == w * h + st["shadow_rays"]
    assert np.array_equal(a.cpu().numpy().view(rt.HIT_DTYPE).reshape(-1), p)
    hit = p["id"] != rt.RT_MISS
    assert np.all(p["t"][hit] > 0) and np.all(p["id"][hit] < 3132) and np.all(s["id"][~hit] == rt.RT_MISS)


@pytest.mark.parametrize("w,h", [(3840, 2160), (7680, 4320)])
def test_full_size_properties_massive_scene(core, ob, rt, w, h):
    """The massive_benchmark scene at the 4K and 8K frame sizes of BASELINE's configurations 3 and 5. The oracle
    cannot trace tens of millions of rays in test time, so: a random sample of pixels against the oracle (one ray
    at a time), tiles dealt to eight "ranks" and reassembled == the whole frame (hit records and the picture),
    every hit is the first of its three exact copies, counters agree with the frame."""
    import torch
    tris = ob.generate_test_triangles(30000)
    core.set_scene(None, tris)
    core.build_grid(cell_size=4.0)
    scene = ob.Scene(None, tris, cell_size=4.0)
    cam = rt.Camera((0, 0, 150), math.pi, 0.0).to_uniform(0.0, float(h))
    n = w * h
    prim = torch.empty((n, 2), dtype=torch.int32, device="cuda")
    shad = torch.empty_like(prim)
    core.set_collect_stats(True)
    core.trace_primary(cam, w, h, out_primary=prim, out_shadow=shad)
    st = core.stats()
    core.set_collect_stats(False)
    ids, sids = prim[:, 0], shad[:, 0]
    hits = int((ids != -1).sum())
    assert st["rays"] == n + hits and st["primary_hits"] == hits == st["shadow_rays"] and st["shadow_hits"] == int((sids != -1).sum())
    assert hits > n // 100 and int(ids[ids != -1].max()) < 10000 and bool(((sids == -1) | (ids != -1)).all())
    # sample against the oracle
    rng = np.random.default_rng(w)
    hp = torch.nonzero(ids != -1).flatten().cpu().numpy()
    sample = np.concatenate([rng.integers(0, n, 60), rng.choice(hp, 60)])
    got = prim[torch.from_numpy(sample).cuda()].cpu().numpy().view(rt.HIT_DTYPE).reshape(-1)
    for k, pix in enumerate(sample):
        info = ob.debug_pixel(scene, ocam(ob, cam), w, h, int(pix % w), int(pix // w))
        if info.hit:
            assert got["id"][k] == int(info.object_id) and np.float32(got["t"][k]).view(np.uint32) == np.float32(info.distance).view(np.uint32)
        else:
            assert got["id"][k] == rt.RT_MISS
    # eight ranks' tiles, gathered and reassembled
    world, tw = 8, 64
    per = max(rt.shard_pixels(w, h, rt.RtShard(tw, tw, r, world)) for r in range(world))
    packed = torch.empty((world, 2, per, 2), dtype=torch.int32, device="cuda")
    pic_packed = torch.zeros((world, per, 4), dtype=torch.uint8, device="cuda")
    for r in range(world):
        sh = rt.RtShard(tw, tw, r, world)
        core.trace_primary(cam, w, h, shard=sh, out_primary=packed[r, 0], out_shadow=packed[r, 1])
        core.render(cam, w, h, shard=sh, max_bounces=1, shadows=True, out=pic_packed[r])
    image = torch.empty((2, n, 2), dtype=torch.int32, device="cuda")
    for b in range(2):
        core.assemble_tiles(packed[0, b], w, h, tw, tw, world, image[b], rank_stride=2 * per)
    pic = torch.empty((n, 4), dtype=torch.uint8, device="cuda")
    core.assemble_tiles_rgba8(pic_packed, w, h, tw, tw, world, pic, rank_stride=per)
    whole = torch.empty((n, 4), dtype=torch.uint8, device="cuda")
    core.render(cam, w, h, max_bounces=1, shadows=True, out=whole)
    core.sync()
    assert torch.equal(image[0], prim) and torch.equal(image[1], shad) and torch.equal(pic, whole)
    # lit pixels are exactly the primary hits
    sky = ob.render(scene, ocam(ob, cam), 8, 8, max_bounces=1)[0]  # smoke: the oracle runs this scene too
    assert sky.shape == (64, 4)


def test_runs_on_torch_stream(core, ob, rt, car_xyz9):
    import torch
    scene = setup_scene(core, ob, None, ob.triangles_from_xyz9(car_xyz9))
    cam = rt.Camera.for_scene("gltf").to_uniform(0.0, 90.0)
    s = torch.cuda.Stream()
    core.set_stream(s.cuda_stream)
    out = torch.empty((160 * 90, 2), dtype=torch.int32, device="cuda")
    with torch.cuda.stream(s):
        core.trace_primary(cam, 160, 90, shadow=False, out_primary=out)
    s.synchronize()
    want, _, _ = ob.trace_primary(scene, ocam(ob, cam), 160, 90, shadow=False)
    assert_hits_equal(out.cpu().numpy().view(rt.HIT_DTYPE).reshape(-1), want)
    core.set_stream(None)


def test_frames_in_flight_on_two_streams(core, ob, rt):
    """Consecutive frames on alternating streams, each with its own output buffers, nothing synchronised in
    between. The scene has moving boxes, whose per-frame records live in scratch memory of the context that all
    frames share: the context has to order that itself. Then the scene is replaced while frames are in flight."""
    import torch
    boxes = random_boxes(ob, 80, 4, extent=50.0, moving_last=3)
    tris = ob.generate_test_triangles(2000)
    scene = setup_scene(core, ob, boxes, tris, cell_size=8.0)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    w, h = 192, 108
    times = [0.0, 0.31, 0.77, 1.9, 2.4, 3.3]
    cams = [rt.Camera((5, 25, 150), math.pi, -0.15).to_uniform(t, float(h)) for t in times]
    outs = [(torch.empty((w * h, 2), dtype=torch.int32, device="cuda"), torch.empty((w * h, 2), dtype=torch.int32, device="cuda"))
            for _ in times]
    for k, cam in enumerate(cams):
        core.set_stream(streams[k % 2].cuda_stream)
        core.trace_primary(cam, w, h, out_primary=outs[k][0], out_shadow=outs[k][1])
    tris2 = ob.generate_test_triangles(500)
    core.set_scene(boxes, tris2)  # must wait for every frame above
    core.build_grid(cell_size=8.0)
    last = core.trace_primary(cams[0], w, h)
    core.set_stream(None)
    core.sync()
    for k, cam in enumerate(cams):
        op, os_, _ = ob.trace_primary(scene, ocam(ob, cam), w, h, params=oparams(ob, core.params))
        assert_hits_equal(outs[k][0].cpu().numpy().view(rt.HIT_DTYPE).reshape(-1), op, f"frame {k} primary")
        assert_hits_equal(outs[k][1].cpu().numpy().view(rt.HIT_DTYPE).reshape(-1), os_, f"frame {k} shadow")
    scene2 = ob.Scene(boxes, tris2, cell_size=8.0)
    op, os_, _ = ob.trace_primary(scene2, ocam(ob, cams[0]), w, h, params=oparams(ob, core.params))
    assert_hits_equal(last[0], op, "after scene change")
    assert_hits_equal(last[1], os_, "after scene change")


def test_errors_are_reported(core, rt, ob):
    with pytest.raises(rt.RtError) as e:
        core.trace_primary(rt.Camera.for_scene("gltf").to_uniform(), 8, 8)
    assert e.value.code == -4 and "grid" in str(e.value)
    core.set_scene(None, ob.pyramid_triangles())
    meta = ob.Grid(None, ob.pyramid_triangles()).meta
    with pytest.raises(rt.RtError):
        core.set_grid_csr(meta, np.array([5, 3], np.uint32), np.zeros(5, np.uint32))  # decreasing offsets
    with pytest.raises(rt.RtError):
        core.update_vertices(np.zeros((7, 9), np.float32))  # more triangles than the scene has


@pytest.mark.gpu
def test_probe_copy_copies_and_rejects_what_it_cannot(rt, core):
    """rt_probe_copy (the bandwidth probe bench.py times) is a byte-exact copy for every size in 16-byte units and an
    error for anything else."""
    import torch
    src = torch.randint(-2**31, 2**31 - 1, (1 << 25,), dtype=torch.int32, device="cuda")
    dst = torch.empty_like(src)
    for units in (0, 1, 255, 256, 257, 5767245, src.numel() // 4):  # 16-byte units: part workgroups, whole ones, all
        words = 4 * units
        assert words <= src.numel() and words <= dst.numel()  # the kernel trusts the size it is given
        dst.fill_(7)
        torch.cuda.synchronize()
        core.probe_copy(src.data_ptr(), dst.data_ptr(), 4 * words)
        core.sync()
        assert torch.equal(dst[:words], src[:words]) and bool((dst[words:] == 7).all()), units
    host = np.zeros(64, np.int32)
    for args in ((src.data_ptr(), dst.data_ptr(), 24), (src.data_ptr() + 4, dst.data_ptr(), 32),
                 (host.ctypes.data, dst.data_ptr(), 64), (0, dst.data_ptr(), 64)):
        with pytest.raises(rt.RtError):
            core.probe_copy(*args)
