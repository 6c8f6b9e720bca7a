"""The BVH path (rt_build_bvh / rt_trace_rays_bvh) against the oracle's restatement of the reference's CPU benchmark
structure (src/core/bvh.rs build, examples/massive_benchmark.rs traversal): same tree statistics, same t bits."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def check(core, ob, tris, o, d, what):
    st = core.build_bvh()
    b = ob.Bvh(tris)
    want_st = b.stats()
    assert (st["node_count"], st["leaf_count"], st["max_depth"], st["total_primitives"]) == \
        (want_st["nodes"], want_st["leaves"], want_st["max_depth"], want_st["prims"]), what
    want, hits = b.trace(o, d, n_threads=4)
    got = core.trace_rays_bvh(o, d)
    bad = np.flatnonzero(got.view(np.uint32) != want.view(np.uint32))
    assert bad.size == 0, f"{what}: {bad.size} of {len(o)} t values differ, first {bad[:5]}: {got[bad[:5]]} vs {want[bad[:5]]}"
    return want, hits


def test_reference_known_answers(core, ob, rt):
    # src/core/bvh.rs:380-425 shapes: five primitives -> internal root; tiny scene hit along -z, miss along +y
    tris = ob.generate_test_triangles(5)
    core.set_scene(None, tris)
    st = core.build_bvh()
    assert st["node_count"] >= 3 and st["total_primitives"] == 5 and st["leaf_count"] >= 2
    one = ob.triangles_from_xyz9(np.array([[-1, -1, -5, 1, -1, -5, 0, 1, -5]], np.float32))
    core.set_scene(None, one)
    core.build_bvh()
    t = core.trace_rays_bvh(np.array([[0, 0, 0], [0, 0, 0]], np.float32), np.array([[0, 0, -1], [0, 1, 0]], np.float32))
    assert t[0] == np.float32(5.0) and t[1] == np.float32(-1.0)


def test_generator_scenes(core, ob, rt):
    for n_tris, n_rays in ((1000, 5000), (30000, 20000)):  # 30 000: every triangle three times
        tris = ob.generate_test_triangles(n_tris)
        o, d = ob.generate_test_rays(n_rays)
        core.set_scene(None, tris)
        want, hits = check(core, ob, tris, o, d, f"{n_tris} triangles")
        assert hits > 100
        lin, _ = ob.linear_scan(tris[:1000], o[:500], d[:500]) if n_tris == 1000 else (None, 0)
        if lin is not None:
            assert np.array_equal(lin.view(np.uint32), want[:500].view(np.uint32))  # BVH == testing everything


def test_car_frame_rays_and_degenerate_directions(core, ob, rt, car_xyz9):
    import torch
    tris = ob.triangles_from_xyz9(car_xyz9)
    core.set_scene(None, tris)
    cam = ob.camera_to_uniform((200, 200, 300), 3.35, -0.28, 0.0, 135.0)
    o, d = ob.primary_rays(cam, 240, 135)
    # axis-aligned and zero components: 1/0 = inf and 0*inf = NaN inside the example's box test
    o2 = np.array([(0, 500, 0), (0, 100, 0), (0, 100, 0), (-2000, 100, 0), (0, 100, 2000), (0, 100, 0)], np.float32)
    d2 = np.array([(0, -1, 0), (0, -1, -0.0), (1e-9, -1, -1e-9), (1, 0, 0), (0, 0, -1), (0, 0, 0)], np.float32)
    o, d = np.concatenate([o, o2]), np.concatenate([d, d2])
    want, hits = check(core, ob, tris, o, d, "car")
    assert hits > 240 * 135 // 3
    # closest t of the BVH walk == closest t of the grid walk wherever both hit (same triangle test, same constants)
    core.build_grid()
    grid = core.trace_rays(o[:240 * 135], d[:240 * 135])
    both = (grid["id"] != rt.RT_MISS) & (want[:240 * 135] >= 0)
    assert both.sum() > 10000 and np.array_equal(grid["t"][both].view(np.uint32), want[:240 * 135][both].view(np.uint32))
    # device memory in and out; stale after the vertices change
    to, td = torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda()
    out = torch.empty(len(o), dtype=torch.float32, device="cuda")
    core.trace_rays_bvh(to, td, out=out)
    core.sync()
    assert np.array_equal(out.cpu().numpy().view(np.uint32), want.view(np.uint32))
    core.update_vertices(car_xyz9 + np.float32(1.0))
    with pytest.raises(rt.RtError):
        core.trace_rays_bvh(o, d)
