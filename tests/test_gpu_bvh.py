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
    import os
    for lanes in ("1", "32", None):  # a lane to a ray, 32 lanes to a ray over the tree's parts, the library's own choice
        if lanes is None:
            os.environ.pop("RT_BVH_LANES", None)
        else:
            os.environ["RT_BVH_LANES"] = lanes
        try:
            got = core.trace_rays_bvh(o, d)
        finally:
            os.environ.pop("RT_BVH_LANES", None)
        bad = np.flatnonzero(got.view(np.uint32) != want.view(np.uint32))
        assert bad.size == 0, f"{what}, lanes {lanes}: {bad.size} of {len(o)} t values differ, first {bad[:5]}: {got[bad[:5]]} vs {want[bad[:5]]}"
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
    assert t[0] == np.float32(5.0) and t[1] == np.float32(-1.0)  # a tree of one leaf: one part
    # trees of 2 .. 40 triangles: fewer parts than lanes, parts that are single leaves
    for n in (2, 5, 6, 9, 17, 40):
        tris = ob.generate_test_triangles(n)
        core.set_scene(None, tris)
        o, d = ob.generate_test_rays(300)
        check(core, ob, tris, o, d, f"{n} triangles")


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


def test_copies_in_a_leaf_are_passed_over_exactly(ob, rt):
    """The generator's scenes hold every triangle many times over; copies share a leaf, and the walk tests the first of a
    run only (rt_params.skip_duplicate_triangles, on by default). Same t bits with the switch off, and as the oracle
    - which tests everything - has them; also for a scene where copies are NOT neighbours in their leaf (two different
    triangles with one centroid, interleaved), which must lose nothing."""
    o, d = ob.generate_test_rays(4000)
    tris = ob.generate_test_triangles(50000)  # five copies of each
    a = np.array([[-1, -1, -5, 1, -1, -5, 0, 2, -5]], np.float32)
    b = np.array([[-1, 1, -6, 1, 1, -6, 0, -2, -6]], np.float32)  # same centroid as a
    mixed = ob.triangles_from_xyz9(np.concatenate([a, b, a, b, b, a, a, a, b], axis=0))
    o2 = np.array([[0, 0, 0], [0.3, 0.2, 0], [0, 0, -5.5], [5, 5, 0]], np.float32)
    d2 = np.array([[0, 0, -1], [0, 0, -1], [0, 0, -1], [0, 0, -1]], np.float32)
    for scene, oo, dd in ((tris, o, d), (mixed, o2, d2)):
        got = []
        for skip in (1, 0):
            core = rt.TraceCore(0, rt.default_params(skip_duplicate_triangles=skip))
            core.set_scene(None, scene)
            want, hits = check(core, ob, scene, oo, dd, f"skip {skip}")
            got.append(want)
            core.close()
        assert hits > 0 and np.array_equal(got[0].view(np.uint32), got[1].view(np.uint32))


def same_tree(core, what):
    """rt_build_bvh (GPU, level by level) against rt_build_bvh_host (the reference's recursion): node for node."""
    st_gpu = core.build_bvh()
    nodes, leaf_first, prims = core.read_bvh()
    st_host = core.build_bvh(on_host=True)
    hn, hl, hp = core.read_bvh()
    assert st_gpu == st_host, (what, st_gpu, st_host)
    assert np.array_equal(prims, hp), f"{what}: primitive order differs at {np.flatnonzero(prims != hp)[:5]}"
    assert np.array_equal(leaf_first, hl), what
    bad = np.flatnonzero((nodes.view(np.uint32) != hn.view(np.uint32)).any(axis=1))
    assert bad.size == 0, f"{what}: {bad.size} of {len(nodes)} nodes differ, first {bad[:3]}: {nodes[bad[:3]]} vs {hn[bad[:3]]}"
    return st_gpu


def test_gpu_builder_makes_the_reference_tree(core, ob, rt, car_xyz9, phoenix_xyz9):
    rng = np.random.default_rng(7)
    # sizes around the leaf limit, the generator's scenes (30 000: every triangle three times, so partitions fail and
    # leaves grow past four), both bundled models, random soups with degenerate and repeated triangles
    for n in (1, 2, 4, 5, 6, 9, 17, 100):
        core.set_scene(None, ob.generate_test_triangles(n))
        same_tree(core, f"{n} generator triangles")
    for n in (1000, 30000):
        core.set_scene(None, ob.generate_test_triangles(n))
        st = same_tree(core, f"{n} generator triangles")
        assert st["total_primitives"] == n
    for name, xyz9 in (("static model", car_xyz9), ("animated model", phoenix_xyz9)):
        core.set_scene(None, rt.triangles_from_xyz9(xyz9))
        st = same_tree(core, name)
        assert st["node_count"] > len(xyz9) // 4
    for trial in range(6):
        n = int(rng.integers(50, 20000))
        c = rng.uniform(-50, 50, (n, 1, 3)).astype(np.float32)
        v = (c + rng.normal(0, [0.01, 1.0, 30.0][trial % 3], (n, 3, 3)).astype(np.float32)).astype(np.float32)
        v[rng.integers(0, n, n // 20)] = v[rng.integers(0, n, n // 20)]      # exact copies
        k = rng.integers(0, n, n // 25)
        v[k, 1] = v[k, 0]                                                    # degenerate: two equal vertices
        if trial % 2:
            v[:, :, trial % 3] = np.float32(3.0)                             # flat along one axis: that axis is never split
        core.set_scene(None, rt.triangles_from_xyz9(v.reshape(n, 9)))
        same_tree(core, f"random soup {trial} ({n} triangles)")
    # all triangles equal: the root's partition fails, one leaf holds everything, in the partition's order
    one = np.tile(np.array([[0, 0, 0, 1, 0, 0, 0, 1, 0]], np.float32), (37, 1))
    core.set_scene(None, rt.triangles_from_xyz9(one))
    st = same_tree(core, "37 copies of one triangle")
    assert st["node_count"] == 1 and st["leaf_count"] == 1
