"""Known-answer tests: the reference's own unit tests for the hot-path routines, run against the
CPU oracle. Each test names the reference test it reproduces (file:line)."""
import math

import numpy as np
import pytest


# ---- src/math/ray.rs:42-71 ----------------------------------------------------------------

def test_intersect_aabb_hit(ob):  # src/math/ray.rs:43-51
    t = ob.intersect_aabb((0, 0, 0), (1, 0, 0), (5, -1, -1), (10, 1, 1))
    assert t > 0.0 and abs(t - 5.0) < 0.01
    assert t == 5.0  # the restatement is exact here


def test_intersect_aabb_miss(ob):  # src/math/ray.rs:54-60
    assert ob.intersect_aabb((0, 0, 0), (1, 0, 0), (5, 2, 2), (10, 3, 3)) < 0.0


def test_intersect_aabb_inside(ob):  # src/math/ray.rs:64-70
    t = ob.intersect_aabb((5, 0, 0), (1, 0, 0), (0, -1, -1), (10, 1, 1))
    assert t > 0.0
    assert t == 5.0  # origin inside: t_far


def test_intersect_aabb_criterion_scenes(ob):  # benches/bounding_volumes.rs:63-111
    assert ob.intersect_aabb((0, 0, 0), (0, 0, -1), (-1, -1, -6), (1, 1, -4)) == 4.0
    assert ob.intersect_aabb((0, 0, 0), (0, 0, -1), (10, 10, -6), (12, 12, -4)) == -1.0


# ---- src/math/grid.rs:21-46 -----------------------------------------------------------------

def test_world_to_cell_origin(ob):
    assert ob.world_to_cell((0, 0, 0), (0, 0, 0), 16.0) == (0, 0, 0)


def test_world_to_cell_offset(ob):
    assert ob.world_to_cell((20, 30, 40), (0, 0, 0), 10.0) == (2, 3, 4)


def test_world_to_cell_negative(ob):
    assert ob.world_to_cell((-5, -5, -5), (-10, -10, -10), 10.0) == (0, 0, 0)


# ---- src/math/aabb.rs:35-107 ------------------------------------------------------------------

def test_aabb_center(ob):
    assert ob.aabb_center((0, 0, 0), (2, 4, 6)) == (1.0, 2.0, 3.0)
    assert ob.aabb_center((-2, -4, -6), (2, 4, 6)) == (0.0, 0.0, 0.0)


def test_aabb_surface_area(ob):
    assert abs(ob.aabb_surface_area((0, 0, 0), (1, 1, 1)) - 6.0) < 0.01
    assert abs(ob.aabb_surface_area((0, 0, 0), (2, 3, 4)) - 52.0) < 0.01


@pytest.mark.parametrize("a,b,want", [
    (((0, 0, 0), (1, 1, 1)), ((2, 2, 2), (3, 3, 3)), ((0, 0, 0), (3, 3, 3))),      # non-overlapping
    (((0, 0, 0), (2, 2, 2)), ((1, 1, 1), (3, 3, 3)), ((0, 0, 0), (3, 3, 3))),      # overlapping
    (((0, 0, 0), (5, 5, 5)), ((1, 1, 1), (2, 2, 2)), ((0, 0, 0), (5, 5, 5))),      # contained
    (((-3, -3, -3), (-1, -1, -1)), ((1, 1, 1), (3, 3, 3)), ((-3, -3, -3), (3, 3, 3))),  # negative coords
])
def test_aabb_union(ob, a, b, want):
    mn, mx = ob.aabb_union(a[0], a[1], b[0], b[1])
    assert mn == tuple(map(float, want[0])) and mx == tuple(map(float, want[1]))


# ---- src/core/triangle_intersection.rs:206-379 ---------------------------------------------------

TRI = ((-1.0, 0.0, -5.0), (1.0, 0.0, -5.0), (0.0, 1.0, -5.0))  # create_test_triangle, :210-216


def test_moller_trumbore_hit(ob):  # :219-232 (the ray hits the edge v0-v1: u = 0.5, v = 0)
    h = ob.moller_trumbore((0, 0, 0), (0, 0, -1), *TRI)
    assert h is not None and h["t"] > 0.0
    assert 0.0 <= h["u"] <= 1.0 and 0.0 <= h["v"] <= 1.0 and h["u"] + h["v"] <= 1.0
    assert (h["t"], h["u"], h["v"]) == (5.0, 0.5, 0.0)


def test_moller_trumbore_miss(ob):  # :235-242
    d = np.array([5.0, 0.0, -1.0], np.float32)
    d = d * np.float32(1.0 / math.sqrt(26.0))
    assert ob.moller_trumbore((0, 0, 0), d, *TRI) is None


def test_behind_ray(ob):  # :245-252
    assert ob.moller_trumbore((0, 0, 0), (0, 0, 1), *TRI) is None


def test_barycentric_coordinates(ob):  # :255-265
    h = ob.moller_trumbore((0, 0, 0), (0, 0, -1), *TRI)
    u, v = h["u"], h["v"]
    assert abs(u + v + (1.0 - u - v) - 1.0) < 1e-5


def test_uv_interpolation(ob):  # :268-283
    h = ob.moller_trumbore((0, 0, 0), (0, 0, -1), *TRI)
    u, v, w = h["u"], h["v"], 1.0 - h["u"] - h["v"]
    uv0, uv1, uv2 = (0.0, 0.0), (1.0, 0.0), (0.5, 1.0)
    s = w * uv0[0] + u * uv1[0] + v * uv2[0]
    t = w * uv0[1] + u * uv1[1] + v * uv2[1]
    assert 0.0 <= s <= 1.0 and 0.0 <= t <= 1.0


def test_watertight_vs_moller_trumbore(ob):  # :286-301
    h1 = ob.moller_trumbore((0, 0, 0), (0, 0, -1), *TRI)
    h2 = ob.watertight((0, 0, 0), (0, 0, -1), *TRI)
    assert h1 is not None and h2 is not None and abs(h1["t"] - h2["t"]) < 0.1


def _two_triangles(ob):
    t = np.zeros(2, ob.TRIANGLE_DTYPE)
    t["v0"], t["v1"], t["v2"] = [(-1, 0, -10), (-1, 0, -5)], [(1, 0, -10), (1, 0, -5)], [(0, 1, -10), (0, 1, -5)]
    return t


def test_triangle_data_intersect_and_batch(ob):  # :304-354
    tris = _two_triangles(ob)
    got = ob.batch_intersect_triangles((0, 0, 0), (0, 0, -1), tris, [0, 1])
    assert got is not None and got[0] == 1  # the closer triangle is index 1
    assert got[1]["t"] == 5.0


def test_batch_tie_keeps_first(ob):  # strict `<` at :196 - an equal t never replaces the first hit
    tris = _two_triangles(ob)
    tris[0] = tris[1]
    assert ob.batch_intersect_triangles((0, 0, 0), (0, 0, -1), tris, [0, 1])[0] == 0
    assert ob.batch_intersect_triangles((0, 0, 0), (0, 0, -1), tris, [1, 0])[0] == 1


def test_edge_case_parallel_ray(ob):  # :357-364
    assert ob.moller_trumbore((0, 0, -5), (1, 0, 0), *TRI) is None


def test_normal_calculation(ob):  # :367-378
    n = ob.moller_trumbore((0, 0, 0), (0, 0, -1), *TRI)["normal"]
    assert abs(math.sqrt(sum(c * c for c in n)) - 1.0) < 1e-5 and n[2] > 0.0


def test_rust_vs_shader_epsilon(ob):
    """src/core/triangle_intersection.rs:38,74 (eps 1e-6, reject t < eps) against
    raytracer_unified.wgsl:3,273 (eps 1e-5, accept t > eps)."""
    tri = ((-1.0, -1.0, -5e-6), (1.0, -1.0, -5e-6), (0.0, 1.0, -5e-6))
    assert ob.moller_trumbore((0, 0, 0), (0, 0, -1), *tri, wgsl_compat=0) is not None
    assert ob.moller_trumbore((0, 0, 0), (0, 0, -1), *tri, wgsl_compat=1) is None


# ---- src/core/benchmark.rs:328-425 ------------------------------------------------------------------

def test_generators(ob):  # :351-364
    assert len(ob.generate_test_triangles(100)) == 100
    o, d = ob.generate_test_rays(50)
    assert len(o) == 50
    assert np.all(np.abs(np.linalg.norm(d.astype(np.float64), axis=1) - 1.0) < 0.01)


def test_generate_test_triangles_values(ob):  # :102-109: x = ((i*7919) % 10000)/100 - 50, period 10000
    t = ob.generate_test_triangles(10001)
    assert tuple(t["v0"][0]) == (-50.0, -50.0, -50.0)
    i = 3
    x = np.float32(np.float32((i * 7919) % 10000) / np.float32(100.0)) - np.float32(50.0)
    assert t["v0"][i][0] == x and t["v1"][i][0] == x + np.float32(1.0)
    assert np.array_equal(t["v0"][0], t["v0"][10000])  # positions repeat with period 10000


# ---- src/core/bvh.rs:361-478, src/core/benchmark.rs:412-425 ----------------------------------------------

def test_bvh_small_and_stats(ob):
    tris = ob.generate_test_triangles(5)  # 5 > MAX_LEAF_SIZE: the root is internal (src/core/bvh.rs:61)
    b = ob.Bvh(tris)
    s = b.stats()
    assert s["prims"] == 5 and s["leaves"] >= 2 and s["nodes"] == 2 * s["leaves"] - 1


def test_bvh_hit_and_miss(ob):  # shape of src/core/benchmark.rs:412-425 with triangles
    tris = _two_triangles(ob)
    b = ob.Bvh(tris)
    # origin y = 0.25: with y = 0 the example's plain 1/dir slab test (massive_benchmark.rs:311-320)
    # forms 0 * inf = NaN against the triangles' y = 0 bound and rejects the node
    t, hits = b.trace([(0, 0.25, 0), (0, 0.25, 0)], [(0, 0, -1), (0, 1, 0)])
    assert hits == 1 and t[0] == 5.0 and t[1] == -1.0


def test_bvh_equals_linear_scan(ob):  # examples/massive_benchmark.rs:226-270 vs :322-342
    tris = ob.generate_test_triangles(2000)
    o, d = ob.generate_test_rays(500)
    tb, hb = ob.Bvh(tris).trace(o, d, n_threads=2)
    tl, hl = ob.linear_scan(tris, o, d, n_threads=2)
    assert hb == hl and np.array_equal(tb, tl)
