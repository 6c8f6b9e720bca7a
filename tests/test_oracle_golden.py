"""The CPU oracle against the committed fixtures (tests/golden/oracle_golden.npz, written by
tests/golden/make_fixtures.py) and against itself: grid traversal must agree with a brute-force
scan over every triangle. The reference has no test or golden vector for grid binning, traversal
or ray generation, so these pin the oracle's own behaviour over time, not the reference's."""
import math

import numpy as np
import pytest


def _eq_hits(a, b):
    return np.array_equal(a["id"], b["id"]) and np.array_equal(a["t"].view(np.uint32), b["t"].view(np.uint32))


def test_grid_statistics_of_the_bundled_models(ob, car_xyz9, phoenix_xyz9):
    car = ob.Grid(None, ob.triangles_from_xyz9(car_xyz9))
    assert car.meta.fine_dims == (67, 12, 84) and car.n_cells == 67536 and car.n_refs == 34430
    occ = np.diff(car.cell_start.astype(np.int64))
    assert int((occ > 0).sum()) == 9305 and int(occ.max()) == 76
    ph = ob.Grid(None, ob.triangles_from_xyz9(phoenix_xyz9))
    assert ph.n_cells == 70760 and ph.n_refs == 37034 and int(np.diff(ph.cell_start.astype(np.int64)).max()) == 150


def test_grids_match_golden(ob, golden, car_xyz9, phoenix_xyz9):
    for name, tris, kw in (("pyramid", ob.pyramid_triangles(), {}), ("pyramid_c1", ob.pyramid_triangles(), dict(cell_size=1.0)),
                           ("car", ob.triangles_from_xyz9(car_xyz9), {}), ("phoenix", ob.triangles_from_xyz9(phoenix_xyz9), {})):
        g = ob.Grid(None, tris, **kw)
        assert np.array_equal(g.cell_start, golden[f"{name}_cell_start"]), name
        assert np.array_equal(g.indices, golden[f"{name}_indices"]), name
    assert tuple(golden["car_dims"]) == (67, 12, 84)


def test_cell_lists_are_ascending(ob, car_xyz9):
    g = ob.Grid(None, ob.triangles_from_xyz9(car_xyz9))
    for c in range(g.n_cells):
        seg = g.indices[g.cell_start[c]:g.cell_start[c + 1]]
        assert np.all(seg[1:] > seg[:-1])


def test_trace_matches_golden(ob, golden, car_xyz9):
    cam = ob.camera_to_uniform((0, 8, 20), math.pi, -0.5, 0.0, 36.0)
    for name, kw in (("pyramid", {}), ("pyramid_c1", dict(cell_size=1.0))):
        p, s, _ = ob.trace_primary(ob.Scene(None, ob.pyramid_triangles(), **kw), cam, 64, 36)
        assert _eq_hits(p, golden[f"{name}_primary"]) and _eq_hits(s, golden[f"{name}_shadow"]), name
    cam = ob.camera_to_uniform((200, 200, 300), 3.35, -0.28, 0.0, 36.0)
    p, s, _ = ob.trace_primary(ob.Scene(None, ob.triangles_from_xyz9(car_xyz9)), cam, 64, 36)
    assert _eq_hits(p, golden["car_primary"]) and _eq_hits(s, golden["car_shadow"])
    tri = ob.generate_test_triangles(10000)
    o, d = ob.generate_test_rays(1000)
    hits, _ = ob.trace_rays(ob.Scene(None, tri), o, d)
    assert _eq_hits(hits, golden["bench10k_rays1000_hits"])


def test_pyramid_picture(ob):
    """src/scenes/pyramid.rs:22-61 seen from src/camera.rs:64: the camera sits at +z looking towards -z, so
    the face it sees is the z = +4 one (triangle 2) and nothing hits the downward-facing base from above."""
    cam = ob.camera_to_uniform((0, 8, 20), math.pi, -0.5, 0.0, 36.0)
    p, s, st = ob.trace_primary(ob.Scene(None, ob.pyramid_triangles()), cam, 64, 36)
    ids, counts = np.unique(p["id"][p["id"] != ob.MISS], return_counts=True)
    assert set(ids.tolist()) <= {1, 2, 3} and ids[np.argmax(counts)] == 2  # 1 and 3 only along the silhouette edges
    assert st["primary_hits"] == 26 and st["shadow_rays"] == 26 and st["rays"] == 64 * 36 + 26


@pytest.mark.parametrize("cell_size", [16.0, 4.0])
def test_dda_equals_bruteforce_car(ob, car_xyz9, cell_size):
    """The traversal must find what a scan over all triangles finds. Exceptions that the shader shares:
    rays cut short by the 200-step cap, and ties (equal t) where traversal order picks another id."""
    scene = ob.Scene(None, ob.triangles_from_xyz9(car_xyz9), cell_size=cell_size)
    cam = ob.camera_to_uniform((200, 200, 300), 3.35, -0.28, 0.0, 90.0)
    o, d = ob.primary_rays(cam, 160, 90)
    params = ob.Params.default(max_dda_steps=100000)
    got, _ = ob.trace_rays(scene, o, d, params=params, cam=cam)
    want = ob.trace_rays_bruteforce(scene, o, d, params=params, cam=cam)
    assert np.array_equal(got["id"] == ob.MISS, want["id"] == ob.MISS)
    assert np.array_equal(got["t"].view(np.uint32), want["t"].view(np.uint32))
    tie = got["id"] != want["id"]
    assert tie.mean() < 0.001  # id may differ only where two triangles give the very same t


def test_dda_equals_bruteforce_with_boxes(ob):
    rng = np.random.default_rng(7)
    boxes = np.zeros(40, ob.BOX_DTYPE)
    for i in range(40):
        c = rng.uniform(-40, 40, 3).astype(np.float32)
        h = rng.uniform(1, 6, 3).astype(np.float32)
        boxes[i] = ob.make_box(c - h, c + h)
    tris = ob.generate_test_triangles(300)
    scene = ob.Scene(boxes, tris, cell_size=8.0)
    cam = ob.camera_to_uniform((0, 10, 120), math.pi, -0.05, 0.3, 64.0)
    o, d = ob.primary_rays(cam, 96, 64)
    params = ob.Params.default(max_dda_steps=100000, moving_box_count=0)
    got, _ = ob.trace_rays(scene, o, d, params=params, cam=cam)
    want = ob.trace_rays_bruteforce(scene, o, d, params=params, cam=cam)
    assert np.array_equal(got["t"].view(np.uint32), want["t"].view(np.uint32))
    assert (got["id"] != want["id"]).mean() < 0.002
    assert (got["id"] < 40).sum() > 100 and (got["id"] != ob.MISS).sum() > (got["id"] < 40).sum()


def test_per_cell_cap_keeps_lowest_ids(ob):
    """src/grid.rs:92-100: inserts into a full cell are dropped, so the lowest ids survive."""
    tris = ob.generate_test_triangles(3000)  # 7^3 cells of size 16: about 9 per cell
    g = ob.Grid(None, tris, max_per_cell=4)
    full = ob.Grid(None, tris)
    assert g.dropped > 0 and int(np.diff(g.cell_start.astype(np.int64)).max()) == 4
    for c in range(g.n_cells):
        a = g.indices[g.cell_start[c]:g.cell_start[c + 1]]
        b = full.indices[full.cell_start[c]:full.cell_start[c + 1]]
        assert np.array_equal(a, b[:4])


def test_triangle_grid_mode(ob, car_xyz9):
    """src/grid_triangles.rs: same binning for in-range scenes, cap 256, dims clamped to [1, 1024]."""
    tris = ob.triangles_from_xyz9(car_xyz9)
    a, b = ob.Grid(None, tris), ob.Grid(None, tris, mode=ob.GRID_TRIANGLES)
    assert b.max_per_cell == 256 and a.meta.fine_dims == b.meta.fine_dims
    assert np.array_equal(a.cell_start, b.cell_start) and np.array_equal(a.indices, b.indices)
    empty = ob.Grid(None, None, mode=ob.GRID_TRIANGLES)
    assert empty.meta.fine_dims == (1, 1, 1) and list(empty.meta.bounds_min) == [-1.0] * 3  # :30-43: no padding
    empty_h = ob.Grid(None, None)
    assert list(empty_h.meta.bounds_min) == [-2.0] * 3 and empty_h.meta.fine_dims == (1, 1, 1)  # grid.rs:117-134
    tiny = ob.Grid(None, tris, cell_size=0.05, mode=ob.GRID_TRIANGLES, max_per_cell=4)
    assert max(tiny.meta.fine_dims) <= 1024


def test_dense_export_roundtrip(ob):
    g = ob.Grid(None, ob.pyramid_triangles(), cell_size=4.0)
    dense = g.dense.reshape(g.n_cells, 32784).view(np.uint32)
    for c in range(g.n_cells):
        n = dense[c, 8192]
        assert n == g.cell_start[c + 1] - g.cell_start[c]
        assert np.array_equal(dense[c, :n], g.indices[g.cell_start[c]:g.cell_start[c + 1]])
        assert not dense[c, n:8192].any()


def test_shadow_ray_definition(ob):
    """A floor under a roof: every floor hit looks towards -normalize((0.5,-1,0.3)) from 0.001 above the
    floor; the roof occludes exactly those floor points whose shadow ray reaches it."""
    floor = [(-50, 0, -50, 50, 0, -50, 50, 0, 50), (-50, 0, -50, 50, 0, 50, -50, 0, 50)]
    roof = [(-5, 10, -5, 5, 10, -5, 5, 10, 5), (-5, 10, -5, 5, 10, 5, -5, 10, 5)]
    scene = ob.Scene(None, ob.triangles_from_xyz9(np.array(floor + roof, np.float32)))
    cam = ob.camera_to_uniform((0, 40, 60), math.pi, -0.6, 0.0, 48.0)
    p, s, st, so, sd = ob.trace_primary(scene, cam, 64, 48, want_shadow_rays=True)
    L = np.array([0.5, -1.0, 0.3]) / np.linalg.norm([0.5, -1.0, 0.3])
    floor_hit = p["id"] < 2
    assert floor_hit.sum() > 500
    assert np.allclose(sd[floor_hit], -L, atol=1e-6)
    assert np.allclose(so[floor_hit][:, 1], 0.001, atol=1e-5)  # lifted along the normal turned towards the camera
    occluded = s["id"] != ob.MISS
    assert occluded.sum() > 0 and set(np.unique(s["id"][occluded]).tolist()) <= {2, 3}
    # geometric check: shadow ray reaches y = 10 at origin + 10/L_y * dir
    src = so[floor_hit].astype(np.float64)
    reach = src + ((10.0 - src[:, 1:2]) / -L[1]) * (-L)
    inside = (np.abs(reach[:, 0]) < 5) & (np.abs(reach[:, 2]) < 5)
    assert (inside != occluded[floor_hit]).sum() <= 4  # only points whose shadow ray grazes the roof's rim may differ
    assert st["shadow_rays"] == st["primary_hits"]


def test_shards_tile_the_frame(ob, car_xyz9):
    scene = ob.Scene(None, ob.triangles_from_xyz9(car_xyz9))
    cam = ob.camera_to_uniform((200, 200, 300), 3.35, -0.28, 0.0, 45.0)
    w, h, tw, th, n = 70, 45, 16, 8, 3
    full, fs, _ = ob.trace_primary(scene, cam, w, h)
    img = np.zeros(w * h, ob.HIT_DTYPE)
    seen = np.zeros(w * h, bool)
    tiles_x = -(-w // tw)
    for r in range(n):
        sh = ob.Shard(tw, th, r, n)
        p, _, _ = ob.trace_primary(scene, cam, w, h, shard=sh)
        p = p.reshape(-1, th, tw)
        for j in range(len(p)):
            tile = r + j * n
            ty, tx = divmod(tile, tiles_x)
            for ly in range(th):
                for lx in range(tw):
                    x, y = tx * tw + lx, ty * th + ly
                    if x < w and y < h:
                        img[y * w + x] = p[j, ly, lx]
                        seen[y * w + x] = True
                    else:
                        assert p[j, ly, lx]["id"] == ob.MISS
    assert seen.all() and _eq_hits(img, full)
