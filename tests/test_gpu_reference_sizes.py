"""Parity at the sizes the reference's own benchmark uses (examples/massive_benchmark.rs:62-98: 100 K, 1 M, 10 M
triangles from src/core/benchmark.rs:99-123) on the reference's grid (cell 16, cap 8192, src/grid.rs:5-7), through the
C ABI. Bar: bit-exact ids and t bit patterns, as everywhere else.

The oracle cannot trace a 4K frame of these scenes in test time (every ray tests hundreds to thousands of candidates
per cell), so whole tiles of the frame are traced by both sides through the same rt_shard (a shard whose stride is the
tile count is exactly one tile), and the frame as a whole is held together by properties: duplicate removal on == off,
tile == the same pixels of the whole frame."""
import math

import numpy as np
import pytest

from test_gpu_parity import assert_hits_equal, ocam, oparams

pytestmark = pytest.mark.gpu


def one_tile(rt, ob, w, h, tile, tx, ty):
    tiles_x, tiles_y = -(-w // tile), -(-h // tile)
    first = ty * tiles_x + tx
    return rt.RtShard(tile, tile, first, tiles_x * tiles_y), ob.Shard(tile, tile, first, tiles_x * tiles_y), first


def tile_of_frame(frame, w, h, tile, tx, ty):
    """The tile's pixels cut out of a row-major frame, in the tile-local row-major order a shard writes."""
    img = frame.reshape(h, w)
    out = np.zeros((tile, tile), frame.dtype)
    out["id"] = 0xFFFFFFFF
    ys, xs = min(tile, h - ty * tile), min(tile, w - tx * tile)
    out[:ys, :xs] = img[ty * tile:ty * tile + ys, tx * tile:tx * tile + xs]
    return out.reshape(-1)


@pytest.mark.parametrize("pose", ["outside", "inside"])
def test_massive_100k_on_the_reference_grid_4k(rt, ob, pose):
    """100 000 triangles, cell 16 (7 x 7 x 7 cells, ~330 references each), 3840 x 2160 primary + shadow.
    outside: the camera the bench line uses; inside: a camera inside the volume, so every primary ray walks the grid."""
    import torch
    w, h, tile = 3840, 2160, 32
    tris = ob.generate_test_triangles(100000)
    cam = (rt.Camera((0, 0, 150), math.pi, 0.0) if pose == "outside" else rt.Camera((3.0, -2.0, 49.0), math.pi + 0.2, -0.1)).to_uniform(0.0, float(h))
    scene = ob.Scene(None, tris)
    frames = {}
    for skip in (1, 0):
        core = rt.TraceCore(0, rt.default_params(skip_duplicate_triangles=skip))
        core.set_scene(None, tris)
        meta, n_cells, n_refs = core.build_grid()
        assert meta.fine_dims == (7, 7, 7) and n_refs == scene.grid.n_refs
        cs, idx = core.read_grid_csr()
        assert np.array_equal(cs, scene.grid.cell_start) and np.array_equal(idx, scene.grid.indices)
        assert core.trace_refs() * 9 < n_refs if skip else core.trace_refs() == n_refs  # ten exact copies of everything
        p = torch.empty((w * h, 2), dtype=torch.int32, device="cuda")
        s = torch.empty_like(p)
        core.trace_primary(cam, w, h, out_primary=p, out_shadow=s)
        core.sync()
        frames[skip] = (p.cpu().numpy().view(rt.HIT_DTYPE).reshape(-1), s.cpu().numpy().view(rt.HIT_DTYPE).reshape(-1))
        if skip:  # whole tiles against the oracle: the centre, a corner, an edge (the last, partial row of tiles)
            for tx, ty in ((60, 33), (0, 0), (119, 67), (47, 20)):
                sh, osh, _ = one_tile(rt, ob, w, h, tile, tx, ty)
                gp, gs = core.trace_primary(cam, w, h, shard=sh)
                op, os_, st = ob.trace_primary(scene, ocam(ob, cam), w, h, params=oparams(ob, core.params), shard=osh, n_threads=8)
                assert_hits_equal(gp, op, f"{pose} tile ({tx},{ty}) primary")
                assert_hits_equal(gs, os_, f"{pose} tile ({tx},{ty}) shadow")
                assert_hits_equal(tile_of_frame(frames[1][0], w, h, tile, tx, ty), op, f"{pose} tile ({tx},{ty}) of the whole frame")
                assert_hits_equal(tile_of_frame(frames[1][1], w, h, tile, tx, ty), os_, f"{pose} tile ({tx},{ty}) of the whole frame, shadow")
        core.close()
    assert_hits_equal(frames[1][0], frames[0][0], "duplicates left out vs every entry tested, primary")
    assert_hits_equal(frames[1][1], frames[0][1], "duplicates left out vs every entry tested, shadow")
    ids = frames[1][0]["id"]
    hit = ids != rt.RT_MISS
    assert np.all(ids[hit] < 10000)  # every hit is the first of its ten exact copies
    if pose == "inside":
        assert hit.mean() > 0.15, hit.mean()
    else:
        assert 0.01 < hit.mean() < 0.1


def test_cap_8192_drops_ids_at_three_million_triangles(rt, ob):
    """3 000 000 triangles on the reference grid: 343 cells with ~8 750 insertions each, so HierarchicalGrid's cap of
    8192 (src/grid.rs:92-100) drops the highest ids of most cells. The cell lists equal the oracle's builder entry for
    entry; tiles of a 1080p frame equal the oracle's with duplicate removal on (it must respect the cut lists) and off."""
    import torch
    n, w, h, tile = 3000000, 1920, 1080, 16
    tris = ob.generate_test_triangles(n)
    scene = ob.Scene(None, tris)
    assert scene.grid.meta.fine_dims == (7, 7, 7)
    counts = np.diff(scene.grid.cell_start)
    assert counts.max() == 8192 and (counts == 8192).sum() > 100  # the cap bites
    cam = rt.Camera((3.0, -2.0, 49.0), math.pi + 0.2, -0.1).to_uniform(0.0, float(h))
    whole = {}
    for skip in (1, 0):
        core = rt.TraceCore(0, rt.default_params(skip_duplicate_triangles=skip))
        core.set_scene(None, tris)
        meta, n_cells, n_refs = core.build_grid()
        assert meta.fine_dims == (7, 7, 7) and n_refs == scene.grid.n_refs and bytes(meta)[:32] == bytes(scene.grid.meta)[:32]
        cs, idx = core.read_grid_csr()
        assert np.array_equal(cs, scene.grid.cell_start) and np.array_equal(idx, scene.grid.indices)
        for tx, ty in ((60, 33), (5, 60)) if skip else ((60, 33),):
            sh, osh, _ = one_tile(rt, ob, w, h, tile, tx, ty)
            gp, gs = core.trace_primary(cam, w, h, shard=sh)
            op, os_, st = ob.trace_primary(scene, ocam(ob, cam), w, h, params=oparams(ob, core.params), shard=osh, n_threads=8)
            assert_hits_equal(gp, op, f"skip={skip} tile ({tx},{ty}) primary")
            assert_hits_equal(gs, os_, f"skip={skip} tile ({tx},{ty}) shadow")
        if skip:
            p = torch.empty((w * h, 2), dtype=torch.int32, device="cuda")
            s = torch.empty_like(p)
            core.trace_primary(cam, w, h, out_primary=p, out_shadow=s)
            core.sync()
            whole = (p.cpu().numpy().view(rt.HIT_DTYPE).reshape(-1), s.cpu().numpy().view(rt.HIT_DTYPE).reshape(-1))
            assert (whole[0]["id"] != rt.RT_MISS).mean() > 0.15
        else:  # every entry tested: a band of the frame against the frame traced with duplicates left out
            band = rt.RtShard(w, 8, 60, 135)  # rows 480 .. 487
            gp, gs = core.trace_primary(cam, w, h, shard=band)
            assert_hits_equal(gp, whole[0].reshape(h, w)[480:488].reshape(-1), "band, every entry vs duplicates left out")
            assert_hits_equal(gs, whole[1].reshape(h, w)[480:488].reshape(-1), "band shadow, every entry vs duplicates left out")
        core.close()
