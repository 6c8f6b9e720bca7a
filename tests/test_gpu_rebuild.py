"""The rebuild path of rt_build_grid that asks the host for nothing (rt_get_build_path == 1): the animated case,
new vertices every frame. Cell lists, metadata and traced frames must equal a fresh CPU build by the oracle
(HierarchicalGrid::build, src/grid.rs:110-216; TriangleGrid::build, src/grid_triangles.rs:29-124) exactly, whichever
path a frame takes - including the frames that do not fit the previous build's buffers and fall back."""
import math

import numpy as np
import pytest

from test_gpu_parity import assert_hits_equal, random_boxes

pytestmark = pytest.mark.gpu


def rotated(xyz9, frame, scale=1.0):
    centre = xyz9.reshape(-1, 3).mean(axis=0).astype(np.float32)
    a = np.float32(2 * math.pi * frame / 120)
    ca, sa = np.float32(math.cos(a)), np.float32(math.sin(a))
    v = (xyz9.reshape(-1, 3) - centre) * np.float32(scale)
    moved = np.stack([v[:, 0] * ca + v[:, 2] * sa, v[:, 1], -v[:, 0] * sa + v[:, 2] * ca], axis=1) + centre
    moved[:, 1] += np.float32(10 * (math.sin(2 * frame / 60) + 1) / 2)
    return moved.astype(np.float32).reshape(-1, 9)


def check_grid(core, ob, boxes, xyz9, **kw):
    scene = ob.Scene(boxes, ob.triangles_from_xyz9(xyz9), **kw)
    meta, nc, nr = core.grid_info()
    assert bytes(meta) == bytes(scene.grid.meta) or (bytes(meta)[:32] == bytes(scene.grid.meta)[:32] and meta.fine_dims == scene.grid.meta.fine_dims)
    assert (nc, nr) == (scene.grid.n_cells, scene.grid.n_refs)
    cs, idx = core.read_grid_csr()
    assert np.array_equal(cs, scene.grid.cell_start) and np.array_equal(idx, scene.grid.indices)
    return scene


def test_animated_rebuilds_take_the_device_side_path(core, ob, rt, phoenix_xyz9):
    core.set_scene(None, ob.triangles_from_xyz9(phoenix_xyz9))
    core.build_grid()
    assert core.build_path() == 0
    cam = ob.camera_to_uniform((-300, 250, 600), 2.7, -0.2, 0.0, 90.0)
    paths = []
    for frame in (1, 2, 30, 77, 119):
        moved = rotated(phoenix_xyz9, frame)
        core.update_vertices(moved)
        core.build_grid()
        paths.append(core.build_path())
        scene = check_grid(core, ob, None, moved)
        p, s = core.trace_primary(cam, 160, 90)
        op, os_, _ = ob.trace_primary(scene, cam, 160, 90)
        assert_hits_equal(p, op, f"frame {frame} primary")
        assert_hits_equal(s, os_, f"frame {frame} shadow")
    assert paths.count(1) >= 4, paths  # a frame may outgrow the buffers once; the others reuse them
    # a frame three times the size does not fit what the last build left behind: general path, same answer;
    # the frames after it reuse the larger buffers
    big = rotated(phoenix_xyz9, 10, scale=3.0)
    core.update_vertices(big)
    core.build_grid()
    assert core.build_path() == 0
    check_grid(core, ob, None, big)
    core.update_vertices(rotated(phoenix_xyz9, 11, scale=2.9))
    core.build_grid()
    assert core.build_path() == 1
    check_grid(core, ob, None, rotated(phoenix_xyz9, 11, scale=2.9))


@pytest.mark.parametrize("mode,cell,cap", [(0, 8.0, 0), (0, 8.0, 6), (1, 4.0, 0), (0, 40.0, 0)])
def test_rebuild_paths_agree_for_modes_caps_and_boxes(core, ob, rt, mode, cell, cap):
    """Boxes and triangles, both grid modes, a cap that cuts lists, and a coarse grid whose lists are longer than
    the ranking kernel accepts (the path then falls back by itself)."""
    boxes = random_boxes(ob, 25, 9, extent=40.0, moving_last=3) if mode == 0 else None
    tris0 = ob.generate_test_triangles(40000 if cell == 40.0 else 3000)
    xyz = np.concatenate([tris0["v0"], tris0["v1"], tris0["v2"]], axis=1).astype(np.float32)
    core.set_scene(boxes, ob.triangles_from_xyz9(xyz))
    core.build_grid(cell_size=cell, max_per_cell=cap, mode=mode)
    kw = dict(cell_size=cell, mode=mode)
    if cap:
        kw["max_per_cell"] = cap
    for frame in (3, 50):
        moved = rotated(xyz, frame)
        core.update_vertices(moved)
        core.build_grid(cell_size=cell, max_per_cell=cap, mode=mode)  # may outgrow the last build's buffers: either path
        check_grid(core, ob, boxes, moved, **kw)
        core.update_vertices(moved)
        core.build_grid(cell_size=cell, max_per_cell=cap, mode=mode)  # now it fits
        # 40 000 triangles in 40-unit cells: lists of more than 1024 entries -> general path
        assert core.build_path() == (0 if cell == 40.0 else 1)
        scene = check_grid(core, ob, boxes, moved, **kw)
        if mode == 0:
            cam = rt.Camera((10, 30, 140), math.pi, -0.2).to_uniform(0.5, 54.0)
            p, s = core.trace_primary(cam, 96, 54)
            op, os_, _ = ob.trace_primary(scene, ob.camera_to_uniform((10, 30, 140), math.pi, -0.2, 0.5, 54.0), 96, 54,
                                          params=ob.Params.default(finest_cell_size=16.0))
            assert_hits_equal(p, op, f"mode {mode} cap {cap} frame {frame}")
            assert_hits_equal(s, os_, f"mode {mode} cap {cap} frame {frame} shadow")
    # other parameters than the last build's: general path
    core.update_vertices(rotated(xyz, 60))
    core.build_grid(cell_size=cell * 0.5, max_per_cell=cap, mode=mode)
    assert core.build_path() == 0


def test_contexts_driven_by_a_thread_each_give_the_same_frames(ob, rt, phoenix_xyz9):
    """A renderer that keeps several frames of an animation in flight gives every context a host thread of its own (a
    rebuild ends with a small read-back its thread waits for; the other threads go on launching). Three contexts, three
    threads, frames dealt in turn: every frame's records equal those one context makes alone, which equal the oracle's
    on the frames compared with it."""
    import threading
    n_ctx, frames, w, h = 3, 18, 320, 180
    cam = ob.camera_to_uniform((-300, 250, 600), 2.7, -0.2, 0.0, float(h))
    moved = [rotated(phoenix_xyz9, 5 * f + 1) for f in range(frames)]
    alone = rt.TraceCore(0)
    alone.set_scene(None, ob.triangles_from_xyz9(phoenix_xyz9))
    alone.build_grid()
    want = []
    for f in range(frames):
        alone.update_vertices(moved[f])
        alone.build_grid()
        want.append(alone.trace_primary(cam, w, h))
    for f in (0, 7, 17):
        scene = ob.Scene(None, ob.triangles_from_xyz9(moved[f]))
        op, os_, _ = ob.trace_primary(scene, cam, w, h)
        assert_hits_equal(want[f][0], op, f"frame {f} primary, one context")
        assert_hits_equal(want[f][1], os_, f"frame {f} shadow, one context")
    alone.close()
    cores = [rt.TraceCore(0) for _ in range(n_ctx)]
    for c in cores:
        c.set_scene(None, ob.triangles_from_xyz9(phoenix_xyz9))
        c.build_grid()
    got, errors = [None] * frames, []

    def frames_of(k):
        try:
            for f in range(k, frames, n_ctx):
                cores[k].update_vertices(moved[f])
                cores[k].build_grid()
                got[f] = cores[k].trace_primary(cam, w, h)
        except Exception as e:  # noqa: BLE001 - reported by the main thread
            errors.append((k, repr(e)))

    for round_no in range(3):  # the same frames three times over: threads meet in a different order every time
        workers = [threading.Thread(target=frames_of, args=(k,)) for k in range(n_ctx)]
        for t in workers:
            t.start()
        for t in workers:
            t.join()
        assert not errors, errors
        for f in range(frames):
            assert_hits_equal(got[f][0], want[f][0], f"round {round_no} frame {f} primary")
            assert_hits_equal(got[f][1], want[f][1], f"round {round_no} frame {f} shadow")
    for c in cores:
        c.close()
