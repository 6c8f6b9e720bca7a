"""Randomised parity: many small random scenes, grids, cameras and parameter sets, GPU against the oracle, ids and
f32 bit patterns of t for primary and shadow rays, plus the shaded picture. The count of rays compared and of
mismatches is printed (expected mismatches: 0)."""
import ctypes as C
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def random_scene(ob, rng):
    kind = rng.integers(0, 4)
    nt = int(rng.integers(0, 1500)) if kind != 1 else 0
    nb = int(rng.integers(0, 60)) if kind != 0 else 0
    if nt == 0 and nb == 0:
        nt = 50
    ext = float(rng.choice([8.0, 40.0, 150.0]))
    tris = None
    if nt:
        c = rng.uniform(-ext, ext, (nt, 1, 3))
        size = rng.choice([0.02, 0.3, 1.0]) * ext
        v = (c + rng.normal(0, size, (nt, 3, 3))).astype(np.float32)
        if rng.random() < 0.3:  # exact duplicates: ties
            v[nt // 2:] = v[:nt - nt // 2]
        if rng.random() < 0.3:  # degenerate triangles
            v[::7, 2] = v[::7, 1]
        if rng.random() < 0.3:  # axis-aligned sheet
            v[::5, :, 1] = np.float32(0.0)
        tris = ob.triangles_from_xyz9(v.reshape(-1, 9))
        tris["material_id"] = rng.integers(0, 6, nt)
    boxes = None
    if nb:
        boxes = np.zeros(nb, ob.BOX_DTYPE)
        for i in range(nb):
            c = rng.uniform(-ext, ext, 3).astype(np.float32)
            hs = rng.uniform(0.01, 0.2, 3).astype(np.float32) * np.float32(ext)
            if i >= nb - 3 and rng.random() < 0.7:
                c1 = (c + rng.uniform(-0.3, 0.3, 3) * ext).astype(np.float32)
                boxes[i] = ob.make_box(np.minimum(c, c1) - hs - np.float32(0.5), np.maximum(c, c1) + hs + np.float32(0.5),
                                       center0=c, center1=c1, half=hs)
            else:
                boxes[i] = ob.make_box(c - hs, c + hs)
            boxes[i]["color"] = rng.uniform(0, 1, 3)
            boxes[i]["reflectivity"] = rng.choice([0.0, 0.0, 0.4, 0.9])
    return boxes, tris, ext


def test_random_scenes(rt, ob):
    from test_oracle_render import alpha_materials
    import os
    # RT_FUZZ_SEED / RT_FUZZ_TRIALS: longer soak runs with other seeds (the defaults are what the suite runs)
    rng = np.random.default_rng(int(os.environ.get("RT_FUZZ_SEED", "20251111")))
    srng = np.random.default_rng(int(os.environ.get("RT_FUZZ_SEED", "20251111")) + 1)  # shard choices, a stream of their own
    rays = mism = pixels = pix_mism = 0
    for trial in range(int(os.environ.get("RT_FUZZ_TRIALS", "48"))):
        boxes, tris, ext = random_scene(ob, rng)
        kw = dict(wgsl_compat=int(rng.integers(0, 2)), lod_cull=int(rng.integers(0, 2)),
                  max_dda_steps=int(rng.choice([200, 200, 23])), max_objects_per_cell=int(rng.choice([8192, 8192, 5])),
                  moving_box_count=int(rng.choice([3, 0])))
        cell = float(rng.choice([16.0, 16.0, 4.0, 1.0])) * (ext / 40.0 if rng.random() < 0.5 else 1.0)
        cap = int(rng.choice([8192, 8192, 6]))
        mats = alpha_materials(ob) if rng.random() < 0.5 else None
        # expensive-block plan: off, the default, or so low that every block with a candidate is traced four lanes per ray
        heavy = int(rng.choice([0, 64, 1, 3]))
        # block mask: every frame of a triangle-only scene without a plan (2), or as the default decides (1: lone frames get none)
        core = rt.TraceCore(0, rt.default_params(heavy_block_threshold=heavy, cull_empty_blocks=2 - trial % 2,
                                                    repack_candidates=int(trial % 3 == 1), **kw))
        core.set_scene(boxes, tris)
        core.set_materials(mats)
        try:
            core.build_grid(cell_size=cell, max_per_cell=cap)
        except rt.RtError:
            core.close()
            continue  # grid too large for this cell size
        scene = ob.Scene(boxes, tris, materials=mats, cell_size=cell, max_per_cell=cap)
        cs, idx = core.read_grid_csr()
        assert np.array_equal(cs, scene.grid.cell_start) and np.array_equal(idx, scene.grid.indices), f"trial {trial}: grid"
        params = ob.Params.default(**kw)
        for view in range(2):
            inside = rng.random() < 0.4
            pos = rng.uniform(-ext, ext, 3) * (0.5 if inside else 3.0)
            yaw, pitch = rng.uniform(0, 2 * math.pi), rng.uniform(-1.2, 1.2)
            if not inside:  # look at the scene
                yaw = math.atan2(-pos[0], -pos[2])
                pitch = math.atan2(-pos[1], math.hypot(pos[0], pos[2]))
            w, h = int(rng.integers(33, 130)), int(rng.integers(17, 80))
            cam = rt.Camera(tuple(pos), yaw, pitch).to_uniform(float(rng.uniform(0, 3)), float(h), show_grid=bool(rng.random() < 0.2))
            oc = ob.CameraUniform()
            C.memmove(C.byref(oc), C.byref(cam), C.sizeof(oc))
            p, s = core.trace_primary(cam, w, h)
            op, os_, st = ob.trace_primary(scene, oc, w, h, params=params)
            bad = int(((p["id"] != op["id"]) | (p["t"].view(np.uint32) != op["t"].view(np.uint32))).sum()
                      + ((s["id"] != os_["id"]) | (s["t"].view(np.uint32) != os_["t"].view(np.uint32))).sum())
            rays += st["rays"]
            mism += bad
            assert bad == 0, f"trial {trial} view {view}: {bad} records differ ({kw}, cell {cell}, cap {cap})"
            for again in range(2):  # the same frame again: now with the plan the first launch made
                p2, s2 = core.trace_primary(cam, w, h)
                assert p2.tobytes() == p.tobytes() and s2.tobytes() == s.tobytes(), \
                    f"trial {trial} view {view}: repeat {again} differs (heavy_block_threshold {heavy}, {core.heavy_blocks()} blocks listed)"
            bounces, shadows = int(rng.choice([1, 3, 8])), bool(rng.random() < 0.5)
            img = core.render(cam, w, h, max_bounces=bounces, shadows=shadows)
            oimg, _ = ob.render(scene, oc, w, h, params=params, max_bounces=bounces, shadows=shadows)
            badp = int((img != oimg).any(axis=1).sum())
            pixels += w * h
            pix_mism += badp
            assert badp == 0, f"trial {trial} view {view}: {badp} pixels differ"
            if view == 1:  # one rank's part of the same frame: plain round-robin, or an owner table with unequal shares
                tw, th, world = int(srng.choice([8, 16, 20, 32])), int(srng.choice([8, 16, 7])), int(srng.integers(2, 6))
                rank = int(srng.integers(0, world))
                owner = None
                if srng.random() < 0.5:
                    owner = [int(k) for k in srng.permutation(np.repeat(np.arange(world), srng.integers(1, 4, world)))]
                shard = rt.RtShard(tw, th, rank, world, owner)
                oshard = ob.Shard(tw, th, rank, world, owner)
                p, s = core.trace_primary(cam, w, h, shard=shard)
                op, os_, _ = ob.trace_primary(scene, oc, w, h, params=params, shard=oshard)
                bad = int(((p["id"] != op["id"]) | (p["t"].view(np.uint32) != op["t"].view(np.uint32))).sum()
                          + ((s["id"] != os_["id"]) | (s["t"].view(np.uint32) != os_["t"].view(np.uint32))).sum())
                assert len(p) == len(op) and bad == 0, f"trial {trial}: shard {tw}x{th} rank {rank}/{world} owner {owner}: {bad} records differ"
                img = core.render(cam, w, h, max_bounces=bounces, shadows=shadows, shard=shard)
                oimg, _ = ob.render(scene, oc, w, h, params=params, max_bounces=bounces, shadows=shadows, shard=oshard)
                assert img.shape == oimg.shape and not (img != oimg).any(), f"trial {trial}: sharded picture differs ({owner})"
        checked = core.checked_violations()  # a checked build of the library (tests/test_gpu_checked.py): no index went astray
        assert checked is None or checked[0] == 0, f"trial {trial}: index violations (count, (kind, a, b)) = {checked}"
        core.close()
    if os.environ.get("RT_EXPECT_CHECKED") == "1":
        assert checked is not None, "RT_LIB_PATH does not name a checked build"
        print(f"checked build: {checked[0]} index violations")
    print(f"random scenes: {rays} rays compared, {mism} mismatches; {pixels} shaded pixels compared, {pix_mism} mismatches")
    assert rays > (300000 if trial >= 40 else 50000)
