#!/usr/bin/env python3
"""Development aid: randomised parity of the paths the randomised suite tests do not reach - the BVH walk (one lane and 32
lanes to a ray, copies passed over) against the oracle's walk, the all-pairs box routine at random sizes, and the
device-side rebuild (moved vertices every frame) against a fresh CPU build. RT_SOAK_SEED, RT_SOAK_TRIALS."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import ray_tracer_amd as rt  # noqa: E402
from oracle import binding as ob  # noqa: E402

seed = int(os.environ.get("RT_SOAK_SEED", "1"))
trials = int(os.environ.get("RT_SOAK_TRIALS", "40"))
rng = np.random.default_rng(seed)
bits = lambda a: np.ascontiguousarray(a).view(np.uint32)
rays_checked = pairs_checked = frames_checked = 0
for trial in range(trials):
    # --- BVH: a soup with copies, flat and degenerate triangles
    n = int(rng.integers(1, 3000))
    c = rng.uniform(-40, 40, (n, 1, 3)).astype(np.float32)
    v = (c + rng.normal(0, rng.choice([0.3, 2.0, 10.0]), (n, 3, 3))).astype(np.float32)
    if rng.random() < 0.5:
        v[rng.integers(0, n, n // 5)] = v[rng.integers(0, n, n // 5)]  # exact copies
    if rng.random() < 0.3:
        v[:, :, int(rng.integers(0, 3))] = np.round(v[:, :, 0])  # axis-aligned, zero-thickness boxes
    tris = ob.triangles_from_xyz9(v.reshape(n, 9))
    m = int(rng.integers(1, 4000))
    o = rng.uniform(-80, 80, (m, 3)).astype(np.float32)
    d = rng.normal(size=(m, 3)).astype(np.float32)
    d[rng.random(m) < 0.1, int(rng.integers(0, 3))] = 0.0
    d /= np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-20).astype(np.float32)
    want, _ = ob.Bvh(tris).trace(o, d, n_threads=4)
    for skip in (1, 0):
        core = rt.TraceCore(0, rt.default_params(skip_duplicate_triangles=skip))
        core.set_scene(None, tris)
        core.build_bvh(on_host=bool(rng.random() < 0.3))
        for lanes in ("1", "32"):
            os.environ["RT_BVH_LANES"] = lanes
            got = core.trace_rays_bvh(o, d)
            assert np.array_equal(bits(got), bits(want)), (seed, trial, "bvh", skip, lanes, int((bits(got) != bits(want)).sum()))
        os.environ.pop("RT_BVH_LANES")
        core.close()
    rays_checked += 4 * m
    # --- all pairs
    nr, nb = int(rng.integers(1, 700)), int(rng.integers(1, 3000))
    oo = rng.uniform(-30, 30, (nr, 3)).astype(np.float32)
    dd = rng.normal(size=(nr, 3)).astype(np.float32)
    dd[rng.random(nr) < 0.2, int(rng.integers(0, 3))] = 0.0
    cc = rng.uniform(-20, 20, (nb, 3)).astype(np.float32)
    hh = rng.uniform(0.0, 6, (nb, 3)).astype(np.float32)
    for wgsl in (0, 1):
        core = rt.TraceCore(0, rt.default_params(wgsl_compat=wgsl))
        got = core.intersect_aabb_pairs(oo, dd, cc - hh, cc + hh)
        assert np.array_equal(bits(got), bits(ob.intersect_aabb_pairs(oo, dd, cc - hh, cc + hh, wgsl))), (seed, trial, "pairs", wgsl)
        core.close()
    pairs_checked += 2 * nr * nb
    # --- rebuilds: the same soup moved about, every frame against a fresh CPU grid
    n = int(rng.integers(50, 2500))
    base = (rng.uniform(-60, 60, (n, 1, 3)) + rng.normal(0, rng.choice([1.0, 6.0]), (n, 3, 3))).astype(np.float32)
    cell = float(rng.choice([4.0, 8.0, 16.0]))
    mode = int(rng.integers(0, 2))
    core = rt.TraceCore(0)
    core.set_scene(None, ob.triangles_from_xyz9(base.reshape(n, 9)))
    core.build_grid(cell_size=cell, mode=mode)
    cam = ob.camera_to_uniform((0, 40, 200), 3.14159, -0.2, 0.0, 90.0)
    for frame in range(4):
        moved = (base * np.float32(1.0 + 0.1 * rng.random()) + rng.normal(0, 0.5, (1, 1, 3)).astype(np.float32)).astype(np.float32).reshape(n, 9)
        core.update_vertices(moved)
        core.build_grid(cell_size=cell, mode=mode)
        scene = ob.Scene(None, ob.triangles_from_xyz9(moved), cell_size=cell, mode=mode)
        cs, idx = core.read_grid_csr()
        assert np.array_equal(cs, scene.grid.cell_start) and np.array_equal(idx, scene.grid.indices), (seed, trial, "rebuild", frame, core.build_path())
        p, s = core.trace_primary(cam, 96, 54)
        op, os_, _ = ob.trace_primary(scene, cam, 96, 54)
        assert np.array_equal(p["id"], op["id"]) and np.array_equal(bits(p["t"]), bits(op["t"])), (seed, trial, "trace", frame)
        assert np.array_equal(s["id"], os_["id"]) and np.array_equal(bits(s["t"]), bits(os_["t"])), (seed, trial, "shadow", frame)
        frames_checked += 1
    core.close()
    if trial % 10 == 9:
        print(f"seed {seed} trial {trial + 1}: {rays_checked} BVH rays, {pairs_checked} pairs, {frames_checked} rebuilt frames, 0 mismatches", flush=True)
print(f"seed {seed}: {trials} trials, {rays_checked} BVH rays, {pairs_checked} box pairs, {frames_checked} rebuilt frames: 0 mismatches")
