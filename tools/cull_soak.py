#!/usr/bin/env python3
"""Development tool: the block mask (rt_params.cull_empty_blocks = 2) against no mask on random cameras, grids and frame
shapes: hit records and pictures must be identical. RT_SOAK_TRIALS (default 2000), RT_SOAK_SEED."""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import ray_tracer_amd as rt  # noqa: E402

trials = int(os.environ.get("RT_SOAK_TRIALS", "2000"))
rng = np.random.default_rng(int(os.environ.get("RT_SOAK_SEED", "7")))
models = {"car": np.load(os.path.join(ROOT, "tests/golden/car_tris.npy")), "phoenix": np.load(os.path.join(ROOT, "tests/golden/phoenix_tris.npy"))}
rays = masked = culled_blocks = blocks_total = 0
t = 0
while t < trials:
    name = ("car", "phoenix")[int(rng.integers(2))]
    xyz9 = models[name]
    if rng.random() < 0.3:  # a random sub-soup at a random scale and place: other grids, other magnitudes
        xyz9 = xyz9[rng.random(len(xyz9)) < rng.uniform(0.02, 0.6)] * np.float32(10.0 ** rng.uniform(-2, 2)) + np.tile(
            (rng.normal(0, 1, 3) * 10.0 ** rng.uniform(0, 3)).astype(np.float32), 3)
        if len(xyz9) == 0:
            continue
    kw = dict(wgsl_compat=int(rng.integers(2)), fov=float(rng.uniform(0.2, 1.3)), max_dda_steps=int(rng.choice([200, 200, 40, 1000])))
    cores = [rt.TraceCore(0, rt.default_params(cull_empty_blocks=c, **kw)) for c in (2, 0)]
    try:
        ext = float(np.ptp(xyz9.reshape(-1, 3), axis=0).max())
        cell = float(rng.choice([0.0, 0.0, ext / rng.uniform(4, 90)]))
        ok = True
        for c in cores:
            c.set_scene(None, rt.triangles_from_xyz9(xyz9))
            try:
                c.build_grid(cell_size=cell)
            except rt.RtError:
                ok = False
                break
        if not ok:
            continue
        meta, _, _ = cores[0].grid_info()
        lo, hi = np.array(meta.bounds_min[:3], np.float64), np.array(meta.bounds_max[:3], np.float64)
        size = hi - lo
        for _ in range(int(rng.integers(4, 12))):
            kind = int(rng.integers(5))
            if kind == 0:
                pos = lo + rng.uniform(0, 1, 3) * size
            elif kind == 1:
                pos = lo + rng.uniform(-0.5, 1.5, 3) * size
            elif kind == 2:
                pos = (lo + hi) / 2 + rng.normal(0, 1, 3) * size.max() * 10.0 ** rng.uniform(0, 2.5)
            elif kind == 3:  # on a face of the bounds, or a hair off it
                pos = lo + rng.uniform(0, 1, 3) * size
                ax = int(rng.integers(3))
                pos[ax] = (lo, hi)[int(rng.integers(2))][ax] + float(rng.choice([0.0, 1e-4, -1e-4, 1e-2])) * max(1.0, abs(pos[ax]) * 1e-3)
            else:  # level with a cell layer: boxes cross the camera plane edge-on
                pos = lo + rng.uniform(-0.2, 1.2, 3) * size
                pos[1] = lo[1] + float(rng.integers(0, 4)) * float(meta.finest_cell_size)
            yaw = float(rng.uniform(0, 2 * math.pi))
            pitch = float(rng.choice([rng.uniform(-1.55, 1.55), 0.0, 1.5707, -1.5707]))
            w, h = int(rng.integers(16, 500)), int(rng.integers(16, 300))
            cam = rt.Camera(tuple(float(v) for v in pos), yaw, pitch).to_uniform(float(rng.uniform(0, 3)), float(h), kw["fov"])
            shard = None
            if rng.random() < 0.3:
                shard = rt.RtShard(int(rng.choice([8, 16, 32, 64])), int(rng.choice([8, 16, 40])), int(rng.integers(0, 3)), int(rng.integers(1, 4)))
            if rng.random() < 0.2:
                mb, shd = int(rng.choice([1, 8])), bool(rng.integers(2))
                a, b = (c.render(cam, w, h, shard=shard, max_bounces=mb, shadows=shd) for c in cores)
                same = np.array_equal(a, b)
                if not same:
                    bad = np.flatnonzero((a != b).any(axis=1))
                    print(f"render bounces {mb} shadows {shd}: {bad.size} pixels differ, first {[(int(q) % w, int(q) // w) for q in bad[:8]]}: {a[bad[:3]]} vs {b[bad[:3]]}")
                    a2, b2 = (c.render(cam, w, h, shard=shard, max_bounces=mb, shadows=shd) for c in cores)
                    print("again: masked == unmasked", np.array_equal(a2, b2), "masked same as before", np.array_equal(a, a2), "unmasked same as before", np.array_equal(b, b2))
            else:
                (p1, s1), (p0, s0) = (c.trace_primary(cam, w, h, shard=shard) for c in cores)
                same = (np.array_equal(p1["id"], p0["id"]) and np.array_equal(p1["t"].view(np.uint32), p0["t"].view(np.uint32))
                        and np.array_equal(s1["id"], s0["id"]) and np.array_equal(s1["t"].view(np.uint32), s0["t"].view(np.uint32)))
            n, k = cores[0].cull_stats()
            if not same:
                fresh = rt.TraceCore(0, rt.default_params(cull_empty_blocks=0, **kw))
                fresh.set_scene(None, rt.triangles_from_xyz9(xyz9))
                fresh.build_grid(cell_size=cell)
                pf, sf = fresh.trace_primary(cam, w, h, shard=shard)
                (p1, s1), (p0, s0) = (c.trace_primary(cam, w, h, shard=shard) for c in cores)
                d1 = np.flatnonzero((p1["id"] != pf["id"]) | (p1["t"].view(np.uint32) != pf["t"].view(np.uint32)))
                d0 = np.flatnonzero((p0["id"] != pf["id"]) | (p0["t"].view(np.uint32) != pf["t"].view(np.uint32)))
                e1 = np.flatnonzero((s1["id"] != sf["id"]) | (s1["t"].view(np.uint32) != sf["t"].view(np.uint32)))
                e0 = np.flatnonzero((s0["id"] != sf["id"]) | (s0["t"].view(np.uint32) != sf["t"].view(np.uint32)))
                print(f"again, against a fresh context: masked primary {d1.size} {d1[:6]} shadow {e1.size} {e1[:6]}; unmasked primary {d0.size} shadow {e0.size}; "
                      f"mask now {cores[0].cull_stats()}", flush=True)
                if d1.size:
                    print(p1[d1[:4]], pf[d1[:4]], [(int(b) % w, int(b) // w) for b in d1[:6]])
                print(f"MISMATCH trial {t}: {name} n={len(xyz9)} cell={cell} kw={kw} kind={kind} pos={pos.tolist()} yaw={yaw} pitch={pitch} {w}x{h} "
                      f"shard={(shard.tile_w, shard.tile_h, shard.first, shard.stride) if shard else None} blocks={n} kept={k}", flush=True)
                sys.exit(1)
            rays += w * h
            masked += 1 if n else 0
            blocks_total += n
            culled_blocks += n - k
            t += 1
    finally:
        for c in cores:
            c.close()
print(f"{t} frames, {rays} pixels, {masked} frames had a mask, {culled_blocks} of {blocks_total} blocks culled, 0 mismatches")
