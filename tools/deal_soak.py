#!/usr/bin/env python3
"""Development tool: the traversals that deal (ray, candidate) pairs out to a wave's lanes against the plain ones on
random scenes and cameras - hit records (rt_params.repack_candidates), ray lists, and pictures with reflection bounces
(rt_render_opts.plain_bounces). Everything must be identical. RT_SOAK_TRIALS (default 300), RT_SOAK_SEED."""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import ray_tracer_amd as rt  # noqa: E402
from render_scenes import bundled_model  # noqa: E402

trials = int(os.environ.get("RT_SOAK_TRIALS", "300"))
rng = np.random.default_rng(int(os.environ.get("RT_SOAK_SEED", "5")))
models = {n: np.load(os.path.join(ROOT, f"tests/golden/{n}_tris.npy")) for n in ("car", "phoenix")}
bad = frames = pixels = 0
for t in range(trials):
    kind = ("car", "phoenix", "soup")[int(rng.integers(3))]
    mats = textures = None
    if kind == "soup":
        n = int(rng.integers(500, 30000))
        c = rng.uniform(-50, 50, (n, 1, 3)).astype(np.float32)
        tris = rt.triangles_from_xyz9((c + rng.normal(0, rng.uniform(0.5, 4.0), (n, 3, 3)).astype(np.float32)).astype(np.float32).reshape(n, 9))
        cell = float(rng.choice([0.0, 2.0, 4.0, 9.0]))
        lo, hi = np.full(3, -50.0), np.full(3, 50.0)
    else:
        tris, mats, textures = bundled_model(rt, kind, models[kind])
        mats = mats.copy()
        style = int(rng.integers(4))
        if style == 1:
            mats["alpha_mode"] = 0
        elif style == 2:
            mats["metallic"] = rng.uniform(0.2, 0.9)
        elif style == 3:
            mats, textures = None, None
        cell = float(rng.choice([0.0, 0.0, 8.0]))
        v = models[kind].reshape(-1, 3)
        lo, hi = v.min(axis=0), v.max(axis=0)
    kw = dict(wgsl_compat=int(rng.integers(2)), max_dda_steps=int(rng.choice([200, 200, 30])), fov=float(rng.uniform(0.3, 1.2)))
    cores = [rt.TraceCore(0, rt.default_params(repack_candidates=r, **kw)) for r in (1, 0)]
    try:
        for c in cores:
            c.set_scene(None, tris)
            c.set_materials(mats)
            if textures is not None:
                c.set_textures(textures)
            c.build_grid(cell_size=cell)
        ext = float((hi - lo).max())
        for _ in range(3):
            where = rng.random()
            pos = rng.uniform(lo, hi) if where < 0.4 else (lo + hi) / 2 + rng.normal(0, 1, 3) * ext * (0.8 if where < 0.8 else 3.0)
            cam = rt.Camera(tuple(float(x) for x in pos), float(rng.uniform(0, 2 * math.pi)), float(rng.uniform(-1.2, 1.2)))
            w, h = int(rng.integers(40, 260)), int(rng.integers(30, 150))
            u = cam.to_uniform(0.0, float(h), kw["fov"])
            shadow = bool(rng.integers(2))
            a = cores[0].trace_primary(u, w, h, shadow=shadow)
            b = cores[1].trace_primary(u, w, h, shadow=shadow)
            ok = np.array_equal(a[0], b[0]) and (not shadow or np.array_equal(a[1], b[1]))
            bounces = int(rng.integers(2, 9))
            sh = bool(rng.integers(2))
            pa = cores[1].render(u, w, h, max_bounces=bounces, shadows=sh)
            pb = cores[1].render(u, w, h, max_bounces=bounces, shadows=sh, plain_bounces=True)
            ok = ok and np.array_equal(pa, pb)
            nr = int(rng.integers(1, 5000))
            o = rng.uniform(lo - 0.2 * ext, hi + 0.2 * ext, (nr, 3)).astype(np.float32)
            d = rng.normal(0, 1, (nr, 3)).astype(np.float32)
            any_hit = bool(rng.integers(2))
            ok = ok and np.array_equal(cores[0].trace_rays(o, d, any_hit=any_hit), cores[1].trace_rays(o, d, any_hit=any_hit))
            frames += 1
            pixels += w * h
            if not ok:
                bad += 1
                print(f"MISMATCH trial {t}: {kind} cell {cell} {kw} cam {pos} {w}x{h} bounces {bounces} shadows {sh}", flush=True)
    finally:
        for c in cores:
            c.close()
    if t % 50 == 49:
        print(f"{t + 1} trials, {frames} frames, {bad} mismatches", flush=True)
print(f"{frames} frames, {pixels} pixels, {bad} mismatches")
